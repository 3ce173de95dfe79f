// One slice of the warehouse collector instantiations (entry points and argument checks: rware_collect.hip).  The including
// .hip defines MARL_RW_COLLECT_BIG(X) = its (agents, max cells) list, MARL_RW_PART_IDQN / MARL_RW_PART_AC = its function names
// and MARL_RW_PART_NEXT_IDQN / _AC = what to return when no shape matches (the next slice, or an error).
#include "ac_collect_kernels.h"
#include "collect_kernels.h"

namespace marl {

int MARL_RW_PART_IDQN(const RwParams& q, const marlhip_net_shape* s, const float* params, float epsilon, uint32_t round,
                           const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, int slot_base, int write_replay, int clear_stale,
                           int use_proper_termination, float* fin_return, int32_t* fin_length, hipStream_t stream) {
#define MARL_ARGS q, agent_map(s), params, epsilon, round, rs, rb, slot_base, write_replay, clear_stale, use_proper_termination, fin_return, \
                  fin_length, stream
#define X(p, cells)                                                                                   \
    if (q.n_agents == p && q.rows * q.cols <= cells) {                                                \
        if (s->hidden == 64) return launch_collect<RwEnvT<p, cells>, 64, false>(MARL_ARGS);           \
        if (s->hidden == 128) return launch_collect<RwEnvT<p, cells>, 128, false>(MARL_ARGS);         \
    }
    MARL_RW_COLLECT_BIG(X)
#undef X
#undef MARL_ARGS
    return MARL_RW_PART_NEXT_IDQN;
}

int MARL_RW_PART_AC(const RwParams& q, const marlhip_net_shape* s, const float* actor_params, uint32_t round, int max_len,
                         int use_proper_termination, float* batch_obs, int64_t* batch_act, float* batch_rew, uint8_t* batch_done,
                         float* batch_filled, float* fin_return, int32_t* fin_length, int32_t* t_max, hipStream_t stream) {
#define MARL_ARGS q, agent_map(s), actor_params, round, max_len, use_proper_termination, batch_obs, batch_act, batch_rew, batch_done, \
                  batch_filled, fin_return, fin_length, t_max, stream
#define X(p, cells)                                                                                   \
    if (q.n_agents == p && q.rows * q.cols <= cells) {                                                \
        if (s->hidden == 64) return launch_ac_collect<RwEnvT<p, cells>, 64, false>(MARL_ARGS);        \
        if (s->hidden == 128) return launch_ac_collect<RwEnvT<p, cells>, 128, false>(MARL_ARGS);      \
    }
    MARL_RW_COLLECT_BIG(X)
#undef X
#undef MARL_ARGS
    return MARL_RW_PART_NEXT_AC;
}

}  // namespace marl
