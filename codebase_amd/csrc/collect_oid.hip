// the env.observe_id instantiations of the fused IDQN collector (their own translation unit: build time)
#include "collect_kernels.h"

namespace marl {

int idqn_collect_dispatch_oid(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const LbfParams& q, const float* params, float epsilon,
                              uint32_t round, const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, int slot_base, int write_replay,
                              int clear_stale, int use_proper_termination, float* fin_return, int32_t* fin_length, hipStream_t stream) {
#define MARL_COLLECT_ARGS q, agent_map(s), params, epsilon, round, rs, rb, slot_base, write_replay, clear_stale, use_proper_termination, \
                          fin_return, fin_length, stream
#define X(p, f)                                                                                            \
    if (cfg->n_agents == p && cfg->n_food == f) {                                                          \
        if (s->hidden == 64) return launch_collect<LbfEnvT<p, f>, 64, true>(MARL_COLLECT_ARGS);                     \
        if (s->hidden == 128) return launch_collect<LbfEnvT<p, f>, 128, true>(MARL_COLLECT_ARGS);                   \
    }
    MARL_LBF_SHAPES(X)
#undef X
    set_error("idqn_collect: no kernel for %dp-%df hidden=%d (observe_id)", cfg->n_agents, cfg->n_food, s->hidden);
    return -1;
}

}  // namespace marl
