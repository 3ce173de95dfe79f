// fused collectors on the small / medium / large warehouse layouts (up to 29 x 16 cells), 2 and 4 agents
#define MARL_RW_COLLECT_BIG(X) X(2, 512) X(4, 512)
#define MARL_RW_PART_IDQN rware_idqn_collect_big
#define MARL_RW_PART_AC rware_ac_collect_big
#include "common.h"
namespace marl {
int rware_idqn_collect_p8(const RwParams& q, const marlhip_net_shape* s, const float* params, float epsilon, uint32_t round,
                          const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, int slot_base, int write_replay, int clear_stale,
                          int use_proper_termination, float* fin_return, int32_t* fin_length, hipStream_t stream);
int rware_ac_collect_p8(const RwParams& q, const marlhip_net_shape* s, const float* actor_params, uint32_t round, int max_len,
                        int use_proper_termination, float* batch_obs, int64_t* batch_act, float* batch_rew, uint8_t* batch_done,
                        float* batch_filled, float* fin_return, int32_t* fin_length, int32_t* t_max, hipStream_t stream);
}
#define MARL_RW_PART_NEXT_IDQN rware_idqn_collect_p8(MARL_ARGS_FWD_IDQN)
#define MARL_RW_PART_NEXT_AC rware_ac_collect_p8(MARL_ARGS_FWD_AC)
#define MARL_ARGS_FWD_IDQN q, s, params, epsilon, round, rs, rb, slot_base, write_replay, clear_stale, use_proper_termination, fin_return, fin_length, stream
#define MARL_ARGS_FWD_AC q, s, actor_params, round, max_len, use_proper_termination, batch_obs, batch_act, batch_rew, batch_done, batch_filled, fin_return, fin_length, t_max, stream
#include "rware_collect_part.h"
