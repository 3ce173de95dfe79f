// the env.observe_id instantiations of the fused actor-critic collector (their own translation unit: build time)
#include "ac_collect_kernels.h"

namespace marl {

int ac_collect_dispatch_oid(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const LbfParams& q, const float* actor_params,
                            uint32_t round, int max_len, int use_proper_termination, float* batch_obs, int64_t* batch_act, float* batch_rew,
                            uint8_t* batch_done, float* batch_filled, float* fin_return, int32_t* fin_length, int32_t* t_max,
                            hipStream_t stream) {
#define MARL_ACOL_ARGS q, agent_map(s), actor_params, round, max_len, use_proper_termination, batch_obs, batch_act, batch_rew, batch_done, \
                       batch_filled, fin_return, fin_length, t_max, stream
#define X(p, f)                                                                                         \
    if (cfg->n_agents == p && cfg->n_food == f) {                                                       \
        if (s->hidden == 64) return launch_ac_collect<LbfEnvT<p, f>, 64, true>(MARL_ACOL_ARGS);                  \
        if (s->hidden == 128) return launch_ac_collect<LbfEnvT<p, f>, 128, true>(MARL_ACOL_ARGS);                \
    }
    MARL_LBF_SHAPES(X)
#undef X
    set_error("ac_collect: no kernel for %dp-%df hidden=%d (observe_id)", cfg->n_agents, cfg->n_food, s->hidden);
    return -1;
}

}  // namespace marl
