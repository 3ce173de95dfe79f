// extern "C" entry point of the fused actor-critic rollout collector; kernel in ac_collect_kernels.h
#include "ac_collect_kernels.h"

using namespace marl;

extern "C" int marlhip_ac_collect(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const float* actor_params, uint32_t round,
                                  int32_t max_len, int32_t use_proper_termination, float* batch_obs, int64_t* batch_act,
                                  float* batch_rew, uint8_t* batch_done, float* batch_filled, float* fin_return,
                                  int32_t* fin_length, int32_t* t_max, void* workspace, int64_t workspace_bytes, void* stream) {
    if (lbf_validate(cfg) != 0) return -1;
    ScratchScope scratch(workspace, workspace_bytes);
    MARL_REQUIRE(s && actor_params && batch_obs && batch_act && batch_rew && batch_done && batch_filled && fin_return && fin_length &&
                     t_max, "ac_collect: NULL pointer");
    MARL_REQUIRE(s->n_agents == cfg->n_agents && s->obs_dim == marlhip_lbf_obs_dim(cfg) && s->n_actions == 6,
                 "ac_collect: net shape does not match the env (P=%d D=%d A=6 expected)", cfg->n_agents,
                 3 * (cfg->n_agents + cfg->n_food));
    MARL_REQUIRE(max_len > 0, "ac_collect: max_len must be > 0");
    if (agent_map_validate(s) != 0) return -1;
    const LbfParams q = to_params(cfg);
    if (cfg->observe_id)
        return ac_collect_dispatch_oid(cfg, s, q, actor_params, round, max_len, use_proper_termination, batch_obs, batch_act, batch_rew,
                                       batch_done, batch_filled, fin_return, fin_length, t_max, (hipStream_t)stream);
#define MARL_ACOL_ARGS q, agent_map(s), actor_params, round, max_len, use_proper_termination, batch_obs, batch_act, batch_rew, batch_done, \
                       batch_filled, fin_return, fin_length, t_max, (hipStream_t)stream
#define X(p, f)                                                                                         \
    if (cfg->n_agents == p && cfg->n_food == f) {                                                       \
        if (s->hidden == 64) return launch_ac_collect<LbfEnvT<p, f>, 64, false>(MARL_ACOL_ARGS);                 \
        if (s->hidden == 128) return launch_ac_collect<LbfEnvT<p, f>, 128, false>(MARL_ACOL_ARGS);               \
    }
    MARL_LBF_SHAPES(X)
#undef X
    set_error("ac_collect: no kernel for %dp-%df hidden=%d", cfg->n_agents, cfg->n_food, s->hidden);
    return -1;
}

namespace {
int ghost_check(const int32_t* env_ids, const int32_t* t_start, int32_t n, int32_t t_stop, int32_t cap, const float* ret, const int32_t* meta,
                const int32_t* cnt, const char* what) {
    MARL_REQUIRE(env_ids && t_start && ret && meta && cnt, "%s: NULL pointer", what);
    MARL_REQUIRE(n > 0 && t_stop > 0 && cap > 0, "%s: n_envs %d, t_stop %d, cap %d must be > 0", what, n, t_stop, cap);
    return 0;
}
}  // namespace

// the second pass of a rollout (AcGhost, common.h): the same collector kernels over the listed envs, writing episode records only.
// The batch / statistics pointers of the first-pass signature are not touched in this mode; they get the record buffers as stand-ins.
extern "C" int marlhip_ac_collect_later_episodes(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const float* actor_params, uint32_t round,
                                                 int32_t max_len, const int32_t* env_ids, const int32_t* t_start, int32_t n_envs, int32_t t_stop,
                                                 int32_t cap, float* ret, int32_t* meta, int32_t* cnt, void* workspace, int64_t workspace_bytes,
                                                 void* stream) {
    MARL_REQUIRE(cfg != nullptr, "ac_collect_later_episodes: NULL config");
    if (ghost_check(env_ids, t_start, n_envs, t_stop, cap, ret, meta, cnt, "ac_collect_later_episodes") != 0) return -1;
    marlhip_lbf_config c2 = *cfg;
    c2.n_envs = n_envs;
    (void)hipMemsetAsync(cnt, 0, (size_t)n_envs * sizeof(int32_t), (hipStream_t)stream);
    const AcGhost g = {env_ids, t_start, t_stop < max_len ? t_stop : max_len, cap, ret, meta, cnt};
    AcGhostScope scope(g);
    return marlhip_ac_collect(&c2, s, actor_params, round, max_len, 0, ret, reinterpret_cast<int64_t*>(meta), ret, reinterpret_cast<uint8_t*>(meta), ret, ret,
                              meta, meta + 2 * (size_t)n_envs * cap - 1, workspace, workspace_bytes, stream);
}
