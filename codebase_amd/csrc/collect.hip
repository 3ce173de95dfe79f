// extern "C" entry points of K2 (batched act) and the fused IDQN collector; kernels in collect_kernels.h
#include "collect_kernels.h"

using namespace marl;

extern "C" int marlhip_dqn_act(const marlhip_net_shape* s, const float* params, const float* obs, int32_t n_envs, float epsilon,
                               const float* u, const int32_t* rand_actions, uint64_t seed, const uint32_t* episode,
                               const int32_t* ep_length, int32_t* actions, float* q_out, void* stream) {
    MARL_REQUIRE(s && params && obs && actions, "dqn_act: NULL pointer");
    if (agent_map_validate(s) != 0) return -1;
    MARL_REQUIRE(n_envs > 0, "dqn_act: n_envs must be > 0");
    MARL_REQUIRE((u != nullptr) == (rand_actions != nullptr), "dqn_act: u and rand_actions must be given together");
    MARL_REQUIRE(u != nullptr || (episode != nullptr && ep_length != nullptr), "dqn_act: need injected noise or episode/ep_length");
    const int grid = (n_envs + 63) / 64;
#define X(d, h, a)                                                                                                          \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) {                                                           \
        using S_ = MlpShape<d, h, a>;                                                                                       \
        const size_t lds_bytes = (size_t)S_::NFWD * sizeof(float);                                                          \
        static LdsAttr attr_set;                                                                                       \
        if (attr_set.need()) {                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dqn_act_kernel<S_>),                                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                          \
            attr_set.done();                                                                                                \
        }                                                                                                                   \
        hipLaunchKernelGGL((dqn_act_kernel<S_>), dim3(grid), dim3(COL_BLOCK), lds_bytes, (hipStream_t)stream, s->n_agents,  \
                           n_envs, agent_map(s), params, obs, epsilon, u, rand_actions, seed, episode, ep_length, actions, q_out);        \
        MARL_CHECK_LAUNCH("dqn_act_kernel");                                                                                \
        return 0;                                                                                                           \
    }
    MARL_NET_SHAPES(X)
#undef X
    set_error("no act kernel for net shape D=%d H=%d A=%d", s->obs_dim, s->hidden, s->n_actions);
    return -1;
}

extern "C" int marlhip_idqn_collect(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const float* params, float epsilon,
                                    uint32_t round, const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb,
                                    int32_t slot_base, int32_t write_replay, int32_t clear_stale, int32_t use_proper_termination,
                                    float* fin_return, int32_t* fin_length, void* workspace, int64_t workspace_bytes, void* stream) {
    if (lbf_validate(cfg) != 0) return -1;
    MARL_REQUIRE(s && params && rs && rb && fin_return && fin_length, "idqn_collect: NULL pointer");
    ScratchScope scratch(workspace, workspace_bytes);
    if (agent_map_validate(s) != 0) return -1;
    MARL_REQUIRE(s->n_agents == cfg->n_agents && s->obs_dim == marlhip_lbf_obs_dim(cfg) && s->n_actions == 6,
                 "idqn_collect: net shape does not match the env (P=%d D=%d A=6 expected)", cfg->n_agents,
                 3 * (cfg->n_agents + cfg->n_food));
    MARL_REQUIRE(rs->n_agents == cfg->n_agents && rs->obs_dim == s->obs_dim && rs->max_len > 0 && rs->capacity > 0,
                 "idqn_collect: replay shape does not match the env");
    MARL_REQUIRE(!write_replay || (rb->obs && rb->act && rb->rew && rb->done && rb->filled), "idqn_collect: NULL replay buffer");
    MARL_REQUIRE(!write_replay || cfg->n_envs <= rs->capacity, "idqn_collect: n_envs %d > replay capacity %d", cfg->n_envs, rs->capacity);
    const LbfParams q = to_params(cfg);
    if (cfg->observe_id)
        return idqn_collect_dispatch_oid(cfg, s, q, params, epsilon, round, rs, rb, slot_base, write_replay, clear_stale, use_proper_termination,
                                         fin_return, fin_length, (hipStream_t)stream);
#define MARL_COLLECT_ARGS q, agent_map(s), params, epsilon, round, rs, rb, slot_base, write_replay, clear_stale, use_proper_termination, \
                          fin_return, fin_length, (hipStream_t)stream
#define X(p, f)                                                                                            \
    if (cfg->n_agents == p && cfg->n_food == f) {                                                          \
        if (s->hidden == 64) return launch_collect<LbfEnvT<p, f>, 64, false>(MARL_COLLECT_ARGS);                    \
        if (s->hidden == 128) return launch_collect<LbfEnvT<p, f>, 128, false>(MARL_COLLECT_ARGS);                  \
    }
    MARL_LBF_SHAPES(X)
#undef X
    set_error("idqn_collect: no kernel for %dp-%df hidden=%d", cfg->n_agents, cfg->n_food, s->hidden);
    return -1;
}
