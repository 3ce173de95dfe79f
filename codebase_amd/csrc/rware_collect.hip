// extern "C" entry points of the fused collectors on the warehouse env (their own translation unit: build time).
// Same kernels as collect.hip / ac_collect.hip, instantiated on RwEnvT: agents in registers, the shelf layer of the
// workgroup's 64 envs as bytes in LDS behind the weight packs.
#include "ac_collect_kernels.h"
#include "collect_kernels.h"

using namespace marl;

// (agents, max grid cells) with compiled fused collectors in this file: the tiny layouts (11 x 10 cells) with 2 / 4 agents;
// 8 agents and the small / medium / large layouts (up to 29 x 16 cells = 29 KB of LDS per workgroup) build in rware_collect_big.hip
#define MARL_RW_COLLECT_SHAPES(X) X(2, 128) X(4, 128)

namespace marl {
int rware_idqn_collect_big(const RwParams& q, const marlhip_net_shape* s, const float* params, float epsilon, uint32_t round,
                           const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, int slot_base, int write_replay, int clear_stale,
                           int use_proper_termination, float* fin_return, int32_t* fin_length, hipStream_t stream);
int rware_ac_collect_big(const RwParams& q, const marlhip_net_shape* s, const float* actor_params, uint32_t round, int max_len,
                         int use_proper_termination, float* batch_obs, int64_t* batch_act, float* batch_rew, uint8_t* batch_done,
                         float* batch_filled, float* fin_return, int32_t* fin_length, int32_t* t_max, hipStream_t stream);
}

static int rw_collect_check(const marlhip_rware_config* cfg, const marlhip_net_shape* s, const char* what) {
    if (rw_validate(cfg) != 0) return -1;
    MARL_REQUIRE(s != nullptr, "%s: net shape is NULL", what);
    MARL_REQUIRE(!cfg->observe_id, "%s: the warehouse collectors are not compiled with env.observe_id", what);
    MARL_REQUIRE(s->n_agents == cfg->n_agents && s->obs_dim == RW_OBS_DIM && s->n_actions == 5,
                 "%s: net shape does not match the env (P=%d D=%d A=5 expected)", what, cfg->n_agents, RW_OBS_DIM);
    return agent_map_validate(s);
}

extern "C" int marlhip_rware_idqn_collect(const marlhip_rware_config* cfg, const marlhip_net_shape* s, const float* params, float epsilon,
                                          uint32_t round, const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, int32_t slot_base,
                                          int32_t write_replay, int32_t clear_stale, int32_t use_proper_termination, float* fin_return,
                                          int32_t* fin_length, void* workspace, int64_t workspace_bytes, void* stream) {
    if (rw_collect_check(cfg, s, "rware_idqn_collect") != 0) return -1;
    ScratchScope scratch(workspace, workspace_bytes);
    MARL_REQUIRE(params && rs && rb && fin_return && fin_length, "rware_idqn_collect: NULL pointer");
    MARL_REQUIRE(rs->n_agents == cfg->n_agents && rs->obs_dim == RW_OBS_DIM && rs->max_len > 0 && rs->capacity > 0,
                 "rware_idqn_collect: replay shape does not match the env");
    MARL_REQUIRE(!write_replay || rs->capacity >= cfg->n_envs, "rware_idqn_collect: replay capacity %d < n_envs %d", rs->capacity, cfg->n_envs);
    const RwParams q = to_rw_params(cfg);
#define MARL_ARGS q, agent_map(s), params, epsilon, round, rs, rb, slot_base, write_replay, clear_stale, use_proper_termination, fin_return, \
                  fin_length, (hipStream_t)stream
#define X(p, cells)                                                                                   \
    if (cfg->n_agents == p && q.rows * q.cols <= cells) {                                             \
        if (s->hidden == 64) return launch_collect<RwEnvT<p, cells>, 64, false>(MARL_ARGS);           \
        if (s->hidden == 128) return launch_collect<RwEnvT<p, cells>, 128, false>(MARL_ARGS);         \
    }
    MARL_RW_COLLECT_SHAPES(X)
#undef X
#undef MARL_ARGS
    return rware_idqn_collect_big(q, s, params, epsilon, round, rs, rb, slot_base, write_replay, clear_stale, use_proper_termination, fin_return,
                                  fin_length, (hipStream_t)stream);
}

extern "C" int marlhip_rware_ac_collect(const marlhip_rware_config* cfg, const marlhip_net_shape* s, const float* actor_params, uint32_t round,
                                        int32_t max_len, int32_t use_proper_termination, float* batch_obs, int64_t* batch_act, float* batch_rew,
                                        uint8_t* batch_done, float* batch_filled, float* fin_return, int32_t* fin_length, int32_t* t_max,
                                        void* workspace, int64_t workspace_bytes, void* stream) {
    if (rw_collect_check(cfg, s, "rware_ac_collect") != 0) return -1;
    ScratchScope scratch(workspace, workspace_bytes);
    MARL_REQUIRE(actor_params && batch_obs && batch_act && batch_rew && batch_done && batch_filled && fin_return && fin_length && t_max,
                 "rware_ac_collect: NULL pointer");
    MARL_REQUIRE(max_len > 0, "rware_ac_collect: max_len must be > 0");
    const RwParams q = to_rw_params(cfg);
#define MARL_ARGS q, agent_map(s), actor_params, round, max_len, use_proper_termination, batch_obs, batch_act, batch_rew, batch_done, \
                  batch_filled, fin_return, fin_length, t_max, (hipStream_t)stream
#define X(p, cells)                                                                                   \
    if (cfg->n_agents == p && q.rows * q.cols <= cells) {                                             \
        if (s->hidden == 64) return launch_ac_collect<RwEnvT<p, cells>, 64, false>(MARL_ARGS);        \
        if (s->hidden == 128) return launch_ac_collect<RwEnvT<p, cells>, 128, false>(MARL_ARGS);      \
    }
    MARL_RW_COLLECT_SHAPES(X)
#undef X
#undef MARL_ARGS
    return rware_ac_collect_big(q, s, actor_params, round, max_len, use_proper_termination, batch_obs, batch_act, batch_rew, batch_done,
                                batch_filled, fin_return, fin_length, t_max, (hipStream_t)stream);
}

// the second pass of a rollout on the warehouse (AcGhost, common.h; marlhip_ac_collect_later_episodes)
extern "C" int marlhip_rware_ac_collect_later_episodes(const marlhip_rware_config* cfg, const marlhip_net_shape* s, const float* actor_params,
                                                       uint32_t round, int32_t max_len, const int32_t* env_ids, const int32_t* t_start,
                                                       int32_t n_envs, int32_t t_stop, int32_t cap, float* ret, int32_t* meta, int32_t* cnt,
                                                       void* workspace, int64_t workspace_bytes, void* stream) {
    MARL_REQUIRE(cfg && env_ids && t_start && ret && meta && cnt, "rware_ac_collect_later_episodes: NULL pointer");
    MARL_REQUIRE(n_envs > 0 && t_stop > 0 && cap > 0, "rware_ac_collect_later_episodes: n_envs %d, t_stop %d, cap %d must be > 0", n_envs, t_stop, cap);
    marlhip_rware_config c2 = *cfg;
    c2.n_envs = n_envs;
    (void)hipMemsetAsync(cnt, 0, (size_t)n_envs * sizeof(int32_t), (hipStream_t)stream);
    const AcGhost g = {env_ids, t_start, t_stop < max_len ? t_stop : max_len, cap, ret, meta, cnt};
    AcGhostScope scope(g);
    return marlhip_rware_ac_collect(&c2, s, actor_params, round, max_len, 0, ret, reinterpret_cast<int64_t*>(meta), ret, reinterpret_cast<uint8_t*>(meta), ret,
                                    ret, meta, meta + 2 * (size_t)n_envs * cap - 1, workspace, workspace_bytes, stream);
}

#if MARL_ACOL_PROF
// profiling builds only: read and clear this translation unit's in-kernel region counters (ac_collect_kernels.h)
extern "C" int marlhip_debug_acol_prof(unsigned long long* out16) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(marl::acol_prof), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(marl::acol_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
