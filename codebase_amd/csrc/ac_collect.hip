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

// ---- the bookkeeping of ONE step of a modular rollout (recurrent actors, actors on the GEMM path: the forward and the env step are
// their own launches) - ac/train.py:90-110 for all N envs in one launch instead of a dozen tensor operations: masked writes of the
// still-running envs into the time-major batch, the first episode's statistics, the records of later episodes (envs that keep
// auto-resetting while the others finish), running &= ~finished.
namespace {
__global__ __launch_bounds__(256) void ac_store_step_kernel(int N, int P, int D, int t, int proper_term, uint8_t* __restrict__ running,
                                                            const float* __restrict__ obs /* [P][N][D] */, const int64_t* __restrict__ acts /* [P][N] */,
                                                            const float* __restrict__ rew /* [P][N] */, const uint8_t* __restrict__ done,
                                                            const uint8_t* __restrict__ trunc, const float* __restrict__ env_fin_return /* [P][N] */,
                                                            const int32_t* __restrict__ env_fin_length, float* __restrict__ b_obs_t1 /* [N][P*D] */,
                                                            int64_t* __restrict__ b_act_t /* [N][P] */, float* __restrict__ b_rew_t, uint8_t* __restrict__ b_done_t1,
                                                            float* __restrict__ b_fill_t, float* __restrict__ fin_ret /* [P][N] */,
                                                            int32_t* __restrict__ fin_len, int32_t* __restrict__ later_cnt, int later_cap,
                                                            float* __restrict__ later_ret /* [cap][P] */, int32_t* __restrict__ later_meta /* [cap][3] */) {
    const int PD = P * D;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * PD) return;
    const int n = (int)(i / PD), e = (int)(i - (int64_t)n * PD);
    const bool run = running[n] != 0;  // (read by every thread of the env before thread e == 0 of it may clear it: see the launch note)
    const bool fin = (done[n] | trunc[n]) != 0;
    if (run) {
        const int p = e / D, d = e - p * D;
        b_obs_t1[(size_t)n * PD + e] = obs[((size_t)p * N + n) * D + d];
    }
    if (e != 0) return;
    b_fill_t[n] = run ? 1.f : 0.f;
    if (run) {
        for (int p = 0; p < P; ++p) {
            b_act_t[(size_t)n * P + p] = acts[(size_t)p * N + n];
            b_rew_t[(size_t)n * P + p] = rew[(size_t)p * N + n];
        }
        b_done_t1[n] = (proper_term ? done[n] != 0 : fin) ? 1 : 0;
        if (fin) {
            for (int p = 0; p < P; ++p) fin_ret[(size_t)p * N + n] = env_fin_return[(size_t)p * N + n];
            fin_len[n] = env_fin_length[n];
        }
    } else if (fin && later_cnt != nullptr) {
        const int k = atomicAdd(later_cnt, 1);
        if (k < later_cap) {
            for (int p = 0; p < P; ++p) later_ret[(size_t)k * P + p] = env_fin_return[(size_t)p * N + n];
            later_meta[3 * k] = t + 1;
            later_meta[3 * k + 1] = n;
            later_meta[3 * k + 2] = env_fin_length[n];
        }
    }
}
// second launch: running &= ~finished (its own launch, so that every thread of the step above has read `running` first)
__global__ __launch_bounds__(256) void ac_clear_running_kernel(int N, uint8_t* __restrict__ running, const uint8_t* __restrict__ done,
                                                               const uint8_t* __restrict__ trunc) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < N && (done[n] | trunc[n]) != 0) running[n] = 0;
}
}  // namespace

extern "C" int marlhip_ac_store_step(int32_t n_envs, int32_t n_agents, int32_t obs_dim, int32_t t, int32_t use_proper_termination, uint8_t* running,
                                     const float* obs, const int64_t* actions, const float* rewards, const uint8_t* done, const uint8_t* truncated,
                                     const float* env_fin_return, const int32_t* env_fin_length, float* batch_obs_t1, int64_t* batch_act_t,
                                     float* batch_rew_t, uint8_t* batch_done_t1, float* batch_filled_t, float* fin_return, int32_t* fin_length,
                                     int32_t* later_count, int32_t later_cap, float* later_returns, int32_t* later_meta, void* stream) {
    MARL_REQUIRE(n_envs > 0 && n_agents > 0 && obs_dim > 0 && t >= 0, "ac_store_step: bad sizes");
    MARL_REQUIRE(running && obs && actions && rewards && done && truncated && env_fin_return && env_fin_length && batch_obs_t1 && batch_act_t &&
                     batch_rew_t && batch_done_t1 && batch_filled_t && fin_return && fin_length, "ac_store_step: NULL pointer");
    MARL_REQUIRE(later_count == nullptr || (later_cap > 0 && later_returns && later_meta), "ac_store_step: later-episode buffers");
    const int64_t total = (int64_t)n_envs * n_agents * obs_dim;
    hipLaunchKernelGGL(ac_store_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_envs, n_agents, obs_dim, t,
                       use_proper_termination, running, obs, actions, rewards, done, truncated, env_fin_return, env_fin_length, batch_obs_t1, batch_act_t,
                       batch_rew_t, batch_done_t1, batch_filled_t, fin_return, fin_length, later_count, later_cap, later_returns, later_meta);
    hipLaunchKernelGGL(ac_clear_running_kernel, dim3((n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_envs, running, done, truncated);
    MARL_CHECK_LAUNCH("ac_store_step");
    return 0;
}

#if MARL_ACOL_PROF
// profiling builds only: read and clear this translation unit's in-kernel region counters (ac_collect_kernels.h)
extern "C" int marlhip_debug_acol_prof_lbf(unsigned long long* out16) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(marl::acol_prof), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(marl::acol_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
