// Batched Level-Based Foraging kernels (K1): reset / observe / step over N env records in HBM.
// One lane = one env; a 256-thread workgroup owns 256 consecutive envs.  Observations are built
// in registers, staged through LDS and written as one contiguous run per (agent, workgroup) so
// the obs[P][N][D] stores are fully coalesced (per-lane stores would be 4-byte writes at a
// 4*D-byte stride).  HBM-bound integer work: algorithmic bytes per env-step =
// 2*S + 4*P (actions i32) + 4*P*D (obs) + 4*P (rew) + 2 (done,trunc), S = state stride.
#include "common.h"

namespace marl {

constexpr int ENV_BLOCK = 256;

template <int P, int F>
__device__ __forceinline__ void write_obs_tile(const LbfParams& q, const LbfState<P, F>& s, bool valid, float* tile,
                                               float* __restrict__ obs, int n0, int cnt) {
    constexpr int D0 = 3 * (F + P);
    const int idw = q.observe_id ? P : 0, D = D0 + idw;  // ObserveID: one-hot agent index first (wrappers.py:97-103)
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (valid) {
            LbfObs<P, F> o;
            lbf_observe(q, s, p, o);
            for (int d = 0; d < idw; ++d) tile[tid * D + d] = d == p ? 1.f : 0.f;
#pragma unroll
            for (int d = 0; d < D0; ++d) tile[tid * D + idw + d] = o.v[d];
        }
        __syncthreads();
        float* dst = obs + ((size_t)p * q.n_envs + n0) * D;
        for (int i = tid; i < cnt * D; i += ENV_BLOCK) dst[i] = tile[i];
        __syncthreads();
    }
}

template <int P, int F>
__global__ __launch_bounds__(ENV_BLOCK) void lbf_reset_kernel(LbfParams q, marlhip_lbf_buffers b, const uint8_t* __restrict__ mask,
                                                              float* __restrict__ obs) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int n0 = blockIdx.x * ENV_BLOCK;
    const int n = n0 + threadIdx.x;
    const int cnt = min(ENV_BLOCK, q.n_envs - n0);
    const int stride = lbf_state_stride(P, F);
    const bool valid = n < q.n_envs;
    LbfState<P, F> s;
    if (valid) {
        if (mask == nullptr || mask[n]) {
            const uint32_t epi = b.episode[n];
            b.episode[n] = epi + 1;
            DrawStream rng;
            rng.init(q.seed, (uint32_t)n, epi, STREAM_RESET);
            lbf_reset(q, s, rng);
            lbf_store(b.state + (size_t)n * stride, s);
#pragma unroll
            for (int p = 0; p < P; ++p) b.ep_return[(size_t)p * q.n_envs + n] = 0.f;
            b.ep_length[n] = 0;
        } else {
            lbf_load(b.state + (size_t)n * stride, s);
        }
    }
    if (obs != nullptr) write_obs_tile<P, F>(q, s, valid, tile, obs, n0, cnt);
}

template <int P, int F>
__global__ __launch_bounds__(ENV_BLOCK) void lbf_observe_kernel(LbfParams q, marlhip_lbf_buffers b, float* __restrict__ obs) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int n0 = blockIdx.x * ENV_BLOCK;
    const int n = n0 + threadIdx.x;
    const int cnt = min(ENV_BLOCK, q.n_envs - n0);
    const bool valid = n < q.n_envs;
    LbfState<P, F> s;
    if (valid) lbf_load(b.state + (size_t)n * lbf_state_stride(P, F), s);
    write_obs_tile<P, F>(q, s, valid, tile, obs, n0, cnt);
}

template <int P, int F>
__global__ __launch_bounds__(ENV_BLOCK) void lbf_step_kernel(LbfParams q, marlhip_lbf_buffers b, const uint8_t* __restrict__ active,
                                                             const int32_t* __restrict__ actions, float* __restrict__ obs,
                                                             float* __restrict__ rewards, uint8_t* __restrict__ done_out,
                                                             uint8_t* __restrict__ trunc_out, float* __restrict__ fin_return,
                                                             int32_t* __restrict__ fin_length, int auto_reset,
                                                             float* __restrict__ final_obs) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr int D = 3 * (F + P);
    const int n0 = blockIdx.x * ENV_BLOCK;
    const int n = n0 + threadIdx.x;
    const int cnt = min(ENV_BLOCK, q.n_envs - n0);
    const int stride = lbf_state_stride(P, F);
    const bool valid = n < q.n_envs;
    LbfState<P, F> s;
    if (valid) {
        uint8_t* rec = b.state + (size_t)n * stride;
        lbf_load(rec, s);
        const bool act_on = (active == nullptr) || active[n];
        float rw[P];
        bool done = false, trunc = false;
#pragma unroll
        for (int p = 0; p < P; ++p) rw[p] = 0.f;
        if (act_on) {
            int a[P];
            double raw[P];
#pragma unroll
            for (int p = 0; p < P; ++p) a[p] = actions[(size_t)p * q.n_envs + n];
            lbf_step(q, s, a, raw, done);
            trunc = q.time_limit > 0 && s.step >= q.time_limit;  // gymnasium TimeLimit
            // RecordEpisodeStatistics: fp32 running sum of the RAW per-agent rewards
            const int len = b.ep_length[n] + 1;
            b.ep_length[n] = len;
            float ret[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                ret[p] = b.ep_return[(size_t)p * q.n_envs + n] + (float)raw[p];
                b.ep_return[(size_t)p * q.n_envs + n] = ret[p];
            }
            lbf_wrap_rewards<P>(q, (uint32_t)n, raw, rw);
            if (done || trunc) {
#pragma unroll
                for (int p = 0; p < P; ++p) fin_return[(size_t)p * q.n_envs + n] = ret[p];
                fin_length[n] = len;
                if (auto_reset) {
                    if (final_obs != nullptr) {
#pragma unroll
                        for (int p = 0; p < P; ++p) {
                            LbfObs<P, F> o;
                            lbf_observe(q, s, p, o);
                            const int idw = q.observe_id ? P : 0;
                            float* fo = final_obs + ((size_t)p * q.n_envs + n) * (D + idw);
                            for (int d = 0; d < idw; ++d) fo[d] = d == p ? 1.f : 0.f;
#pragma unroll
                            for (int d = 0; d < D; ++d) fo[idw + d] = o.v[d];
                        }
                    }
                    const uint32_t epi = b.episode[n];
                    b.episode[n] = epi + 1;
                    DrawStream rng;
                    rng.init(q.seed, (uint32_t)n, epi, STREAM_RESET);
                    lbf_reset(q, s, rng);
#pragma unroll
                    for (int p = 0; p < P; ++p) b.ep_return[(size_t)p * q.n_envs + n] = 0.f;
                    b.ep_length[n] = 0;
                }
            }
            lbf_store(rec, s);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) rewards[(size_t)p * q.n_envs + n] = rw[p];
        done_out[n] = done ? 1 : 0;
        trunc_out[n] = trunc ? 1 : 0;
    }
    write_obs_tile<P, F>(q, s, valid, tile, obs, n0, cnt);
}

}  // namespace marl

using namespace marl;

extern "C" int marlhip_lbf_state_stride(const marlhip_lbf_config* cfg) {
    if (lbf_validate(cfg) != 0) return -1;
    return lbf_state_stride(cfg->n_agents, cfg->n_food);
}

extern "C" int marlhip_lbf_obs_dim(const marlhip_lbf_config* cfg) {
    MARL_REQUIRE(cfg != nullptr, "lbf config is NULL");
    return 3 * (cfg->n_agents + cfg->n_food) + (cfg->observe_id ? cfg->n_agents : 0);
}

static int check_buffers(const marlhip_lbf_buffers* b) {
    MARL_REQUIRE(b && b->state && b->episode && b->ep_return && b->ep_length, "lbf buffers: NULL pointer");
    return 0;
}

extern "C" int marlhip_lbf_reset(const marlhip_lbf_config* cfg, const marlhip_lbf_buffers* buf, const uint8_t* mask, float* obs,
                                 void* stream) {
    if (lbf_validate(cfg) != 0 || check_buffers(buf) != 0) return -1;
    const LbfParams q = to_params(cfg);
    const int grid = (cfg->n_envs + ENV_BLOCK - 1) / ENV_BLOCK;
    const size_t lds = (size_t)ENV_BLOCK * marlhip_lbf_obs_dim(cfg) * sizeof(float);
#define X(p, f)                                                                                                     \
    if (cfg->n_agents == p && cfg->n_food == f)                                                                     \
        hipLaunchKernelGGL((lbf_reset_kernel<p, f>), dim3(grid), dim3(ENV_BLOCK), lds, (hipStream_t)stream, q, *buf, mask, obs);
    MARL_LBF_SHAPES(X)
#undef X
    MARL_CHECK_LAUNCH("lbf_reset");
    return 0;
}

extern "C" int marlhip_lbf_observe(const marlhip_lbf_config* cfg, const marlhip_lbf_buffers* buf, float* obs, void* stream) {
    if (lbf_validate(cfg) != 0 || check_buffers(buf) != 0) return -1;
    MARL_REQUIRE(obs != nullptr, "obs is NULL");
    const LbfParams q = to_params(cfg);
    const int grid = (cfg->n_envs + ENV_BLOCK - 1) / ENV_BLOCK;
    const size_t lds = (size_t)ENV_BLOCK * marlhip_lbf_obs_dim(cfg) * sizeof(float);
#define X(p, f)                                                                                                     \
    if (cfg->n_agents == p && cfg->n_food == f)                                                                     \
        hipLaunchKernelGGL((lbf_observe_kernel<p, f>), dim3(grid), dim3(ENV_BLOCK), lds, (hipStream_t)stream, q, *buf, obs);
    MARL_LBF_SHAPES(X)
#undef X
    MARL_CHECK_LAUNCH("lbf_observe");
    return 0;
}

extern "C" int marlhip_lbf_step(const marlhip_lbf_config* cfg, const marlhip_lbf_buffers* buf, const uint8_t* active,
                                const int32_t* actions, float* obs, float* rewards, uint8_t* done, uint8_t* truncated,
                                float* fin_return, int32_t* fin_length, int32_t auto_reset, float* final_obs, void* stream) {
    if (lbf_validate(cfg) != 0 || check_buffers(buf) != 0) return -1;
    MARL_REQUIRE(actions && obs && rewards && done && truncated && fin_return && fin_length, "lbf_step: NULL pointer");
    const LbfParams q = to_params(cfg);
    const int grid = (cfg->n_envs + ENV_BLOCK - 1) / ENV_BLOCK;
    const size_t lds = (size_t)ENV_BLOCK * marlhip_lbf_obs_dim(cfg) * sizeof(float);
    timing_begin(TIMER_ENVSTEP, (hipStream_t)stream);
#define X(p, f)                                                                                                        \
    if (cfg->n_agents == p && cfg->n_food == f)                                                                        \
        hipLaunchKernelGGL((lbf_step_kernel<p, f>), dim3(grid), dim3(ENV_BLOCK), lds, (hipStream_t)stream, q, *buf, active, \
                           actions, obs, rewards, done, truncated, fin_return, fin_length, (int)auto_reset, final_obs);
    MARL_LBF_SHAPES(X)
#undef X
    timing_end(TIMER_ENVSTEP, (hipStream_t)stream);
    MARL_CHECK_LAUNCH("lbf_step");
    return 0;
}
