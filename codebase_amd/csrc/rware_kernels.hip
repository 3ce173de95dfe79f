// Batched multi-robot warehouse kernels: reset / observe / step over N env records in HBM (rware_core.h).
// One lane = one env; same organisation as lbf_kernels.hip: observations are built per lane, staged through LDS and
// written as one contiguous run per (agent, workgroup).  HBM-bound byte work: algorithmic bytes per env-step =
// 2 * (7P + 4) agent/queue/counter bytes + the grid cells touched (<= 9 P reads, <= 2 P writes) + 4 P (actions)
// + 4 P D (obs) + 4 P (rewards) + 2.
#include "common.h"

namespace marl {

constexpr int RW_BLOCK = 256;

// (tried: staging half a row at a time to double the resident waves per CU - 36 KB instead of 73 KB of LDS per workgroup - is
// slower, 598 vs 438 us at 2^20 envs: shorter output runs and twice the barriers cost more than the occupancy gives)
template <int P>
__device__ __forceinline__ void rw_write_obs_tile(const RwParams& q, const RwState<P>& s, const RwGrid& grid, bool valid, float* tile,
                                                  float* __restrict__ obs, int n0, int cnt) {
    const int idw = q.observe_id ? P : 0, D = RW_OBS_DIM + idw;
    const int tid = threadIdx.x;
    RwRequested<P> rq;
    if (valid) rq.build(q, s);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (valid) {
            const uint64_t word = rw_window_word(q, s, grid, rq, p);
            for (int d = 0; d < idw; ++d) tile[tid * D + d] = d == p ? 1.f : 0.f;
#pragma unroll
            for (int d = 0; d < RW_OBS_DIM; ++d) tile[tid * D + idw + d] = rw_obs_elem_word(q, s, p, word, d);
        }
        __syncthreads();
        float* dst = obs + ((size_t)p * q.n_envs + n0) * D;
        for (int i = tid; i < cnt * D; i += RW_BLOCK) dst[i] = tile[i];
        __syncthreads();
    }
}

template <int P>
__global__ __launch_bounds__(RW_BLOCK) void rw_reset_kernel(RwParams q, marlhip_lbf_buffers b, const uint8_t* __restrict__ mask,
                                                            float* __restrict__ obs) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int n0 = blockIdx.x * RW_BLOCK, n = n0 + threadIdx.x;
    const int cnt = min(RW_BLOCK, q.n_envs - n0), cells = q.rows * q.cols;
    const int stride = rw_state_stride(P, q.rows, q.cols);
    const bool valid = n < q.n_envs;
    RwState<P> s;
    uint8_t* rec = b.state + (size_t)(valid ? n : 0) * stride;
    const RwGrid grid{rec, 1};
    if (valid) {
        if (mask == nullptr || mask[n]) {
            const uint32_t epi = b.episode[n];
            b.episode[n] = epi + 1;
            DrawStream rng;
            rng.init(q.seed, (uint32_t)n, epi, STREAM_RESET);
            rw_reset(q, s, grid, rng);
            rw_store(rec, cells, s);
#pragma unroll
            for (int p = 0; p < P; ++p) b.ep_return[(size_t)p * q.n_envs + n] = 0.f;
            b.ep_length[n] = 0;
        } else {
            rw_load(rec, cells, s);
        }
    }
    if (obs != nullptr) rw_write_obs_tile<P>(q, s, grid, valid, tile, obs, n0, cnt);
}

template <int P>
__global__ __launch_bounds__(RW_BLOCK) void rw_observe_kernel(RwParams q, marlhip_lbf_buffers b, float* __restrict__ obs) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int n0 = blockIdx.x * RW_BLOCK, n = n0 + threadIdx.x;
    const int cnt = min(RW_BLOCK, q.n_envs - n0), cells = q.rows * q.cols;
    const bool valid = n < q.n_envs;
    RwState<P> s;
    uint8_t* rec = b.state + (size_t)(valid ? n : 0) * rw_state_stride(P, q.rows, q.cols);
    const RwGrid grid{rec, 1};
    if (valid) rw_load(rec, cells, s);
    rw_write_obs_tile<P>(q, s, grid, valid, tile, obs, n0, cnt);
}

template <int P>
__global__ __launch_bounds__(RW_BLOCK) void rw_step_kernel(RwParams q, marlhip_lbf_buffers b, const uint8_t* __restrict__ active,
                                                           const int32_t* __restrict__ actions, float* __restrict__ obs,
                                                           float* __restrict__ rewards, uint8_t* __restrict__ done_out,
                                                           uint8_t* __restrict__ trunc_out, float* __restrict__ fin_return,
                                                           int32_t* __restrict__ fin_length, int auto_reset,
                                                           float* __restrict__ final_obs) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int n0 = blockIdx.x * RW_BLOCK, n = n0 + threadIdx.x;
    const int cnt = min(RW_BLOCK, q.n_envs - n0), cells = q.rows * q.cols;
    const int stride = rw_state_stride(P, q.rows, q.cols);
    const bool valid = n < q.n_envs;
    RwState<P> s;
    uint8_t* rec = b.state + (size_t)(valid ? n : 0) * stride;
    const RwGrid grid{rec, 1};
    if (valid) {
        rw_load(rec, cells, s);
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
            DrawStream req;
            req.init(q.seed, (uint32_t)n, b.episode[n] - 1u, STREAM_REQUEST);  // the running episode's stream
            rw_step(q, s, grid, a, raw, done, req);
            trunc = q.time_limit > 0 && s.steps >= q.time_limit;  // gymnasium TimeLimit
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
                        const int idw = q.observe_id ? P : 0;
                        RwRequested<P> frq;
                        frq.build(q, s);
#pragma unroll
                        for (int p = 0; p < P; ++p) {
                            const uint64_t word = rw_window_word(q, s, grid, frq, p);
                            float* fo = final_obs + ((size_t)p * q.n_envs + n) * (RW_OBS_DIM + idw);
                            for (int d = 0; d < idw; ++d) fo[d] = d == p ? 1.f : 0.f;
                            for (int d = 0; d < RW_OBS_DIM; ++d) fo[idw + d] = rw_obs_elem_word(q, s, p, word, d);
                        }
                    }
                    const uint32_t epi = b.episode[n];
                    b.episode[n] = epi + 1;
                    DrawStream rng;
                    rng.init(q.seed, (uint32_t)n, epi, STREAM_RESET);
                    rw_reset(q, s, grid, rng);
#pragma unroll
                    for (int p = 0; p < P; ++p) b.ep_return[(size_t)p * q.n_envs + n] = 0.f;
                    b.ep_length[n] = 0;
                }
            }
            rw_store(rec, cells, s);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) rewards[(size_t)p * q.n_envs + n] = rw[p];
        done_out[n] = done ? 1 : 0;
        trunc_out[n] = trunc ? 1 : 0;
    }
    rw_write_obs_tile<P>(q, s, grid, valid, tile, obs, n0, cnt);
}

}  // namespace marl

using namespace marl;

extern "C" int marlhip_rware_state_stride(const marlhip_rware_config* cfg) {
    if (rw_validate(cfg) != 0) return -1;
    const RwParams q = to_rw_params(cfg);
    return rw_state_stride(cfg->n_agents, q.rows, q.cols);
}

extern "C" int marlhip_rware_obs_dim(const marlhip_rware_config* cfg) {
    MARL_REQUIRE(cfg != nullptr, "rware config is NULL");
    return RW_OBS_DIM + (cfg->observe_id ? cfg->n_agents : 0);
}

extern "C" int marlhip_rware_grid(const marlhip_rware_config* cfg, int32_t* rows, int32_t* cols, int32_t* n_shelves) {
    if (rw_validate(cfg) != 0) return -1;
    const RwParams q = to_rw_params(cfg);
    if (rows) *rows = q.rows;
    if (cols) *cols = q.cols;
    if (n_shelves) *n_shelves = q.n_shelves;
    return 0;
}

static int rw_check_buffers(const marlhip_lbf_buffers* b) {
    MARL_REQUIRE(b && b->state && b->episode && b->ep_return && b->ep_length, "rware buffers: NULL pointer");
    return 0;
}

extern "C" int marlhip_rware_reset(const marlhip_rware_config* cfg, const marlhip_lbf_buffers* buf, const uint8_t* mask, float* obs,
                                   void* stream) {
    if (rw_validate(cfg) != 0 || rw_check_buffers(buf) != 0) return -1;
    const RwParams q = to_rw_params(cfg);
    const int grid = (cfg->n_envs + RW_BLOCK - 1) / RW_BLOCK;
    const size_t lds = (size_t)RW_BLOCK * marlhip_rware_obs_dim(cfg) * sizeof(float);
#define X(p)                                                                                                        \
    if (cfg->n_agents == p) {                                                                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rw_reset_kernel<p>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((rw_reset_kernel<p>), dim3(grid), dim3(RW_BLOCK), lds, (hipStream_t)stream, q, *buf, mask, obs);                \
    }
    MARL_RW_SHAPES(X)
#undef X
    MARL_CHECK_LAUNCH("rware_reset");
    return 0;
}

extern "C" int marlhip_rware_observe(const marlhip_rware_config* cfg, const marlhip_lbf_buffers* buf, float* obs, void* stream) {
    if (rw_validate(cfg) != 0 || rw_check_buffers(buf) != 0) return -1;
    MARL_REQUIRE(obs != nullptr, "obs is NULL");
    const RwParams q = to_rw_params(cfg);
    const int grid = (cfg->n_envs + RW_BLOCK - 1) / RW_BLOCK;
    const size_t lds = (size_t)RW_BLOCK * marlhip_rware_obs_dim(cfg) * sizeof(float);
#define X(p)                                                                                                        \
    if (cfg->n_agents == p) {                                                                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rw_observe_kernel<p>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((rw_observe_kernel<p>), dim3(grid), dim3(RW_BLOCK), lds, (hipStream_t)stream, q, *buf, obs);                      \
    }
    MARL_RW_SHAPES(X)
#undef X
    MARL_CHECK_LAUNCH("rware_observe");
    return 0;
}

extern "C" int marlhip_rware_step(const marlhip_rware_config* cfg, const marlhip_lbf_buffers* buf, const uint8_t* active,
                                  const int32_t* actions, float* obs, float* rewards, uint8_t* done, uint8_t* truncated,
                                  float* fin_return, int32_t* fin_length, int32_t auto_reset, float* final_obs, void* stream) {
    if (rw_validate(cfg) != 0 || rw_check_buffers(buf) != 0) return -1;
    MARL_REQUIRE(actions && obs && rewards && done && truncated && fin_return && fin_length, "rware_step: NULL pointer");
    const RwParams q = to_rw_params(cfg);
    const int grid = (cfg->n_envs + RW_BLOCK - 1) / RW_BLOCK;
    const size_t lds = (size_t)RW_BLOCK * marlhip_rware_obs_dim(cfg) * sizeof(float);
    timing_begin(TIMER_ENVSTEP, (hipStream_t)stream);
#define X(p)                                                                                                           \
    if (cfg->n_agents == p) {                                                                                          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rw_step_kernel<p>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((rw_step_kernel<p>), dim3(grid), dim3(RW_BLOCK), lds, (hipStream_t)stream, q, *buf, active, actions, obs,      \
                           rewards, done, truncated, fin_return, fin_length, (int)auto_reset, final_obs);                                   \
    }
    MARL_RW_SHAPES(X)
#undef X
    timing_end(TIMER_ENVSTEP, (hipStream_t)stream);
    MARL_CHECK_LAUNCH("rware_step");
    return 0;
}
