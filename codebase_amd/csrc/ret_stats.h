// standardise_returns: RunningMeanStd (marlbase/utils/standardise_stream.py:6-41) kept on the device.
// mean / var are fp32 [P] like the reference's tensors, count is the python float (fp64 here); the parallel-variance
// update runs in fp64 and is rounded once per update (the reference rounds every intermediate to fp32).
#pragma once
#include "common.h"

namespace marl {

struct RetStats {
    float* mean;    // [P]   (per-column statistics: [B])
    float* var;     // [P]   (per-column statistics: [B])
    double* count;  // [1]
    int columns;    // 0: one (mean, var) per agent (IDQN); B > 0: one per batch column (VDN / QMIX, see colstd_returns_kernel)
    // data-parallel training (marlhip_ret_stats): batch moments summed over the ranks before the running update; nullptr = one process
    marlhip_exchange_f64_fn exchange = nullptr;
    void* exchange_ctx = nullptr;
    double* moments = nullptr;  // [2P + 1]
};

inline void ret_stats_fill(RetStats& r, const marlhip_ret_stats* s) {
    r.mean = s->mean; r.var = s->var; r.count = s->count; r.columns = s->columns;
    r.exchange = s->exchange; r.exchange_ctx = s->exchange_ctx; r.moments = s->moments;
}

// per-block sums of x and x^2 over a block's (row, agent) values: fixed-order wave butterfly + 4 waves
__device__ __forceinline__ void ret_block_partials(float x, float* sh /*[2][4]*/, float* out2 /*[2]*/) {
    float s = x, q = x * x;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
    }
    __syncthreads();  // sh may still be read from the previous agent
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = s;
        sh[4 + (threadIdx.x >> 6)] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        out2[1] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
    }
}

// RunningMeanStd.update: batch mean / unbiased variance of the n values of every agent from the block partials
// partial[blk][p][2], then update_from_moments.  One workgroup of 64 threads, thread p owns agent p.
static __global__ __launch_bounds__(64) void ret_stats_update_kernel(RetStats st, const float* __restrict__ partial, int nblk, int P,
                                                                     int n) {
    const int p = threadIdx.x;
    const double count = st.count[0];
    if (p < P) {
        double s = 0.0, q = 0.0;
        for (int b = 0; b < nblk; ++b) {
            s += (double)partial[((size_t)b * P + p) * 2];
            q += (double)partial[((size_t)b * P + p) * 2 + 1];
        }
        const double bc = (double)n, bm = s / bc;
        const double bv = n > 1 ? (q - s * s / bc) / (bc - 1.0) : 0.0;
        const double mean = (double)st.mean[p], var = (double)st.var[p];
        const double delta = bm - mean, tot = count + bc;
        const double m2 = var * count + bv * bc + delta * delta * count * bc / tot;
        st.mean[p] = (float)(mean + delta * bc / tot);
        st.var[p] = (float)(m2 / tot);
    }
    __syncthreads();
    if (p == 0) st.count[0] = count + (double)n;
}

// the same update in two halves around the ranks' exchange: fold the local block partials into moments = {s_p, q_p}[P], n (fp64) ...
static __global__ __launch_bounds__(64) void ret_moments_kernel(const float* __restrict__ partial, int nblk, int P, int n,
                                                                double* __restrict__ moments) {
    const int p = threadIdx.x;
    if (p < P) {
        double s = 0.0, q = 0.0;
        for (int b = 0; b < nblk; ++b) {
            s += (double)partial[((size_t)b * P + p) * 2];
            q += (double)partial[((size_t)b * P + p) * 2 + 1];
        }
        moments[2 * p] = s;
        moments[2 * p + 1] = q;
    }
    if (p == 0) moments[2 * P] = (double)n;
}

// ... and update_from_moments with the moments of the GLOBAL batch (sums over the ranks): mean / unbiased variance of all
// n_total values, exactly what RunningMeanStd.update computes on the concatenated batch (standardise_stream.py:15-20)
static __global__ __launch_bounds__(64) void ret_stats_update_global_kernel(RetStats st, const double* __restrict__ moments, int P) {
    const int p = threadIdx.x;
    const double count = st.count[0];
    const double bc = moments[2 * P];
    if (p < P) {
        const double s = moments[2 * p], q = moments[2 * p + 1];
        const double bm = s / bc;
        const double bv = bc > 1.0 ? (q - s * s / bc) / (bc - 1.0) : 0.0;
        const double mean = (double)st.mean[p], var = (double)st.var[p];
        const double delta = bm - mean, tot = count + bc;
        const double m2 = var * count + bv * bc + delta * delta * count * bc / tot;
        st.mean[p] = (float)(mean + delta * bc / tot);
        st.var[p] = (float)(m2 / tot);
    }
    __syncthreads();
    if (p == 0) st.count[0] = count + bc;
}

// RunningMeanStd.update of the per-agent statistics from the block partials: one launch, or (data-parallel) fold -> exchange -> update
inline int launch_stats_update(const RetStats& st, const float* partial, int nblk, int P, int n, hipStream_t stream) {
    if (st.exchange == nullptr) {
        hipLaunchKernelGGL(ret_stats_update_kernel, dim3(1), dim3(64), 0, stream, st, partial, nblk, P, n);
        return 0;
    }
    MARL_REQUIRE(st.moments != nullptr && P <= 64, "standardise_returns: the exchange needs `moments` (2P + 1 doubles on the device)");
    hipLaunchKernelGGL(ret_moments_kernel, dim3(1), dim3(64), 0, stream, partial, nblk, P, n, st.moments);
    const int rc = st.exchange(st.exchange_ctx, st.moments, (int64_t)(2 * P + 1), (void*)stream);
    MARL_REQUIRE(rc == 0, "standardise_returns: the moments exchange callback failed (%d)", rc);
    hipLaunchKernelGGL(ret_stats_update_global_kernel, dim3(1), dim3(64), 0, stream, st, (const double*)st.moments, P);
    return 0;
}

// IDQN with standardised returns (QNetwork._compute_loss, dqn/model.py:146-163), between the agent-forward and
// agent-backward passes: returns from de-standardised bootstrap values (kept in the dq plane), statistics over ALL T*B
// entries of every agent, then dq_p = 2 filled (chosen_p - standardised return_p).
static __global__ __launch_bounds__(256) void std_returns_kernel(int P, int n, float gamma, RetStats st, const float* __restrict__ tqsel,
                                                                 const float* __restrict__ rew, const float* __restrict__ dn,
                                                                 float* __restrict__ ret, float* __restrict__ partial) {
    __shared__ float sh[8];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool in = i < n;
    const float nd = in ? 1.f - dn[i] : 0.f;
    for (int p = 0; p < P; ++p) {
        float r = 0.f;
        if (in) {
            const float tq = tqsel[(size_t)p * n + i] * sqrtf(st.var[p]) + st.mean[p];
            r = rew[(size_t)p * n + i] + gamma * tq * nd;
            ret[(size_t)p * n + i] = r;
        }
        ret_block_partials(r, sh, partial + ((size_t)blockIdx.x * P + p) * 2);
    }
}

static __global__ __launch_bounds__(256) void std_dq_kernel(int P, int n, RetStats st, const float* __restrict__ chosen,
                                                            const float* __restrict__ fl, float* __restrict__ ret_dq,
                                                            float* __restrict__ lrow) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float f = fl[i];
    float l = 0.f;
    for (int p = 0; p < P; ++p) {
        const float r = (ret_dq[(size_t)p * n + i] - st.mean[p]) / sqrtf(st.var[p]);
        const float delta = chosen[(size_t)p * n + i] - r;
        ret_dq[(size_t)p * n + i] = 2.f * f * delta;
        l += delta * delta;
    }
    lrow[i] = f * l;
}

// VDN / QMIX with standardise_returns (dqn/model.py:256-264, 415-422).  Their `RunningMeanStd(shape=(1,))` is fed the [T, B] returns:
// `update` flattens to [T, B], takes mean / unbiased variance over dim 0 (the T time steps) and broadcasts the (1,)-shaped state
// against the [B] moments - from the first update on the state IS [B]: one running (mean, var) per BATCH COLUMN, count += T per
// update.  Reproduced as the reference computes it: one thread per column b,
//   tq_tot[t] = sum of the `n_planes` planes (VDN: the agents' bootstrap values; QMIX: the target mixer's output)
//   ret[t] = r0[t] + gamma * (tq_tot[t] * sqrt(var[b]) + mean[b]) * (1 - done[t])          with the OLD statistics
//   update_from_moments(mean_t ret, var_t ret (unbiased), T) in the reference's fp32 operation order (the moments themselves are
//   accumulated in fp64 and rounded once), then out[t] = (ret[t] - mean[b]) / sqrt(var[b])  with the NEW statistics.
// `out` may alias the single plane (QMIX transforms the target mixer's output in place).
static __global__ __launch_bounds__(256) void colstd_returns_kernel(int T, int B, float gamma, RetStats st, const float* __restrict__ tq,
                                                                    int n_planes, size_t plane_stride, const float* __restrict__ r0,
                                                                    const float* __restrict__ dn, float* out) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float mean = st.mean[b], sd = sqrtf(st.var[b]), var = st.var[b];
    const float cnt = (float)st.count[0];  // the python float meets fp32 tensors: every product / quotient below is fp32
    auto ret_at = [&](int t) {
        const size_t i = (size_t)t * B + b;
        float tot = 0.f;
        for (int p = 0; p < n_planes; ++p) tot += tq[(size_t)p * plane_stride + i];
        return r0[i] + gamma * (tot * sd + mean) * (1.f - dn[i]);
    };
    double s = 0.0;
    for (int t = 0; t < T; ++t) s += (double)ret_at(t);
    const double bm64 = s / (double)T;
    double q = 0.0;
    for (int t = 0; t < T; ++t) {
        const double d = (double)ret_at(t) - bm64;
        q += d * d;
    }
    const float bm = (float)bm64, bv = (float)(q / (double)(T - 1)), bc = (float)T;
    const float delta = bm - mean;
    const float tot_f = (float)(st.count[0] + (double)T);
    const float new_mean = mean + delta * bc / tot_f;
    const float m_a = var * cnt, m_b = bv * bc;
    const float m_2 = m_a + m_b + delta * delta * cnt * bc / tot_f;
    const float new_var = m_2 / tot_f;
    const float nsd = sqrtf(new_var);
    for (int t = 0; t < T; ++t) {
        const float r = ret_at(t);
        out[(size_t)t * B + b] = (r - new_mean) / nsd;
    }
    st.mean[b] = new_mean;
    st.var[b] = new_var;
}

static __global__ void ret_count_add_kernel(double* count, double n) { count[0] += n; }

inline int launch_colstd(int T, int B, float gamma, const RetStats& st, const float* tq, int n_planes, size_t plane_stride, const float* r0,
                         const float* dn, float* out, hipStream_t stream) {
    MARL_REQUIRE(st.columns == B, "standardise_returns: the statistics hold %d batch columns, the batch has %d (the reference's shapes "
                 "pin the batch size after the first update)", st.columns, B);
    hipLaunchKernelGGL(colstd_returns_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, T, B, gamma, st, tq, n_planes, plane_stride, r0, dn,
                       out);
    hipLaunchKernelGGL(ret_count_add_kernel, dim3(1), dim3(1), 0, stream, st.count, (double)T);
    MARL_CHECK_LAUNCH("standardise_returns (per batch column)");
    return 0;
}

// the three launches; `partial` needs 2 * P * ceil(n / 256) floats
inline int launch_std_mixer(int P, int n, float gamma, const RetStats& st, const float* chosen, const float* tqsel, const float* rew,
                            const float* dn, const float* fl, float* dq, float* lrow, float* partial, hipStream_t stream) {
    const int nblk = (n + 255) / 256;
    hipLaunchKernelGGL(std_returns_kernel, dim3(nblk), dim3(256), 0, stream, P, n, gamma, st, tqsel, rew, dn, dq, partial);
    if (launch_stats_update(st, partial, nblk, P, n, stream) != 0) return -1;
    hipLaunchKernelGGL(std_dq_kernel, dim3(nblk), dim3(256), 0, stream, P, n, st, chosen, fl, dq, lrow);
    MARL_CHECK_LAUNCH("standardise_returns mixer");
    return 0;
}

}  // namespace marl
