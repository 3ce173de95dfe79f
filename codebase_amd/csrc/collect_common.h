// helpers shared by the fused collectors (collect.hip, ac_collect.hip)
#pragma once
#include "common.h"
#include "mlp.h"

namespace marl {

__device__ __forceinline__ uint32_t act_noise_word(uint64_t seed, uint32_t env, uint32_t episode, uint32_t t, int widx) {
    U4 c;
    c.x = env; c.y = episode; c.z = t | ((uint32_t)(widx >> 2) << 16); c.w = STREAM_ACT;
    const U4 o = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int s = widx & 3;
    return s == 0 ? o.x : (s == 1 ? o.y : (s == 2 ? o.z : o.w));
}

// obs element selection: lane g of an env takes elements 4ks+g of the observation vector
template <int P, int F, int KS1>
__device__ __forceinline__ void pick_obs(const LbfObs<P, F>& o, int g, float (&x)[KS1]) {
    constexpr int D = 3 * (P + F);
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        const float e0 = (4 * ks + 0 < D) ? o.v[(4 * ks + 0 < D) ? 4 * ks + 0 : 0] : 0.f;
        const float e1 = (4 * ks + 1 < D) ? o.v[(4 * ks + 1 < D) ? 4 * ks + 1 : 0] : 0.f;
        const float e2 = (4 * ks + 2 < D) ? o.v[(4 * ks + 2 < D) ? 4 * ks + 2 : 0] : 0.f;
        const float e3 = (4 * ks + 3 < D) ? o.v[(4 * ks + 3 < D) ? 4 * ks + 3 : 0] : 0.f;
        x[ks] = g == 0 ? e0 : (g == 1 ? e1 : (g == 2 ? e2 : e3));
    }
}

// the same with the ObserveID prefix (utils/wrappers.py:97-103): element e < P is the one-hot agent index, then the observation
template <int P, int F, int KS1, bool OID>
__device__ __forceinline__ void pick_obs_id(const LbfObs<P, F>& o, int p, int g, float (&x)[KS1]) {
    if (!OID) {
        pick_obs<P, F, KS1>(o, g, x);
        return;
    }
    constexpr int D0 = 3 * (P + F), D = D0 + P;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        float e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = 4 * ks + c;
            e[c] = i < P ? (i == p ? 1.f : 0.f) : (i < D ? o.v[(i >= P && i < D) ? i - P : 0] : 0.f);
        }
        x[ks] = g == 0 ? e[0] : (g == 1 ? e[1] : (g == 2 ? e[2] : e[3]));
    }
}

// Pre-packed actor / critic weights for the collectors: one tiny kernel turns the canonical parameter blocks into the
// MFMA A-operand packs ([P][NFWD], L2-resident), workgroups then stage them with straight 16-byte copies - once when all
// agents fit in LDS, once per (step, agent) when they do not (hidden 128, > 1 agent).  The scratch is the caller's workspace
// (common.h: ScratchScope / collect_pack_scratch).

template <class S>
__global__ __launch_bounds__(256) void fwd_pack_kernel(const float* __restrict__ params, AgentMap am, float* __restrict__ packs) {
    const int p = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < S::NFWD) packs[(size_t)p * S::NFWD + idx] = mlp_fwd_pack_elem<S>(params + (size_t)am.net[p] * S::NPARAM, idx);
}

template <class S>
__device__ __forceinline__ void stage_packed(const float* __restrict__ pack, float* lds, int tid, int nthreads) {
    static_assert(S::NFWD % 4 == 0, "pack is a whole number of float4");
    const f4* src = reinterpret_cast<const f4*>(pack);
    f4* dst = reinterpret_cast<f4*>(lds);
    copy_f4_to_lds(src, dst, S::NFWD / 4, tid, nthreads);
}

// arr[p] for a run-time p without a dynamically indexed register array
template <int P, class T>
__device__ __forceinline__ T pick_agent(const T (&arr)[P], int p) {
    T v = arr[0];
#pragma unroll
    for (int o = 1; o < P; ++o) v = o == p ? arr[o] : v;
    return v;
}

// how a collector keeps P forward packs on chip
template <class S, int P, size_t ENV_LDS = 0>
struct PackPlan {
    static constexpr size_t LDS_CAP = 150u * 1024u - ENV_LDS;  // the env's own LDS sits behind the packs
    static constexpr bool RESIDENT = (size_t)P * S::NFWD * sizeof(float) <= LDS_CAP;                 // whole packs in LDS
    static constexpr bool A3REG = !RESIDENT && (size_t)P * S::NFWD_NOA3 * sizeof(float) <= LDS_CAP;   // output layer in registers
    static constexpr int STRIDE = RESIDENT ? S::NFWD : S::NFWD_NOA3;                                   // floats per resident agent
    static constexpr size_t LDS_BYTES = ((RESIDENT || A3REG) ? (size_t)P * STRIDE : (size_t)S::NFWD) * sizeof(float);
};

template <class S>
__device__ __forceinline__ void stage_packed_prefix(const float* __restrict__ pack, float* lds, int nfloat, int tid, int nthreads) {
    const f4* src = reinterpret_cast<const f4*>(pack);
    f4* dst = reinterpret_cast<f4*>(lds);
    for (int i = tid; i < nfloat / 4; i += nthreads) dst[i] = src[i];
}

// The same copy WITHOUT passing through registers or waiting: every wave requests its share of the 1 KB pieces with global_load_lds
// (LDS address = wave-uniform base + 16 * lane, as in gru.h) and goes on - the collectors reset their envs and build the first
// observation while the packs land; `stage_async_wait` (vmcnt(0) + workgroup barrier) sits in front of the first forward pass.
__device__ __forceinline__ void stage_packed_async(const float* __restrict__ pack, float* lds, int nfloat, int wave, int nwaves, int lane) {
    const char* src = reinterpret_cast<const char*>(pack) + 16 * lane;
    char* dst = reinterpret_cast<char*>(lds);
    const int bytes = nfloat * 4, full = bytes >> 10;
    for (int piece = wave; piece < full; piece += nwaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 1024 * piece),
                                         (__attribute__((address_space(3))) void*)(dst + 1024 * piece), 16, 0, 0);
    const int tail = bytes - (full << 10);  // a multiple of 16: the pack is whole float4s
    if (tail > 0 && wave == full % nwaves && 16 * lane < tail)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 1024 * full),
                                         (__attribute__((address_space(3))) void*)(dst + 1024 * full), 16, 0, 0);
}

__device__ __forceinline__ void stage_async_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

template <class S>
int launch_fwd_pack(int P, const AgentMap& am, const float* params, float** packs_out, hipStream_t st) {
    float* packs = collect_pack_scratch((size_t)P * S::NFWD * sizeof(float), st);
    if (packs == nullptr) return -1;  // error text set by collect_pack_scratch
    hipLaunchKernelGGL((fwd_pack_kernel<S>), dim3((S::NFWD + 255) / 256, P), dim3(256), 0, st, params, am, packs);
    MARL_CHECK_LAUNCH("fwd_pack_kernel");
    *packs_out = packs;
    return 0;
}

}  // namespace marl
