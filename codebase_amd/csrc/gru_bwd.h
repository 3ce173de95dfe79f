// Backward of the recurrent Q-networks (BPTT through gru_seq_fwd_kernel's activation record) and the TD error.
//   gru_td_kernel      : online q / target q of every observation -> dL/dq rows (dense [P][T+1][B][A], unnormalised) and
//                        the per-row loss (QNetwork._compute_loss dqn/model.py:118-163, VDNetwork :224-269)
//   gru_seq_bwd_kernel : a wave walks its 16 sequences BACKWARDS with dL/dh in registers: dh += W3^T dq, the gate
//                        derivatives of the GRU cell, dh_prev = dh z + W_hh^T dgh, dx1 = relu'(W_ih^T dgi); the per-step
//                        gate gradients (dr, dz, dn, r*dn, dx1) go to a second record
//   gru_wgrad_kernel   : the weight gradients are plain sums over all (t, b) rows of outer products of the two records
//                        (dW_ih = dgi^T x1, dW_hh = dgh^T h_prev, dW1 = dx1^T x, dW3 = dq^T h): one workgroup pass, the
//                        four waves own disjoint parameter slices (gate r / z / n, and first + output layer), operands
//                        transposed through LDS tiles as in the feed-forward kernels; one partial record per workgroup,
//                        summed in fixed order by dqn_reduce_kernel.
#pragma once
#include "td_rows.h"
#include "dqn_update_kernels.h"
#include "gru.h"

namespace marl {

template <class S>
struct GruBwd {
    static constexpr int MT = S::MT;
    // backward-data pack: T3[MT][64][4] | Thh[3][MT][MT][64][4] | Tih[3][MT][MT][64][4]
    static constexpr int pT3 = 0, pThh = pT3 + MT * 256, pTih = pThh + 3 * MT * MT * 256, NBWD = pTih + 3 * MT * MT * 256;
    static constexpr int REC2_ARRAYS = 5, REC2 = REC2_ARRAYS * MT * 256;  // dr, dz, dn, r * dn, dx1
    // hidden 128: T3 resident, the six transposed gate matrices stream through one chunk buffer (as in the forward kernel)
    static constexpr int CHUNK = MT * MT * 256;
    static constexpr bool STREAM = S::STREAM;
    static constexpr int LDS_FLOATS = STREAM ? pThh + CHUNK : NBWD;
};

template <class S>
__device__ __forceinline__ float gru_bwd_pack_elem(const float* __restrict__ w, int idx) {
    using Bk = GruBwd<S>;
    if (idx < Bk::pThh) {  // T3[mt][lane][r]: A[i = unit 16mt+i][k = a 4g+r] = W3[4g+r][16mt+i]
        const int r = idx & 3, lane = (idx >> 2) & 63, mt = idx >> 8;
        const int a = 4 * (lane >> 4) + r;
        return a < S::A ? w[S::oW3 + a * S::H + 16 * mt + (lane & 15)] : 0.f;
    }
    // T??[gate][mt1][mt2][lane][r]: A[i = input unit 16mt1+i][k = gate unit 16mt2+4g+r] = W[gate*H + 16mt2+4g+r][16mt1+i]
    const bool ih = idx >= Bk::pTih;
    const int j = idx - (ih ? Bk::pTih : Bk::pThh);
    const int r = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
    const int mt2 = rest % S::MT, mt1 = (rest / S::MT) % S::MT, gate = rest / (S::MT * S::MT);
    return w[(ih ? S::oWih : S::oWhh) + (gate * S::H + 16 * mt2 + 4 * (lane >> 4) + r) * S::H + 16 * mt1 + (lane & 15)];
}

template <class S>
__global__ __launch_bounds__(256) void gru_bwd_pack_kernel(const float* __restrict__ params, AgentMap am, float* __restrict__ packs,
                                                           int block_floats = S::NPARAM, int layer_off = 0) {
    const int p = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < GruBwd<S>::NBWD) packs[(size_t)p * GruBwd<S>::NBWD + idx] = gru_bwd_pack_elem<S>(params + (size_t)am.net[p] * block_floats + layer_off, idx);
}

// transposed product: out[mt1] += sum_{gate-unit tiles mt2, r} T[gate][mt1][mt2][lane][r] * dg[mt2][r]
template <class S>
__device__ __forceinline__ void gru_tgate(const f4* Tm /* [MT][MT][64] chunk */, int lane, const f4 (&dg)[S::MT], f4 (&out)[S::MT]) {
    constexpr int MT = S::MT;
#pragma unroll
    for (int m2 = 0; m2 < MT; ++m2) {
        f4 a[MT];
#pragma unroll
        for (int m1 = 0; m1 < MT; ++m1) a[m1] = Tm[(m1 * MT + m2) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m1 = 0; m1 < MT; ++m1) out[m1] = MARL_MFMA(a[m1][r], dg[m2][r], out[m1]);
    }
}

// rec: forward record [P][S][nblk][REC]; dq: [P][S][B][A]; rec2: [P][S][nblk][REC2]
// Stacked layers: dh_above = the backward record of the layer above - dL/dh[t] of this layer is that layer's dx1[t] instead of W3^T dq[t];
// inner (layer >= 1): this layer's input is the hidden state below, not a ReLU output - dx1 is not masked
template <class S>
__global__ __launch_bounds__(256) void gru_seq_bwd_kernel(const float* __restrict__ packs, int steps, int B, const float* __restrict__ rec,
                                                          const float* __restrict__ dq, float* __restrict__ rec2,
                                                          const float* __restrict__ dh_above = nullptr, int inner = 0) {
    using Bk = GruBwd<S>;
    constexpr int MT = S::MT, A = S::A;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int p = blockIdx.y;
    constexpr bool STREAM = Bk::STREAM;
    const float* pack = packs + (size_t)p * Bk::NBWD;
    copy_f4_to_lds(reinterpret_cast<const f4*>(pack), reinterpret_cast<f4*>(lds), (STREAM ? Bk::pThh : Bk::NBWD) / 4, tid, 256);
    __syncthreads();
    const int nblk = (B + 15) >> 4;
    const int blk0 = blockIdx.x * 4 + wave;
    const bool active = blk0 < nblk;
    if (!STREAM && !active) return;  // streamed: every wave keeps staging and meeting the barriers
    const int blk = active ? blk0 : nblk - 1;
    const int b0 = blk * 16;
    const bool rowok = active && b0 + j < B;
    const int bj = b0 + j < B ? b0 + j : B - 1;
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // transposed gate matrix `c` (0..2: W_hh^T r, z, n; 3..5: W_ih^T r, z, n)
    auto tchunk = [&](int c) -> const f4* {
        if (!STREAM) return reinterpret_cast<const f4*>(lds + Bk::pThh) + (size_t)c * MT * MT * 64;
        __syncthreads();
        copy_f4_to_lds(reinterpret_cast<const f4*>(pack + Bk::pThh) + (size_t)c * MT * MT * 64, reinterpret_cast<f4*>(lds + Bk::pThh), Bk::CHUNK / 4,
                       tid, 256);
        __syncthreads();
        return reinterpret_cast<const f4*>(lds + Bk::pThh);
    };
    const f4* T3 = reinterpret_cast<const f4*>(lds + Bk::pT3);
    f4 carry[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) carry[mt] = zero4;
    for (int t = steps - 1; t >= 0; --t) {
        asm volatile("" ::: "memory");  // keep the weight reads inside the loop (see gru_seq_fwd_kernel)
        const f4* R = reinterpret_cast<const f4*>(rec + (((size_t)p * steps + t) * nblk + blk) * S::REC);
        const f4* Rp = reinterpret_cast<const f4*>(rec + (((size_t)p * steps + (t > 0 ? t - 1 : 0)) * nblk + blk) * S::REC);
        f4 x1[MT], rg[MT], zg[MT], ng[MT], ghn[MT], hp[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            x1[mt] = R[(0 * MT + mt) * 64 + lane];
            rg[mt] = R[(1 * MT + mt) * 64 + lane];
            zg[mt] = R[(2 * MT + mt) * 64 + lane];
            ng[mt] = R[(3 * MT + mt) * 64 + lane];
            ghn[mt] = R[(5 * MT + mt) * 64 + lane];
            hp[mt] = t > 0 ? Rp[(4 * MT + mt) * 64 + lane] : zero4;
        }
        f4 dh[MT];
        if (dh_above != nullptr) {  // dh = carried + the layer above's dx1
            const f4* Ra = reinterpret_cast<const f4*>(dh_above + (((size_t)p * steps + t) * nblk + blk) * Bk::REC2);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) dh[mt] = carry[mt] + Ra[(4 * MT + mt) * 64 + lane];
        } else {
        f4 dQ;
        {
            const float* drow = dq + (((size_t)p * steps + t) * B + bj) * A;
#pragma unroll
            for (int r = 0; r < 4; ++r) dQ[r] = (rowok && 4 * g + r < A) ? drow[4 * g + r < A ? 4 * g + r : A - 1] : 0.f;
        }
        // dh = carried + W3^T dq
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            dh[mt] = carry[mt];
            const f4 a = T3[mt * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) dh[mt] = MARL_MFMA(a[r], dQ[r], dh[mt]);
        }
        }
        // gate derivatives (h' = (1 - z) n + z h;  n = tanh(gi_n + r ghn);  r, z = sigmoid(...))
        f4 dr[MT], dz[MT], dng[MT], drn[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            dng[mt] = dh[mt] * (1.f - zg[mt]) * (1.f - ng[mt] * ng[mt]);
            dz[mt] = dh[mt] * (hp[mt] - ng[mt]) * zg[mt] * (1.f - zg[mt]);
            drn[mt] = dng[mt] * rg[mt];
            dr[mt] = dng[mt] * ghn[mt] * rg[mt] * (1.f - rg[mt]);
            carry[mt] = dh[mt] * zg[mt];
        }
        gru_tgate<S>(tchunk(0), lane, dr, carry);
        gru_tgate<S>(tchunk(1), lane, dz, carry);
        gru_tgate<S>(tchunk(2), lane, drn, carry);
        f4 dx1[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dx1[mt] = zero4;
        gru_tgate<S>(tchunk(3), lane, dr, dx1);
        gru_tgate<S>(tchunk(4), lane, dz, dx1);
        gru_tgate<S>(tchunk(5), lane, dng, dx1);
        if (!active) continue;
        f4* R2 = reinterpret_cast<f4*>(rec2 + (((size_t)p * steps + t) * nblk + blk) * Bk::REC2);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dx1[mt][r] = (inner || x1[mt][r] > 0.f) ? dx1[mt][r] : 0.f;
            R2[(0 * MT + mt) * 64 + lane] = dr[mt];
            R2[(1 * MT + mt) * 64 + lane] = dz[mt];
            R2[(2 * MT + mt) * 64 + lane] = dng[mt];
            R2[(3 * MT + mt) * 64 + lane] = drn[mt];
            R2[(4 * MT + mt) * 64 + lane] = dx1[mt];
        }
    }
}

// The same walk with TWO waves per block of 16 sequences: B / 16 blocks of a bench batch are 512 waves for 1024 SIMDs, and the
// walk is a chain of dependent steps, so the one-wave form leaves half of the chip idle.  Here each wave of a pair owns half of the hidden
// tiles: it loads the record, forms dh and the gate derivatives and keeps the carried dh for ITS tiles only, the pair swaps the four
// gate-gradient arrays through LDS (the transposed products need every gate unit as input), and each wave multiplies out only its own
// output tiles - half of the MFMAs, of the record traffic and of the gate arithmetic per wave and step, for two barriers.
// Hidden 128 streams the six transposed gate matrices through the chunk buffer exactly as the one-wave form does (the workgroup's two
// pairs share each staged matrix); the staging per workgroup and step is unchanged, the ~49 k cycles of MFMAs per wave and step halve.
template <class S>
struct GruBwdSplit : std::integral_constant<bool, S::MT == 4 || S::MT == 8> {};
template <class S>
constexpr size_t gru_bwd_lds_bytes() {
    return (size_t)(GruBwd<S>::LDS_FLOATS + (GruBwdSplit<S>::value ? 2 * 4 * S::MT * 256 : 0)) * sizeof(float);
}

template <class S>
__global__ __launch_bounds__(256) void gru_seq_bwd2_kernel(const float* __restrict__ packs, int steps, int B, const float* __restrict__ rec,
                                                           const float* __restrict__ dq, float* __restrict__ rec2,
                                                           const float* __restrict__ dh_above = nullptr, int inner = 0) {
    using Bk = GruBwd<S>;
    constexpr int MT = S::MT, MH = MT / 2, A = S::A;
    constexpr bool STREAM = Bk::STREAM;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int pairw = wave >> 1, half = wave & 1, m0 = half * MH;  // the pair's block inside the workgroup, this wave's tiles m0 .. m0 + MH - 1
    const int p = blockIdx.y;
    const float* pack = packs + (size_t)p * Bk::NBWD;
    copy_f4_to_lds(reinterpret_cast<const f4*>(pack), reinterpret_cast<f4*>(lds), (STREAM ? Bk::pThh : Bk::NBWD) / 4, tid, 256);
    __syncthreads();
    f4* X = reinterpret_cast<f4*>(lds + Bk::LDS_FLOATS) + (size_t)pairw * 4 * MT * 64;  // the pair's exchange: [dr, dz, dn, r * dn][unit tile][lane]
    const int nblk = (B + 15) >> 4;
    const int blk0 = blockIdx.x * 2 + pairw;
    const bool active = blk0 < nblk;  // (an idle pair keeps walking: the barriers are workgroup-wide)
    const int blk = active ? blk0 : nblk - 1;
    const int b0 = blk * 16;
    const bool rowok = active && b0 + j < B;
    const int bj = b0 + j < B ? b0 + j : B - 1;
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f4* T3 = reinterpret_cast<const f4*>(lds + Bk::pT3);
    auto tmat = [&](int c) -> const f4* {
        if (!STREAM) return reinterpret_cast<const f4*>(lds + Bk::pThh) + (size_t)c * MT * MT * 64;
        __syncthreads();  // (every wave of the workgroup walks every step: see `active`)
        copy_f4_to_lds(reinterpret_cast<const f4*>(pack + Bk::pThh) + (size_t)c * MT * MT * 64, reinterpret_cast<f4*>(lds + Bk::pThh), Bk::CHUNK / 4,
                       tid, 256);
        __syncthreads();
        return reinterpret_cast<const f4*>(lds + Bk::pThh);
    };
    // out[own tile] += (transposed gate matrix c) x dg, every gate-unit tile of dg as input
    auto tgate = [&](int c, const f4 (&dgx)[MT], f4 (&out)[MH]) {
        const f4* Tm = tmat(c);
#pragma unroll
        for (int m2 = 0; m2 < MT; ++m2) {
            f4 a[MH];
#pragma unroll
            for (int u = 0; u < MH; ++u) a[u] = Tm[((m0 + u) * MT + m2) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < MH; ++u) out[u] = MARL_MFMA(a[u][r], dgx[m2][r], out[u]);
        }
    };
    f4 carry[MH];
#pragma unroll
    for (int u = 0; u < MH; ++u) carry[u] = zero4;
    for (int t = steps - 1; t >= 0; --t) {
        asm volatile("" ::: "memory");  // keep the weight reads inside the loop (see gru_seq_fwd_kernel)
        const f4* R = reinterpret_cast<const f4*>(rec + (((size_t)p * steps + t) * nblk + blk) * S::REC);
        const f4* Rp = reinterpret_cast<const f4*>(rec + (((size_t)p * steps + (t > 0 ? t - 1 : 0)) * nblk + blk) * S::REC);
        f4 x1[MH], rg[MH], zg[MH], ng[MH], ghn[MH], hp[MH];
#pragma unroll
        for (int u = 0; u < MH; ++u) {
            const int mt = m0 + u;
            x1[u] = R[(0 * MT + mt) * 64 + lane];
            rg[u] = R[(1 * MT + mt) * 64 + lane];
            zg[u] = R[(2 * MT + mt) * 64 + lane];
            ng[u] = R[(3 * MT + mt) * 64 + lane];
            ghn[u] = R[(5 * MT + mt) * 64 + lane];
            hp[u] = t > 0 ? Rp[(4 * MT + mt) * 64 + lane] : zero4;
        }
        f4 dQ = zero4;
        const f4* Ra = nullptr;
        if (dh_above != nullptr) {
            Ra = reinterpret_cast<const f4*>(dh_above + (((size_t)p * steps + t) * nblk + blk) * Bk::REC2);
        } else {
            const float* drow = dq + (((size_t)p * steps + t) * B + bj) * A;
#pragma unroll
            for (int r = 0; r < 4; ++r) dQ[r] = (rowok && 4 * g + r < A) ? drow[4 * g + r < A ? 4 * g + r : A - 1] : 0.f;
        }
        f4 dr[MH], dz[MH], dng[MH], drn[MH];
#pragma unroll
        for (int u = 0; u < MH; ++u) {
            f4 dh = carry[u];  // dh = carried + W3^T dq (or + the layer above's dx1)
            if (Ra != nullptr) {
                dh += Ra[(4 * MT + m0 + u) * 64 + lane];
            } else {
            const f4 a = T3[(m0 + u) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) dh = MARL_MFMA(a[r], dQ[r], dh);
            }
            dng[u] = dh * (1.f - zg[u]) * (1.f - ng[u] * ng[u]);
            dz[u] = dh * (hp[u] - ng[u]) * zg[u] * (1.f - zg[u]);
            drn[u] = dng[u] * rg[u];
            dr[u] = dng[u] * ghn[u] * rg[u] * (1.f - rg[u]);
            carry[u] = dh * zg[u];
            X[(0 * MT + m0 + u) * 64 + lane] = dr[u];
            X[(1 * MT + m0 + u) * 64 + lane] = dz[u];
            X[(2 * MT + m0 + u) * 64 + lane] = dng[u];
            X[(3 * MT + m0 + u) * 64 + lane] = drn[u];
        }
        __syncthreads();
        f4 ar[MT], az[MT], an[MT], arn[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ar[mt] = X[(0 * MT + mt) * 64 + lane];
            az[mt] = X[(1 * MT + mt) * 64 + lane];
            an[mt] = X[(2 * MT + mt) * 64 + lane];
            arn[mt] = X[(3 * MT + mt) * 64 + lane];
        }
        __syncthreads();  // everybody has read: the next step may overwrite the exchange
        tgate(0, ar, carry);
        tgate(1, az, carry);
        tgate(2, arn, carry);
        f4 dx1[MH];
#pragma unroll
        for (int u = 0; u < MH; ++u) dx1[u] = zero4;
        tgate(3, ar, dx1);
        tgate(4, az, dx1);
        tgate(5, an, dx1);
        if (!active) continue;
        f4* R2 = reinterpret_cast<f4*>(rec2 + (((size_t)p * steps + t) * nblk + blk) * Bk::REC2);
#pragma unroll
        for (int u = 0; u < MH; ++u) {
            const int mt = m0 + u;
#pragma unroll
            for (int r = 0; r < 4; ++r) dx1[u][r] = (inner || x1[u][r] > 0.f) ? dx1[u][r] : 0.f;
            R2[(0 * MT + mt) * 64 + lane] = dr[u];
            R2[(1 * MT + mt) * 64 + lane] = dz[u];
            R2[(2 * MT + mt) * 64 + lane] = dng[u];
            R2[(3 * MT + mt) * 64 + lane] = drn[u];
            R2[(4 * MT + mt) * 64 + lane] = dx1[u];
        }
    }
}

// the backward walk of P agents over B sequences: the two-wave form where it exists (MARLHIP_GRU_BWD_ONE_WAVE=1 keeps the one-wave form).
// `alone` = no sibling launch shares the chip (the actor-critic learner walks actors and critics on two streams: with streamed
// matrices the pair of one-wave launches already fills the CUs and the two-wave form only adds staging - measured 7.54 vs 7.69 ms
// per recurrent IA2C GRU-128 round; alone, the hidden-128 walk goes 1045 -> 650 us).
template <class S>
void gru_launch_seq_bwd(int P, int B, const float* packB, int steps, const float* rec, const float* dq, float* rec2, hipStream_t st, bool alone = true,
                        const float* dh_above = nullptr, int inner = 0) {
    static const bool one_wave = getenv("MARLHIP_GRU_BWD_ONE_WAVE") != nullptr;
    if constexpr (GruBwdSplit<S>::value) {
        if (!one_wave && (alone || !GruBwd<S>::STREAM)) {
            static LdsAttr attr;
            if (attr.need()) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_bwd2_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)gru_bwd_lds_bytes<S>());
                attr.done();
            }
            hipLaunchKernelGGL((gru_seq_bwd2_kernel<S>), dim3((B + 31) / 32, P), dim3(256), gru_bwd_lds_bytes<S>(), st, packB, steps, B, rec, dq, rec2, dh_above, inner);
            return;
        }
    }
    hipLaunchKernelGGL((gru_seq_bwd_kernel<S>), dim3((B + 63) / 64, P), dim3(256), GruBwd<S>::LDS_FLOATS * sizeof(float), st, packB, steps, B, rec, dq, rec2, dh_above, inner);
}

// hidden 64 (four unit tiles): the three gate roles below are one workgroup role (the FUSED branch of the kernel); launches use
// gru_wgrad_roles / gru_wgrad_lds_bytes
template <class S>
struct GruWgradFused : std::integral_constant<bool, S::MT == 4> {};
template <class S>
constexpr int gru_wgrad_roles() { return GruWgradFused<S>::value ? 2 : 4; }
template <class S>
constexpr size_t gru_wgrad_lds_bytes() {
    return (GruWgradFused<S>::value ? (size_t)6 * S::H * 24 : (size_t)4 * 16 * S::H + 256) * sizeof(float);
}

// One partial record [NPARAM + 2] per blockIdx.x (canonical parameter order, then loss and n_filled from agent 0's rows), filled
// by four workgroups with disjoint roles (blockIdx.z): 0..2 = the rows of gate r / z / n of dW_ih and dW_hh, their four waves
// owning a quarter of the columns each; 3 = the small parts (wave 0: dW1, db1; wave 1: dW3, db3, loss; wave 2: bias sums of
// gates r, z; wave 3: bias sums of gate n).  Every role reads only the record arrays it needs.
template <class S>
__global__ __launch_bounds__(256) void gru_wgrad_kernel(int steps, int B, const float* __restrict__ obs, size_t obs_as, size_t obs_rs,
                                                        const float* __restrict__ rec,
                                                        const float* __restrict__ rec2, const float* __restrict__ dq,
                                                        const float* __restrict__ lrow, const float* __restrict__ filled, int loss_steps,
                                                        float* __restrict__ partials, int rec_floats = S::NPARAM + 2, int layer_off = 0,
                                                        int parts = 3) {
    // Stacked layers: rec_floats = the whole block's partial record (S::nparam(L) + 2), layer_off = l * S::LAYER (this layer's gate
    // gradients land at the block's oWih + layer_off ...), parts: bit 0 - this is layer 0 (dW1, db1), bit 1 - the last layer (dW3, db3, loss)
    using Bk = GruBwd<S>;
    constexpr int MT = S::MT, H = S::H, D = S::D, A = S::A, NT1 = S::DP / 16, TILE = 16 * H, MTN = MT / 4;
    extern __shared__ __attribute__((aligned(16))) float tiles[];  // 4 * TILE + 256 floats
    float* T0 = tiles;
    float* T1 = T0 + TILE;
    float* T2 = T1 + TILE;
    float* T3 = T2 + TILE;
    float* TQ = T3 + TILE;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    constexpr bool FUSED = GruWgradFused<S>::value;  // hidden 64: ONE workgroup role for the three gates (grid.z = 2)
    const int p = blockIdx.y, role = FUSED ? (blockIdx.z == 0 ? 0 : 3) : (int)blockIdx.z;
    const int nblk = (B + 15) >> 4, T = loss_steps;  // lrow / filled have loss_steps rows (DQN: steps - 1; actor-critic: steps)
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float* recd = partials + ((size_t)p * gridDim.x + blockIdx.x) * rec_floats + layer_off;
    const int total = steps * nblk;
    if constexpr (FUSED) {
        if (role == 0) {
            // ---- all three gates in one workgroup: x1 and h_{t-1} are read once instead of once per gate, the gate gradients once
            // instead of twice (their row sums = the gate biases are taken here, from the registers on their way to the tiles), and a
            // wave issues 96 MFMAs between two barriers instead of 32.  Six padded [unit][24] tiles: dr, dz, dn, r*dn, x1, h_{t-1};
            // the 24 (array, unit tile) pairs are transposed 6 per wave; the next item's loads fly during the MFMAs.
            constexpr int TS = 24, TL = H * TS;
            f4 dWa[3][MT], dWb[3][MT], bs[6], v[6];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) { dWa[a][b] = zero4; dWb[a][b] = zero4; }
#pragma unroll
            for (int k = 0; k < 6; ++k) bs[k] = zero4;
            auto load = [&](int item) {
                const int t = item / nblk, blk = item - t * nblk;
                const f4* R = reinterpret_cast<const f4*>(rec + (((size_t)p * steps + t) * nblk + blk) * S::REC);
                const f4* Rp = reinterpret_cast<const f4*>(rec + (((size_t)p * steps + (t > 0 ? t - 1 : 0)) * nblk + blk) * S::REC);
                const f4* R2 = reinterpret_cast<const f4*>(rec2 + (((size_t)p * steps + t) * nblk + blk) * Bk::REC2);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int u = wave * 6 + k, arr = u >> 2, mt = u & 3;
                    if (arr < 4) v[k] = R2[(arr * MT + mt) * 64 + lane];
                    else if (arr == 4) v[k] = R[(0 * MT + mt) * 64 + lane];
                    else v[k] = t > 0 ? Rp[(4 * MT + mt) * 64 + lane] : zero4;
                }
            };
            if ((int)blockIdx.x < total) load(blockIdx.x);
            for (int item = blockIdx.x; item < total; item += gridDim.x) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int u = wave * 6 + k, arr = u >> 2, mt = u & 3;
                    bs[k] += v[k];
#pragma unroll
                    for (int r = 0; r < 4; ++r) tiles[arr * TL + (16 * mt + 4 * g + r) * TS + j] = v[k][r];
                }
                __syncthreads();
                if (item + (int)gridDim.x < total) load(item + gridDim.x);
                const f4 bX = tile_read_s<TS>(tiles + 4 * TL, wave, g, j), bH = tile_read_s<TS>(tiles + 5 * TL, wave, g, j);
#pragma unroll
                for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f4 aI = tile_read_s<TS>(tiles + gate * TL, mt, g, j);
                        const f4 aH = gate == 2 ? tile_read_s<TS>(tiles + 3 * TL, mt, g, j) : aI;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            dWa[gate][mt] = MARL_MFMA(aI[ks], bX[ks], dWa[gate][mt]);
                            dWb[gate][mt] = MARL_MFMA(aH[ks], bH[ks], dWb[gate][mt]);
                        }
                    }
                __syncthreads();  // tiles are rewritten by the next item
            }
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int vv = gate * H + 16 * mt + 4 * g + r;
                        recd[S::oWih + vv * H + 16 * wave + j] = dWa[gate][mt][r];
                        recd[S::oWhh + vv * H + 16 * wave + j] = dWb[gate][mt][r];
                    }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int u = wave * 6 + k, arr = u >> 2, mt = u & 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sv = sum16(bs[k][r]);
                    const int vv = 16 * mt + 4 * g + r;
                    if (j == 0 && arr < 4) {
                        if (arr < 2) { recd[S::obih + arr * H + vv] = sv; recd[S::obhh + arr * H + vv] = sv; }  // gates r, z: dgi == dgh
                        else if (arr == 2) recd[S::obih + 2 * H + vv] = sv;                                         // dn
                        else recd[S::obhh + 2 * H + vv] = sv;                                                       // r * dn
                    }
                }
            }
            return;
        }
        if (wave >= 2) return;  // the gate biases were summed by the gate workgroup
    }
    if (role < 3) {
        // ---- gate `role`: dW_ih[gate rows][my columns] += dgi^T x1, dW_hh[...] += dgh^T h_prev
        // The record loads go through ONE wave-uniform source pointer per item as eight unconditional loads.  The per-tile
        // `if (wave == 0) ... else if ...` form of rounds 1-2 compiled to a branch around every load with `s_waitcnt vmcnt(0)` in front
        // of the next one - eight dependent memory round trips per item - and ran hidden-128 recurrent IDQN at 1.19 M env-steps/s
        // against 1.41 M with this form (gru_wgrad 1054 -> ~640 us per update; profiles/r03_flatload_ab.md).
        const int nt0 = wave * MTN;
        f4 dWa[MT][MTN], dWb[MT][MTN];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < MTN; ++b) { dWa[a][b] = zero4; dWb[a][b] = zero4; }
        const int gi_arr = role, gh_arr = role == 2 ? 3 : role;  // rec2 arrays: dr, dz, dn, r * dn
        // the four waves transpose one array each into the shared tiles ([unit][16 rows]): wave 0 dgi, 1 dgh, 2 x1, 3 h_{t-1}
        auto fetch = [&](int item, f4 (&v)[MT]) {
            const int t = item / nblk, blk = item - t * nblk;
            const f4* R = reinterpret_cast<const f4*>(rec + (((size_t)p * steps + t) * nblk + blk) * S::REC);
            const f4* Rp = reinterpret_cast<const f4*>(rec + (((size_t)p * steps + (t > 0 ? t - 1 : 0)) * nblk + blk) * S::REC);
            const f4* R2 = reinterpret_cast<const f4*>(rec2 + (((size_t)p * steps + t) * nblk + blk) * Bk::REC2);
            const f4* src = wave == 0 ? R2 + gi_arr * MT * 64 : (wave == 1 ? R2 + gh_arr * MT * 64 : (wave == 2 ? R : Rp + 4 * MT * 64));
            const bool none = wave == 3 && t == 0;  // h_{-1} = 0 (the row read instead is step 0's own: a valid address)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) v[mt] = src[mt * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) v[mt] = none ? zero4 : v[mt];
        };
        for (int item = blockIdx.x; item < total; item += gridDim.x) {
            f4 v[MT];
            fetch(item, v);  // (requesting the NEXT item's arrays here instead, to land under the MFMAs: 2170 -> 2202 us per update, measured r3B - not kept)
            tile_write<MT>(wave == 0 ? T0 : (wave == 1 ? T1 : (wave == 2 ? T2 : T3)), v, g, j);
            __syncthreads();
            f4 bX[MTN], bH[MTN];
#pragma unroll
            for (int nt = 0; nt < MTN; ++nt) {
                bX[nt] = tile_read(T2, nt0 + nt, g, j);
                bH[nt] = tile_read(T3, nt0 + nt, g, j);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f4 aI = tile_read(T0, mt, g, j), aH = tile_read(T1, mt, g, j);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int nt = 0; nt < MTN; ++nt) {
                        dWa[mt][nt] = MARL_MFMA(aI[ks], bX[nt][ks], dWa[mt][nt]);
                        dWb[mt][nt] = MARL_MFMA(aH[ks], bH[nt][ks], dWb[mt][nt]);
                    }
            }
            __syncthreads();  // tiles are rewritten by the next item
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int v = role * H + 16 * mt + 4 * g + r;
#pragma unroll
                for (int nt = 0; nt < MTN; ++nt) {
                    recd[S::oWih + v * H + 16 * (nt0 + nt) + j] = dWa[mt][nt][r];
                    recd[S::oWhh + v * H + 16 * (nt0 + nt) + j] = dWb[mt][nt][r];
                }
            }
        return;
    }
    // ---- role 3: the small parts, one per wave (no cross-wave traffic: per-wave tiles)
    if ((wave == 0 && !(parts & 1)) || (wave == 1 && !(parts & 2))) return;
    float* TA = wave == 0 ? T0 : T1;  // wave 0: dx1 tile; wave 1: h tile (+ TQ)
    f4 acc1[MT][NT1], acc3[MT], sa[MT], sb[MT], sc[MT], db3 = zero4;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        acc3[a] = zero4; sa[a] = zero4; sb[a] = zero4; sc[a] = zero4;
#pragma unroll
        for (int b = 0; b < NT1; ++b) acc1[a][b] = zero4;
    }
    float loss_acc = 0.f, nf_acc = 0.f;
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        const int t = item / nblk, blk = item - t * nblk;
        const int b0 = blk * 16;
        const f4* R = reinterpret_cast<const f4*>(rec + (((size_t)p * steps + t) * nblk + blk) * S::REC);
        const f4* R2 = reinterpret_cast<const f4*>(rec2 + (((size_t)p * steps + t) * nblk + blk) * Bk::REC2);
        if (wave == 0) {  // dW1 += dx1^T x (B operand straight from the observations), db1
            f4 v[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { v[mt] = R2[(4 * MT + mt) * 64 + lane]; sa[mt] += v[mt]; }
            wave_lds_fence();
            tile_write<MT>(TA, v, g, j);
            wave_lds_fence();
            float bx[NT1][4];
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int row = b0 + 4 * g + ks, d = 16 * nt + j;
                    const float xv = obs[(size_t)p * obs_as + ((size_t)t * B + (row < B ? row : B - 1)) * obs_rs + (d < D ? d : D - 1)];
                    bx[nt][ks] = (row < B && d < D) ? xv : 0.f;
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f4 aX = tile_read(TA, mt, g, j);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int nt = 0; nt < NT1; ++nt) acc1[mt][nt] = MARL_MFMA(aX[ks], bx[nt][ks], acc1[mt][nt]);
            }
        } else if (wave == 1) {  // dW3 += dq^T h, db3, loss bookkeeping
            f4 v[MT], dQ[1];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) v[mt] = R[(4 * MT + mt) * 64 + lane];
            const bool rowok = b0 + j < B;
            const float* drow = dq + (((size_t)p * steps + t) * B + (rowok ? b0 + j : B - 1)) * A;
#pragma unroll
            for (int r = 0; r < 4; ++r) dQ[0][r] = (rowok && 4 * g + r < A) ? drow[4 * g + r < A ? 4 * g + r : A - 1] : 0.f;
            db3 += dQ[0];
            if (g == 0 && p == 0 && rowok && t < T) {
                loss_acc += lrow[(size_t)t * B + b0 + j];
                nf_acc += filled[(size_t)t * B + b0 + j];
            }
            wave_lds_fence();
            tile_write<MT>(TA, v, g, j);
            tile_write<1>(TQ, dQ, g, j);
            wave_lds_fence();
            const f4 aQ = tile_read(TQ, 0, g, j);
#pragma unroll
            for (int nt = 0; nt < MT; ++nt) {
                const f4 bH = tile_read(TA, nt, g, j);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc3[nt] = MARL_MFMA(aQ[ks], bH[ks], acc3[nt]);
            }
        } else if (wave == 2) {  // bias sums of gates r and z (dgi == dgh there)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { sa[mt] += R2[(0 * MT + mt) * 64 + lane]; sb[mt] += R2[(1 * MT + mt) * 64 + lane]; }
        } else {  // bias sums of gate n: dn (ih) and r * dn (hh)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { sa[mt] += R2[(2 * MT + mt) * 64 + lane]; sc[mt] += R2[(3 * MT + mt) * 64 + lane]; }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = 16 * mt + 4 * g + r;
            const float s_a = sum16(sa[mt][r]), s_b = sum16(sb[mt][r]), s_c = sum16(sc[mt][r]);
            if (wave == 0) {
#pragma unroll
                for (int nt = 0; nt < NT1; ++nt)
                    if (16 * nt + j < D) recd[S::oW1 + v * D + 16 * nt + j] = acc1[mt][nt][r];
                if (j == 0) recd[S::ob1 + v] = s_a;
            } else if (wave == 1) {
                const int a = 4 * g + r;  // acc3[nt][r]: element (a, unit 16nt + j) - here `mt` plays nt
                if (a < A) recd[S::oW3 + a * H + 16 * mt + j] = acc3[mt][r];
            } else if (wave == 2) {
                if (j == 0) {
                    recd[S::obih + v] = s_a; recd[S::obhh + v] = s_a;
                    recd[S::obih + H + v] = s_b; recd[S::obhh + H + v] = s_b;
                }
            } else if (j == 0) {
                recd[S::obih + 2 * H + v] = s_a;
                recd[S::obhh + 2 * H + v] = s_c;
            }
        }
    if (wave == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s3 = sum16(db3[r]);
            if (j == 0 && 4 * g + r < A) recd[S::ob3 + 4 * g + r] = s3;
        }
        const float ls = sum16(loss_acc), nf = sum16(nf_acc);
        if (lane == 0) {
            recd[S::NPARAM] = ls;
            recd[S::NPARAM + 1] = nf;
        }
    }
}

}  // namespace marl
