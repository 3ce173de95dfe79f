// Per-agent 3-layer MLP (Linear-ReLU-Linear-ReLU-Linear) on f32 MFMA for gfx950.
//
// Replaces FCNetwork.forward (marlbase/utils/models.py:34-48) as used by
// MultiAgentIndependentNetwork (utils/models.py:156-170) inside QNetwork.act
// (dqn/model.py:94-116) and QNetwork._compute_loss (dqn/model.py:127-134).
//
// Formulation: activations are kept TRANSPOSED, Y^T[out x 16 rows] = W[out x in] *
// X^T[in x 16 rows], on v_mfma_f32_16x16x4_f32 (exact f32, == an fmaf chain):
//   A operand (lane l: A[i=l&15][k=l>>4])  = weights, read from LDS
//   B operand (lane l: B[k=l>>4][j=l&15])  = activations, one register per k-step
//   C/D       (lane l: D[row=(l>>4)*4+r][col=l&15], r=0..3)
// so lane (g=l>>4, j=l&15) holds outputs 16*mt+4g+r of batch row j.  That is
// exactly the B-operand shape of the next layer if its k-steps are enumerated as
// (mt, r) with k-index 16*mt+4g+r: layers chain with NO cross-lane movement; only
// the weight packing knows about the permuted k order.  Per output the f32 sum is
//   acc = bias; for mt: for r: for g=0..3: acc = fmaf(W[o][16mt+4g+r], x[16mt+4g+r], acc)
// (layer 1: for ks: for g: k = 4ks+g) - oracle/mfma_emul.py is the lane-level statement of this order (the learner kernels of
// dqn_update_kernels.h split the output layer into two chains and shortcut one-hot products: same math, different rounding order).
//
// Canonical parameters of one agent = torch parameters() order of FCNetwork:
//   W1[H][D] b1[H] W2[H][H] b2[H] W3[A][H] b3[A]      (row-major, nn.Linear layout)
#pragma once
#include <type_traits>
#include <utility>
#include <hip/hip_runtime.h>

namespace marl {

typedef float f4 __attribute__((ext_vector_type(4)));

#define MARL_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#ifndef MARL_MFMA_VGPR
#define MARL_MFMA_VGPR 1
#endif
#define MARL_MFMA_VGPR_DEFAULT MARL_MFMA_VGPR

template <int D_, int H_, int A_>
struct MlpShape {
    static constexpr int D = D_, H = H_, A = A_;
    static constexpr int DP = (D_ + 15) / 16 * 16;  // layer-1 K padded to 16 (float4 of k-steps)
    static constexpr int KS1 = DP / 4;              // layer-1 k-steps
    static constexpr int MT = H_ / 16;              // hidden tiles of 16
    static constexpr int AP = 16;                   // output tile (A <= 16)
    static_assert(H_ % 16 == 0 && A_ <= 16, "shape");
    // canonical offsets
    static constexpr int oW1 = 0, ob1 = H_ * D_, oW2 = ob1 + H_, ob2 = oW2 + H_ * H_, oW3 = ob2 + H_, ob3 = oW3 + A_ * H_;
    static constexpr int NPARAM = ob3 + A_;
    // forward pack (floats): A1[MT][KS1/4][64][4] A2[MT][MT][64][4] b1[H] b2[H] b3[16] A3[MT][64][4]
    // (A3 last: the prefix of NFWD_NOA3 floats is a complete pack for callers that keep the output layer in registers)
    static constexpr int pA1 = 0, pA2 = pA1 + MT * KS1 * 64, pb1 = pA2 + MT * MT * 256, pb2 = pb1 + H_, pb3 = pb2 + H_, pA3 = pb3 + 16;
    static constexpr int NFWD = pA3 + MT * 256, NFWD_NOA3 = pA3;
    // backward-data pack: T3[MT][64][4] = W3^T tiles (M=h2, K=a), T2[MT][MT][64][4] = W2^T tiles (M=h1, K=h2)
    static constexpr int pT3 = 0, pT2 = pT3 + MT * 256;
    static constexpr int NBWD = pT2 + MT * MT * 256;
};

// value of forward-pack element `idx` taken from the canonical parameter block
template <class S>
__device__ __forceinline__ float mlp_fwd_pack_elem(const float* __restrict__ w, int idx) {
    if (idx < S::pA2) {  // A1[mt][ks4][lane][e]: W1[16mt+i][4(4ks4+e)+g]
        const int e = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
        const int ks4 = rest % (S::KS1 / 4), mt = rest / (S::KS1 / 4);
        const int o = 16 * mt + (lane & 15), k = 4 * (4 * ks4 + e) + (lane >> 4);
        return k < S::D ? w[S::oW1 + o * S::D + k] : 0.f;
    } else if (idx < S::pb1) {  // A2[mt2][mt1][lane][r]: W2[16mt2+i][16mt1+4g+r]
        const int j = idx - S::pA2;
        const int r = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
        const int mt1 = rest % S::MT, mt2 = rest / S::MT;
        return w[S::oW2 + (16 * mt2 + (lane & 15)) * S::H + 16 * mt1 + 4 * (lane >> 4) + r];
    } else if (idx < S::pb2) {
        return w[S::ob1 + idx - S::pb1];
    } else if (idx < S::pb3) {
        return w[S::ob2 + idx - S::pb2];
    } else if (idx < S::pA3) {
        const int o = idx - S::pb3;
        return o < S::A ? w[S::ob3 + o] : 0.f;
    } else {  // A3[mt1][lane][r]: W3[i][16mt1+4g+r]
        const int j = idx - S::pA3;
        const int r = j & 3, lane = (j >> 2) & 63, mt1 = j >> 8;
        const int o = lane & 15;
        return o < S::A ? w[S::oW3 + o * S::H + 16 * mt1 + 4 * (lane >> 4) + r] : 0.f;
    }
}

// cooperative (whole workgroup) staging of one network's forward pack into LDS
template <class S>
__device__ __forceinline__ void mlp_stage_fwd(const float* __restrict__ w, float* lds, int tid, int nthreads) {
    for (int idx = tid; idx < S::NFWD; idx += nthreads) lds[idx] = mlp_fwd_pack_elem<S>(w, idx);
}

template <class S>
__device__ __forceinline__ float mlp_bwd_pack_elem(const float* __restrict__ w, int idx) {
    if (idx < S::pT2) {  // T3[mt][lane][r]: A[i=h2 16mt+i][k=a 4g+r] = W3[4g+r][16mt+i]
        const int r = idx & 3, lane = (idx >> 2) & 63, mt = idx >> 8;
        const int a = 4 * (lane >> 4) + r;
        return a < S::A ? w[S::oW3 + a * S::H + 16 * mt + (lane & 15)] : 0.f;
    }
    // T2[mt1][mt2][lane][r]: A[i=h1 16mt1+i][k=h2 16mt2+4g+r] = W2[16mt2+4g+r][16mt1+i]
    const int j = idx - S::pT2;
    const int r = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
    const int mt2 = rest % S::MT, mt1 = rest / S::MT;
    return w[S::oW2 + (16 * mt2 + 4 * (lane >> 4) + r) * S::H + 16 * mt1 + (lane & 15)];
}

template <class S>
__device__ __forceinline__ void mlp_stage_bwd(const float* __restrict__ w, float* lds, int tid, int nthreads) {
    for (int idx = tid; idx < S::NBWD; idx += nthreads) {
        float v;
        if (idx < S::pT2) {  // T3[mt][lane][r]: A[i=h2 16mt+i][k=a 4g+r] = W3[4g+r][16mt+i]
            const int r = idx & 3, lane = (idx >> 2) & 63, mt = idx >> 8;
            const int a = 4 * (lane >> 4) + r;
            v = a < S::A ? w[S::oW3 + a * S::H + 16 * mt + (lane & 15)] : 0.f;
        } else {  // T2[mt1][mt2][lane][r]: A[i=h1 16mt1+i][k=h2 16mt2+4g+r] = W2[16mt2+4g+r][16mt1+i]
            const int j = idx - S::pT2;
            const int r = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
            const int mt2 = rest % S::MT, mt1 = rest / S::MT;
            v = w[S::oW2 + (16 * mt2 + 4 * (lane >> 4) + r) * S::H + 16 * mt1 + (lane & 15)];
        }
        lds[idx] = v;
    }
}

// inverse of mlp_fwd_pack_elem / mlp_bwd_pack_elem: where canonical parameter k of a network sits in the forward pack (always) and
// in the backward pack (W2, W3; -1 otherwise).  Zero padding of the packs (k-steps beyond D, outputs beyond A) has no parameter.
template <class S>
__device__ __forceinline__ void mlp_param_to_pack(int k, int& fwd, int& bwd) {
    constexpr int MT = S::MT;
    bwd = -1;
    if (k < S::ob1) {  // W1[o][d]
        const int o = k / S::D, d = k - o * S::D;
        const int ks = d >> 2, gq = d & 3;
        fwd = S::pA1 + (((o >> 4) * (S::KS1 / 4) + (ks >> 2)) * 64 + gq * 16 + (o & 15)) * 4 + (ks & 3);
    } else if (k < S::oW2) {
        fwd = S::pb1 + (k - S::ob1);
    } else if (k < S::ob2) {  // W2[o2][i1]
        const int j = k - S::oW2, o2 = j / S::H, i1 = j - o2 * S::H;
        fwd = S::pA2 + (((o2 >> 4) * MT + (i1 >> 4)) * 64 + ((i1 & 15) >> 2) * 16 + (o2 & 15)) * 4 + (i1 & 3);
        bwd = S::pT2 + (((i1 >> 4) * MT + (o2 >> 4)) * 64 + ((o2 & 15) >> 2) * 16 + (i1 & 15)) * 4 + (o2 & 3);
    } else if (k < S::oW3) {
        fwd = S::pb2 + (k - S::ob2);
    } else if (k < S::ob3) {  // W3[a][h]
        const int j = k - S::oW3, a = j / S::H, h = j - a * S::H;
        fwd = S::pA3 + ((h >> 4) * 64 + ((h & 15) >> 2) * 16 + a) * 4 + (h & 3);
        bwd = S::pT3 + ((h >> 4) * 64 + (a >> 2) * 16 + (h & 15)) * 4 + (a & 3);
    } else {
        fwd = S::pb3 + (k - S::ob3);
    }
}

// workgroup-cooperative 16-byte copy global -> LDS with 8 independent loads in flight per thread before the first LDS store
__device__ __forceinline__ void copy_f4_to_lds(const f4* __restrict__ src, f4* dst, int n4, int tid, int nthreads) {
    constexpr int U = 8;
    int i = tid;
    for (; i + (U - 1) * nthreads < n4; i += U * nthreads) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * nthreads];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * nthreads] = v[u];
    }
    for (; i < n4; i += nthreads) dst[i] = src[i];
}

// relu on four accumulator values.  (fmaxf lowers to a canonicalising v_max(v, v) in front of the max; the one-instruction asm
// spelling is relu4_settled below - inline asm is invisible to the compiler's MFMA -> VALU hazard pass, so it may only read accumulators
// that an explicit mfma_settle() has already waited for.  Used anywhere else it reads stale registers: r02 ac_collector regression.)
__device__ __forceinline__ f4 relu4(f4 v) { return f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }

// forward of one 16-row block.  x[ks] = X[row j][4ks+g] (zero beyond D).
// h1/h2 (post-ReLU) and q come back in C layout: [16mt+4g+r][row j].
template <class S>
__device__ __forceinline__ void mlp_forward(const float* lds, int lane, const float (&x)[S::KS1], f4 (&h1)[S::MT], f4 (&h2)[S::MT], f4& q) {
    const int g = lane >> 4;
    const f4* A1 = reinterpret_cast<const f4*>(lds + S::pA1);
    const f4* A2 = reinterpret_cast<const f4*>(lds + S::pA2);
    const f4* A3 = reinterpret_cast<const f4*>(lds + S::pA3);
    f4 acc[S::MT];
#pragma unroll
    for (int mt = 0; mt < S::MT; ++mt) acc[mt] = *reinterpret_cast<const f4*>(lds + S::pb1 + 16 * mt + 4 * g);
#pragma unroll
    for (int ks4 = 0; ks4 < S::KS1 / 4; ++ks4) {
        f4 a[S::MT];
#pragma unroll
        for (int mt = 0; mt < S::MT; ++mt) a[mt] = A1[(mt * (S::KS1 / 4) + ks4) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < S::MT; ++mt) acc[mt] = MARL_MFMA(a[mt][e], x[4 * ks4 + e], acc[mt]);
    }
#pragma unroll
    for (int mt = 0; mt < S::MT; ++mt) h1[mt] = relu4(acc[mt]);

#pragma unroll
    for (int mt = 0; mt < S::MT; ++mt) acc[mt] = *reinterpret_cast<const f4*>(lds + S::pb2 + 16 * mt + 4 * g);
#pragma unroll
    for (int k1 = 0; k1 < S::MT; ++k1) {
        f4 a[S::MT];
#pragma unroll
        for (int mt = 0; mt < S::MT; ++mt) a[mt] = A2[(mt * S::MT + k1) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < S::MT; ++mt) acc[mt] = MARL_MFMA(a[mt][r], h1[k1][r], acc[mt]);
    }
#pragma unroll
    for (int mt = 0; mt < S::MT; ++mt) h2[mt] = relu4(acc[mt]);

    // layer 3: two interleaved partial chains would change the sum order; keep one chain
    f4 o = *reinterpret_cast<const f4*>(lds + S::pb3 + 4 * g);
#pragma unroll
    for (int k1 = 0; k1 < S::MT; ++k1) {
        const f4 a = A3[k1 * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) o = MARL_MFMA(a[r], h2[k1][r], o);
    }
    q = o;
}

// Software-pipelined forward (one network, or critic+target on the same rows when DUAL): the
// A-operand fetch (ds_read_b128) of step s+1 is issued BEFORE the MFMAs of step s and pinned there
// with sched_barrier, so with one wave per SIMD the LDS latency hides under >= 16 MFMAs instead of
// stalling the matrix pipe in front of every group (hipcc places the reads right before their use).
// Steps: layer 1 (KS1/4 steps), layer 2 (MT steps), layer 3 (1 step).  Per-output summation order
// is unchanged (bitwise identical to mlp_forward).
// a3r != nullptr (single network only): the layer-3 operands come from those MT registers instead of the pack's A3 block,
// so the pack in LDS can be the NFWD_NOA3 prefix.
template <class S, bool DUAL>
__device__ __forceinline__ void mlp_forward_p(const float* ldsA, const float* ldsB, int lane, const float (&x)[S::KS1],
                                              f4 (&h1)[S::MT], f4 (&h2)[S::MT], f4& qA, f4& qB, const f4* a3r = nullptr) {
    constexpr int MT = S::MT, N1 = S::KS1 / 4;
    const int g = lane >> 4;
    const f4* A1 = reinterpret_cast<const f4*>(ldsA + S::pA1);
    const f4* A2 = reinterpret_cast<const f4*>(ldsA + S::pA2);
    const f4* A3 = reinterpret_cast<const f4*>(ldsA + S::pA3);
    const f4* B1 = reinterpret_cast<const f4*>(ldsB + S::pA1);
    const f4* B2 = reinterpret_cast<const f4*>(ldsB + S::pA2);
    const f4* B3 = reinterpret_cast<const f4*>(ldsB + S::pA3);
    f4 opa[2][MT], opb[2][MT], accA[MT], accB[MT], nbA[MT], nbB[MT], g1[MT], g2[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        opa[0][mt] = A1[(mt * N1 + 0) * 64 + lane];
        accA[mt] = *reinterpret_cast<const f4*>(ldsA + S::pb1 + 16 * mt + 4 * g);
        if (DUAL) {
            opb[0][mt] = B1[(mt * N1 + 0) * 64 + lane];
            accB[mt] = *reinterpret_cast<const f4*>(ldsB + S::pb1 + 16 * mt + 4 * g);
        }
    }
    // ---- layer 1
#pragma unroll
    for (int s = 0; s < N1; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (s + 1 < N1) {
                opa[nxt][mt] = A1[(mt * N1 + s + 1) * 64 + lane];
                if (DUAL) opb[nxt][mt] = B1[(mt * N1 + s + 1) * 64 + lane];
            } else {  // first layer-2 step + its bias
                opa[nxt][mt] = A2[(mt * MT + 0) * 64 + lane];
                nbA[mt] = *reinterpret_cast<const f4*>(ldsA + S::pb2 + 16 * mt + 4 * g);
                if (DUAL) {
                    opb[nxt][mt] = B2[(mt * MT + 0) * 64 + lane];
                    nbB[mt] = *reinterpret_cast<const f4*>(ldsB + S::pb2 + 16 * mt + 4 * g);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                accA[mt] = MARL_MFMA(opa[cur][mt][e], x[4 * s + e], accA[mt]);
                if (DUAL) accB[mt] = MARL_MFMA(opb[cur][mt][e], x[4 * s + e], accB[mt]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        h1[mt] = relu4(accA[mt]);
        accA[mt] = nbA[mt];
        if (DUAL) {
            g1[mt] = relu4(accB[mt]);
            accB[mt] = nbB[mt];
        }
    }
    // ---- layer 2
    f4 o3A, o3B;
#pragma unroll
    for (int k1 = 0; k1 < MT; ++k1) {
        const int cur = (N1 + k1) & 1, nxt = cur ^ 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (k1 + 1 < MT) {
                opa[nxt][mt] = A2[(mt * MT + k1 + 1) * 64 + lane];
                if (DUAL) opb[nxt][mt] = B2[(mt * MT + k1 + 1) * 64 + lane];
            } else {  // layer-3 operands (MT tiles of K) + bias
                opa[nxt][mt] = a3r != nullptr ? a3r[mt] : A3[mt * 64 + lane];
                if (DUAL) opb[nxt][mt] = B3[mt * 64 + lane];
            }
        }
        if (k1 + 1 == MT) {
            o3A = *reinterpret_cast<const f4*>(ldsA + S::pb3 + 4 * g);
            if (DUAL) o3B = *reinterpret_cast<const f4*>(ldsB + S::pb3 + 4 * g);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                accA[mt] = MARL_MFMA(opa[cur][mt][r], h1[k1][r], accA[mt]);
                if (DUAL) accB[mt] = MARL_MFMA(opb[cur][mt][r], g1[k1][r], accB[mt]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        h2[mt] = relu4(accA[mt]);
        if (DUAL) g2[mt] = relu4(accB[mt]);
    }
    // ---- layer 3 (one chain per network, K order as mlp_forward)
    constexpr int c3 = (N1 + MT) & 1;
#pragma unroll
    for (int k1 = 0; k1 < MT; ++k1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o3A = MARL_MFMA(opa[c3][k1][r], h2[k1][r], o3A);
            if (DUAL) o3B = MARL_MFMA(opb[c3][k1][r], g2[k1][r], o3B);
        }
    qA = o3A;
    if (DUAL) qB = o3B;
}

// One network on TWO waves (round 4: the collectors on launches that would leave half of the SIMDs idle - 4096 envs x 2 agents are 512
// one-agent waves).  Wave `half` owns hidden tiles [half MT/2, (half + 1) MT/2) of layers 1 and 2 (every output tile is accumulated by one
// wave in mlp_forward_p's k order); the relu'd layer-1 tiles cross through LDS (xh: [2][MT/2][64] f4 per network); layer 3 is ONE chain
// in mlp_forward_p's order - half 0 starts from the bias with k1 < MT/2, hands the partial sum over (xq: [64] f4), half 1 finishes it.
// The result (valid in the half-1 wave only) has the bits of the one-wave forward.  Two workgroup barriers: every wave of the workgroup
// must make the call.
// XR: the layer-1 tiles cross in XR rounds through an xh of [2][MT / 2 / XR][64] f4 (hidden 128 with the output layer's operands in registers
// leaves 10 KB of LDS next to the packs: two rounds of 2 tiles); a3r: the half's MT/2 layer-3 operand tiles when the pack in LDS is the
// NFWD_NOA3 prefix.
template <class S, int XR = 1>
__device__ __forceinline__ void mlp_forward_h2(const float* lds, int lane, const float (&x)[S::KS1], int half, f4* xh, f4* xq, f4& q, const f4* a3r = nullptr) {
    constexpr int MT = S::MT, MH = MT / 2, N1 = S::KS1 / 4, MX = MH / XR;
    static_assert(MT % 2 == 0 && MH % XR == 0, "two waves split the hidden tiles evenly");
    const int g = lane >> 4, t0 = half * MH;
    const f4* A1 = reinterpret_cast<const f4*>(lds + S::pA1);
    const f4* A2 = reinterpret_cast<const f4*>(lds + S::pA2);
    const f4* A3 = reinterpret_cast<const f4*>(lds + S::pA3);
    f4 acc[MH], h1[MT], h2[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) acc[m] = *reinterpret_cast<const f4*>(lds + S::pb1 + 16 * (t0 + m) + 4 * g);
#pragma unroll
    for (int s = 0; s < N1; ++s) {
        f4 op[MH];
#pragma unroll
        for (int m = 0; m < MH; ++m) op[m] = A1[((t0 + m) * N1 + s) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < MH; ++m) acc[m] = MARL_MFMA(op[m][e], x[4 * s + e], acc[m]);
    }
    f4 other[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) acc[m] = relu4(acc[m]);
#pragma unroll
    for (int xr = 0; xr < XR; ++xr) {
        if (xr > 0) __syncthreads();  // the previous round has been read
#pragma unroll
        for (int m = 0; m < MX; ++m) xh[(half * MX + m) * 64 + lane] = acc[xr * MX + m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MX; ++m) other[xr * MX + m] = xh[((1 - half) * MX + m) * 64 + lane];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) h1[mt] = (mt / MH == half) ? acc[mt % MH] : other[mt % MH];  // (compile-time register indices, run-time select)
#pragma unroll
    for (int m = 0; m < MH; ++m) acc[m] = *reinterpret_cast<const f4*>(lds + S::pb2 + 16 * (t0 + m) + 4 * g);
#pragma unroll
    for (int k1 = 0; k1 < MT; ++k1) {
        f4 op[MH];
#pragma unroll
        for (int m = 0; m < MH; ++m) op[m] = A2[((t0 + m) * MT + k1) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MH; ++m) acc[m] = MARL_MFMA(op[m][r], h1[k1][r], acc[m]);
    }
#pragma unroll
    for (int m = 0; m < MH; ++m) h2[m] = relu4(acc[m]);
    f4 op3[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) op3[m] = a3r != nullptr ? a3r[m] : A3[(t0 + m) * 64 + lane];
    f4 o3 = *reinterpret_cast<const f4*>(lds + S::pb3 + 4 * g);
    if (half == 0) {
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) o3 = MARL_MFMA(op3[m][r], h2[m][r], o3);
        xq[lane] = o3;
    }
    __syncthreads();
    if (half == 1) {
        o3 = xq[lane];
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) o3 = MARL_MFMA(op3[m][r], h2[m][r], o3);
    }
    q = o3;
}

template <int I, int N, class F>
__device__ __forceinline__ void marl_static_for(F& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        marl_static_for<I + 1, N>(f);
    }
}

// mlp_forward_p for a pack that stays in GLOBAL memory (the collectors' agent-per-wave form when the packs of all agents do not fit
// the LDS): the same operand groups in the same order - N1 first-layer k-groups, MT second-layer groups, the output group - but
// fetched THREE groups deep (a group = MT 16-byte loads feeding 4 * MT MFMAs, ~1000 cycles at hidden 128: one group of lookahead
// does not cover an L2 round trip, two do).  Same per-output summation order as mlp_forward_p / mlp_forward.
template <class S>
__device__ __forceinline__ void mlp_forward_g(const float* __restrict__ gpack, int lane, const float (&x)[S::KS1], f4& q) {
    constexpr int MT = S::MT, N1 = S::KS1 / 4, G = N1 + MT + 1;
    // Addressing (round 4): BUFFER loads - the pack pointer is wave-uniform (the callers derive the agent from a readfirstlane'd wave index)
    // and becomes a 4-SGPR resource; every operand is then `resource + scalar offset + 16 * lane`: one vector register of lane offsets for
    // the whole pass.  With 64-bit per-lane pointers (global_load) the ~100 distinct operand addresses of a hidden-128 pass lived in
    // vector registers - the compiler does not split `uniform base + constant + lane offset` into the scalar-base form beyond the 4 KB
    // immediate range - and the 8-agent kernels spilled them: every reload waited out the loads in flight (s_waitcnt vmcnt(0) in front
    // of the dependent load), 85 k cycles per step for two forward passes against 16 k for one.
    typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gpack), 0, S::NFWD * 4, 0x00020000);
    const int lo16 = lane * 16, g16 = (lane >> 4) * 16;
    auto ld16 = [&](int float_off, int voff) -> f4 {
        const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, float_off * 4, 0);
        return __builtin_bit_cast(f4, v);
    };
    f4 op[3][MT], acc[MT], nb[MT], h1[MT], h2[MT], o3;
    auto fetch = [&](auto gi_c) {
        constexpr int gi = decltype(gi_c)::value;
        if constexpr (gi < G) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                op[gi % 3][mt] = gi < N1 ? ld16(S::pA1 + (mt * N1 + gi) * 256, lo16)
                                         : (gi < N1 + MT ? ld16(S::pA2 + (mt * MT + (gi - N1)) * 256, lo16) : ld16(S::pA3 + mt * 256, lo16));
        }
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        acc[mt] = ld16(S::pb1 + 16 * mt, g16);
        nb[mt] = ld16(S::pb2 + 16 * mt, g16);
    }
    o3 = ld16(S::pb3, g16);
    fetch(std::integral_constant<int, 0>{});
    fetch(std::integral_constant<int, 1>{});
    auto step = [&](auto gi_c) {
        constexpr int gi = decltype(gi_c)::value;
        fetch(std::integral_constant<int, gi + 2>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (gi < N1) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = MARL_MFMA(op[gi % 3][mt][e], x[4 * gi + e], acc[mt]);
            if constexpr (gi == N1 - 1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    h1[mt] = relu4(acc[mt]);
                    acc[mt] = nb[mt];
                }
            }
        } else if constexpr (gi < N1 + MT) {
            constexpr int k1 = gi - N1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = MARL_MFMA(op[gi % 3][mt][r], h1[k1][r], acc[mt]);
            if constexpr (k1 == MT - 1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) h2[mt] = relu4(acc[mt]);
            }
        } else {
#pragma unroll
            for (int k1 = 0; k1 < MT; ++k1)
#pragma unroll
                for (int r = 0; r < 4; ++r) o3 = MARL_MFMA(op[gi % 3][k1][r], h2[k1][r], o3);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    marl_static_for<0, G>(step);  // every group index is a compile-time constant
    q = o3;
}

// One network on TWO 16-row blocks at once: every A operand fetched from LDS feeds two MFMAs, which halves the LDS
// operand traffic per flop (at hidden 128 a single-block forward reads 327 KB of operands per 320 MFMAs - the LDS port
// and the matrix pipe then run at the same rate).  Same pipelining and per-output summation order as mlp_forward_p.
template <class S, bool WANT_H2 = false, bool WANT_H1 = false>
__device__ __forceinline__ void mlp_forward_p2(const float* lds, int lane, const float (&x)[2][S::KS1], f4 (&q)[2], f4 (*h2_out)[S::MT] = nullptr,
                                               f4 (*h1_out)[S::MT] = nullptr) {
    constexpr int MT = S::MT, N1 = S::KS1 / 4;
    const int g = lane >> 4;
    const f4* A1 = reinterpret_cast<const f4*>(lds + S::pA1);
    const f4* A2 = reinterpret_cast<const f4*>(lds + S::pA2);
    const f4* A3 = reinterpret_cast<const f4*>(lds + S::pA3);
    f4 op[2][MT], acc[2][MT], nb[MT], h1[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        op[0][mt] = A1[(mt * N1 + 0) * 64 + lane];
        acc[0][mt] = *reinterpret_cast<const f4*>(lds + S::pb1 + 16 * mt + 4 * g);
        acc[1][mt] = acc[0][mt];
    }
#pragma unroll
    for (int s = 0; s < N1; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (s + 1 < N1) {
                op[nxt][mt] = A1[(mt * N1 + s + 1) * 64 + lane];
            } else {
                op[nxt][mt] = A2[(mt * MT + 0) * 64 + lane];
                nb[mt] = *reinterpret_cast<const f4*>(lds + S::pb2 + 16 * mt + 4 * g);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[0][mt] = MARL_MFMA(op[cur][mt][e], x[0][4 * s + e], acc[0][mt]);
                acc[1][mt] = MARL_MFMA(op[cur][mt][e], x[1][4 * s + e], acc[1][mt]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        h1[0][mt] = relu4(acc[0][mt]);
        h1[1][mt] = relu4(acc[1][mt]);
        if constexpr (WANT_H1) {
            h1_out[0][mt] = h1[0][mt];
            h1_out[1][mt] = h1[1][mt];
        }
        acc[0][mt] = nb[mt];
        acc[1][mt] = nb[mt];
    }
    f4 o3;
#pragma unroll
    for (int k1 = 0; k1 < MT; ++k1) {
        const int cur = (N1 + k1) & 1, nxt = cur ^ 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) op[nxt][mt] = (k1 + 1 < MT) ? A2[(mt * MT + k1 + 1) * 64 + lane] : A3[mt * 64 + lane];
        if (k1 + 1 == MT) o3 = *reinterpret_cast<const f4*>(lds + S::pb3 + 4 * g);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[0][mt] = MARL_MFMA(op[cur][mt][r], h1[0][k1][r], acc[0][mt]);
                acc[1][mt] = MARL_MFMA(op[cur][mt][r], h1[1][k1][r], acc[1][mt]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int c3 = (N1 + MT) & 1;
    q[0] = o3;
    q[1] = o3;
    if constexpr (WANT_H2) {  // the second hidden layer (post-relu, C layout) for a backward pass that does not want to recompute it
#pragma unroll
        for (int k1 = 0; k1 < MT; ++k1) { h2_out[0][k1] = relu4(acc[0][k1]); h2_out[1][k1] = relu4(acc[1][k1]); }
    }
#pragma unroll
    for (int k1 = 0; k1 < MT; ++k1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f4 h20 = relu4(acc[0][k1]), h21 = relu4(acc[1][k1]);
            q[0] = MARL_MFMA(op[c3][k1][r], h20[r], q[0]);
            q[1] = MARL_MFMA(op[c3][k1][r], h21[r], q[1]);
        }
}

// ---- cross-lane helpers on gfx950's v_permlane16_swap / v_permlane32_swap (VALU only: no LDS queue entry, no lgkmcnt wait).
// permlane16_swap(a, b): rows (16 lanes) 1, 3 of a <-> rows 0, 2 of b; permlane32_swap(a, b): lanes 32..63 of a <-> lanes 0..31 of b.
// With a == b == v the two results are [r0 r0 r2 r2] / [r1 r1 r3 r3] (resp. [lo lo] / [hi hi]): their sum is the xor-16 (xor-32)
// pair sum in every lane, and "the other lane's value" is result[1] in even rows (lower half), result[0] in odd rows (upper half).
__device__ __forceinline__ float xor16_other(float v, int lane) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float xor32_other(float v, int lane) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ int xor16_other(int v, int lane) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((lane & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ int xor32_other(int v, int lane) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((lane & 32) ? r[0] : r[1]);
}
// sum over the four lanes {j, j+16, j+32, j+48} that carry batch row j (every lane gets it)
__device__ __forceinline__ float sum_g(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// argmax_rows / gather_rows without LDS traffic (same results: first index of the maximum; exactly one non-zero addend)
template <int A>
__device__ __forceinline__ int argmax_rows_pl(const f4& q, int lane) {
    // selects only (& / | on the predicates, no short-circuit): a branch here would split the MFMA region it hides in
    const int g = lane >> 4;
    float bv = -__builtin_huge_valf();
    int ba = 0x7FFFFFFF;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a = 4 * g + r;
        const bool take = (a < A) & ((q[r] > bv) | (ba == 0x7FFFFFFF));
        bv = take ? q[r] : bv;
        ba = take ? a : ba;
    }
    {
        const float ov = xor16_other(bv, lane);
        const int oa = xor16_other(ba, lane);
        const bool take = (oa != 0x7FFFFFFF) & ((ba == 0x7FFFFFFF) | (ov > bv) | ((ov == bv) & (oa < ba)));
        bv = take ? ov : bv;
        ba = take ? oa : ba;
    }
    if (A > 8) {
        const float ov = xor32_other(bv, lane);
        const int oa = xor32_other(ba, lane);
        const bool take = (oa != 0x7FFFFFFF) & ((ba == 0x7FFFFFFF) | (ov > bv) | ((ov == bv) & (oa < ba)));
        bv = take ? ov : bv;
        ba = take ? oa : ba;
    } else {  // up to 8 actions sit in lane groups g = 0, 1: the 16-lane exchange decided; groups 2, 3 just take the result over
        const int oa = xor32_other(ba, lane);
        ba = (lane & 32) ? oa : ba;
    }
    return ba;
}
template <int A>
__device__ __forceinline__ float gather_rows_pl(const f4& q, int lane, int a_sel) {
    const int g = lane >> 4;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) v += (4 * g + r == a_sel) ? q[r] : 0.f;
    return sum_g(v);
}

// relu for the learner kernels: ONE VALU instruction per element (equal to fmaxf(v, 0) for every non-NaN input)
// one v_max_f32 per element, spelled in asm (every VALU instruction next to f32 MFMAs costs matrix time - see MARL_BURST below).
// ONLY for accumulators already waited for by mfma_settle(): the compiler inserts no MFMA -> VALU wait states in front of inline asm.
__device__ __forceinline__ f4 relu4_settled(f4 v) {
    f4 o;
    asm("v_max_f32 %0, 0, %1" : "=v"(o.x) : "v"(v.x));
    asm("v_max_f32 %0, 0, %1" : "=v"(o.y) : "v"(v.y));
    asm("v_max_f32 %0, 0, %1" : "=v"(o.z) : "v"(v.z));
    asm("v_max_f32 %0, 0, %1" : "=v"(o.w) : "v"(v.w));
    return o;
}

// the operands of the first MFMA group (first layer-1 k-step of every hidden tile + the layer-1 bias in accumulator layout): the
// same registers every time step, so a learner wave keeps them for the whole kernel instead of re-reading them from LDS in front of
// every forward, where nothing hides the read latency
template <class S>
struct FwdHead {
    // (not with VGPR accumulators, which need the registers; nor for wide first layers, whose x / dW1 tiles do)
    static constexpr bool RESIDENT = S::DP <= 16 && !(MARL_MFMA_VGPR_DEFAULT && S::MT == 4);
    f4 op0[S::MT], b1[S::MT];
    __device__ __forceinline__ void load(const float* lds, int lane) {
        const f4* A1 = reinterpret_cast<const f4*>(lds + S::pA1);
#pragma unroll
        for (int mt = 0; mt < S::MT; ++mt) {
            op0[mt] = A1[(mt * (S::KS1 / 4) + 0) * 64 + lane];
            b1[mt] = *reinterpret_cast<const f4*>(lds + S::pb1 + 16 * mt + 4 * (lane >> 4));
        }
    }
};

// MARL_BURST (default 1): VALU work is issued in BURSTS between MFMA groups, never interleaved with them.  On gfx950 the f32-input
// MFMA runs on the vector ALU's own datapath (guide: "at the f32 VECTOR rate"): scripts/mfma_ubench2/3.hip measure that a VALU
// instruction placed between two v_mfma_f32_16x16x4_f32 costs 9-13 cycles of matrix time (nothing is hidden), the same instruction in
// a run of 16-64 costs 4-7, and LDS reads / b32 writes cost ~0.  So: MFMAs back to back, LDS traffic among them, VALU clumped.
#ifndef MARL_BURST
#define MARL_BURST 1
#endif
#if MARL_BURST
#define MARL_VB() __builtin_amdgcn_sched_barrier(0);
#else
#define MARL_VB()
#endif

// WITH_L3 = false stops after the second hidden layer (h2 only): the caller wants ONE output per row (the target value of the
// bootstrap action) and forms it as a dot product (mlp_output_at) instead of 16 output-layer MFMAs that are 10/16 padding
// MARL_MFMA_VGPR (default 1): the MFMA groups whose results the VALU reads next (hidden-layer pre-activations, dH1) are written
// as inline asm with VGPR accumulators.  hipcc gives every MFMA of a kernel that may use more than 256 registers an AGPR
// destination, so each element a relu / mask touches first costs a v_accvgpr_read - and next to f32 MFMAs a VALU instruction is
// never free (MARL_BURST above).  The weight-gradient accumulators, which no VALU instruction reads before the fold, stay on the
// builtin (AGPR) form.  Wait states the compiler does not insert for asm (guide 5.7 item 2): `s_nop 1` in front (a just-written
// VGPR operand), and mfma_settle() - 12 states - between the last MFMA of a chain and the first non-MFMA reader of its result.
#ifndef MARL_MFMA_VGPR
#define MARL_MFMA_VGPR 1
#endif

// acc[mt] += a[mt][e] * b_e for e = 0..3 (outer), mt = 0..3 (inner): 16 MFMAs on four chains, VGPR accumulators
__device__ __forceinline__ void mfma16_v(f4 (&acc)[4], const f4 (&a)[4], float b0, float b1, float b2, float b3) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %4, %20, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %8, %20, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %12, %20, %2\n\t"
        "v_mfma_f32_16x16x4_f32 %3, %16, %20, %3\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %5, %21, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %9, %21, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %13, %21, %2\n\t"
        "v_mfma_f32_16x16x4_f32 %3, %17, %21, %3\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %6, %22, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %10, %22, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %14, %22, %2\n\t"
        "v_mfma_f32_16x16x4_f32 %3, %18, %22, %3\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %7, %23, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %11, %23, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %15, %23, %2\n\t"
        "v_mfma_f32_16x16x4_f32 %3, %19, %23, %3\n\t"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
        : "v"(a[0].x), "v"(a[0].y), "v"(a[0].z), "v"(a[0].w), "v"(a[1].x), "v"(a[1].y), "v"(a[1].z), "v"(a[1].w), "v"(a[2].x), "v"(a[2].y),
          "v"(a[2].z), "v"(a[2].w), "v"(a[3].x), "v"(a[3].y), "v"(a[3].z), "v"(a[3].w), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}
// one quarter of that group: acc[mt] += a_mt * b for mt = 0..3 (4 MFMAs), so that LDS instructions can sit between the quarters -
// next to f32 MFMAs an LDS read or b32 write costs nothing when it issues between two MFMAs, and its full issue time when it is
// clumped in front of the group (scripts/mfma_ubench2.hip)
__device__ __forceinline__ void mfma4_v(f4 (&acc)[4], float a0, float a1, float a2, float a3, float b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %4, %8, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %5, %8, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %6, %8, %2\n\t"
        "v_mfma_f32_16x16x4_f32 %3, %7, %8, %3\n\t"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));
}
// acc0 += a0[r] * b0[r], acc1 += a1[r] * b1[r] for r = 0..3: 8 MFMAs on two chains (the output layer's even / odd k-tiles)
__device__ __forceinline__ void mfma8_v2(f4& acc0, f4& acc1, const f4& a0, const f4& a1, const f4& b0, const f4& b1) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %2, %10, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %6, %14, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %7, %15, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %8, %16, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %9, %17, %1\n\t"
        : "+v"(acc0), "+v"(acc1)
        : "v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w), "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w), "v"(b0.x), "v"(b0.y), "v"(b0.z), "v"(b0.w),
          "v"(b1.x), "v"(b1.y), "v"(b1.z), "v"(b1.w));
}
__device__ __forceinline__ void mfma_settle(f4 (&acc)[4]) {
    asm volatile("s_nop 11" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
}
__device__ __forceinline__ void mfma_settle2(f4& a, f4& b) { asm volatile("s_nop 11" : "+v"(a), "+v"(b)); }

template <class S, bool WITH_L3 = true, class F>
__device__ __forceinline__ void mlp_forward_f(const float* lds, const FwdHead<S>& head, int lane, const float (&x)[S::KS1], f4 (&h1)[S::MT],
                                              f4 (&h2)[S::MT], f4& qa, f4& qb, F&& fill) {
    // groups of 4 MT MFMAs (layer-1 k-steps, layer-2 k-tiles, layer 3), each issued as four quarters; the A operands of group
    // s+1 are requested one hidden tile per quarter of group s.  `fill(k, -1)` - the caller's VALU work for group k - is emitted in
    // front of the group as one burst, `fill(k, e)` (e = 0..3: LDS traffic only) in front of quarter e; the relu of a layer is one
    // burst behind its last group.  Layer 3 runs as two chains (even / odd k-tiles -> qa / qb, added by
    // the caller where it first needs q): a single dependent 16x16x4 chain would cost 40 cycles per MFMA instead of 32.
    constexpr int MT = S::MT, N1 = S::KS1 / 4;
    constexpr bool VG = MARL_MFMA_VGPR && MT == 4;  // asm groups are written for four hidden tiles
    const int g = lane >> 4;
    const f4* A1 = reinterpret_cast<const f4*>(lds + S::pA1);
    const f4* A2 = reinterpret_cast<const f4*>(lds + S::pA2);
    const f4* A3 = reinterpret_cast<const f4*>(lds + S::pA3);
    f4 op[2][MT], acc[MT], nb[MT];
    if (FwdHead<S>::RESIDENT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            op[0][mt] = head.op0[mt];
            acc[mt] = head.b1[mt];
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            op[0][mt] = A1[(mt * N1 + 0) * 64 + lane];
            acc[mt] = *reinterpret_cast<const f4*>(lds + S::pb1 + 16 * mt + 4 * g);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- layer 1
#pragma unroll
    for (int s = 0; s < N1; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        auto next_ops1 = [&](int mt) {
            if (s + 1 < N1) {
                op[nxt][mt] = A1[(mt * N1 + s + 1) * 64 + lane];
            } else {
                op[nxt][mt] = A2[(mt * MT + 0) * 64 + lane];
                nb[mt] = *reinterpret_cast<const f4*>(lds + S::pb2 + 16 * mt + 4 * g);
            }
        };
        if constexpr (!VG) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) next_ops1(mt);
        }
        fill(s, -1);
        MARL_VB()
        if constexpr (VG) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                next_ops1(e);  // one hidden tile of the next group's operands per quarter (MT == 4)
                fill(s, e);
                __builtin_amdgcn_sched_barrier(0);
                mfma4_v(acc, op[cur][0][e], op[cur][1][e], op[cur][2][e], op[cur][3][e], x[4 * s + e]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) fill(s, e);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = MARL_MFMA(op[cur][mt][e], x[4 * s + e], acc[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (VG) mfma_settle(acc);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        h1[mt] = VG ? relu4_settled(acc[mt]) : relu4(acc[mt]);
        acc[mt] = nb[mt];
    }
    // ---- layer 2
    f4 o3a, o3b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k1 = 0; k1 < MT; ++k1) {
        const int cur = (N1 + k1) & 1, nxt = cur ^ 1;
        auto next_ops2 = [&](int mt) {
            if (k1 + 1 < MT) op[nxt][mt] = A2[(mt * MT + k1 + 1) * 64 + lane];
            else if (WITH_L3) op[nxt][mt] = A3[mt * 64 + lane];
            if (WITH_L3 && k1 + 1 == MT && mt == 0) o3a = *reinterpret_cast<const f4*>(lds + S::pb3 + 4 * g);
        };
        if constexpr (!VG) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) next_ops2(mt);
        }
        fill(N1 + k1, -1);
        MARL_VB()
        if constexpr (VG) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                next_ops2(r);
                fill(N1 + k1, r);
                __builtin_amdgcn_sched_barrier(0);
                mfma4_v(acc, op[cur][0][r], op[cur][1][r], op[cur][2][r], op[cur][3][r], h1[k1][r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) fill(N1 + k1, r);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = MARL_MFMA(op[cur][mt][r], h1[k1][r], acc[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (VG) mfma_settle(acc);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) h2[mt] = VG ? relu4_settled(acc[mt]) : relu4(acc[mt]);
    // ---- layer 3
    constexpr int c3 = (N1 + MT) & 1;
    fill(N1 + MT, -1);
#pragma unroll
    for (int r = 0; r < 4; ++r) fill(N1 + MT, r);
    if (WITH_L3) {
        MARL_VB()
        if constexpr (VG) {
            mfma8_v2(o3a, o3b, op[c3][0], op[c3][1], h2[0], h2[1]);
            mfma8_v2(o3a, o3b, op[c3][2], op[c3][3], h2[2], h2[3]);
            mfma_settle2(o3a, o3b);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k1 = 0; k1 < MT; k1 += 2) {
                    o3a = MARL_MFMA(op[c3][k1][r], h2[k1][r], o3a);
                    if (k1 + 1 < MT) o3b = MARL_MFMA(op[c3][k1 + 1][r], h2[k1 + 1][r], o3b);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        qa = o3a;
        qb = o3b;
    }
}

// row `a` of the output layer as the registers of lane (g, .): W3[a][16mt + 4g + r] for mt = 0..MT-1, r = 0..3 - straight out of the
// A3 block of the forward pack (A3[mt][lane = (g, i = a)][r]), one ds_read_b128 per hidden tile with a per-lane address
template <class S>
__device__ __forceinline__ void mlp_w3_row(const float* lds, int lane, int a, f4 (&w)[S::MT]) {
    const f4* A3 = reinterpret_cast<const f4*>(lds + S::pA3);
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < S::MT; ++mt) w[mt] = A3[mt * 64 + g * 16 + a];
}

// output `a` (per row j, a may differ between rows) of the network whose second hidden layer is h2 (C layout): b3[a] + sum_h
// W3[a][h] h2[h], the 16 products of a lane first, then the four lane groups of the row; every lane of row j gets it
template <class S>
__device__ __forceinline__ float mlp_output_at(const float* lds, int lane, int a, const f4 (&h2)[S::MT]) {
    f4 w[S::MT];
    mlp_w3_row<S>(lds, lane, a, w);
    const float b3 = lds[S::pb3 + a];
    float acc = 0.f;
#pragma unroll
    for (int mt = 0; mt < S::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = fmaf(w[mt][r], h2[mt][r], acc);
    return sum_g(acc) + b3;
}

// greedy action of batch row j from q in C layout (lane (g,j) holds Q[4g+r]):
// first index of the maximum (torch.argmax tie rule), identical in all 4 lanes of j.
template <int A>
__device__ __forceinline__ int argmax_rows(const f4& q, int lane, float* best_val = nullptr) {
    const int g = lane >> 4;
    float bv = -__builtin_huge_valf();
    int ba = 0x7FFFFFFF;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a = 4 * g + r;
        if (a < A && (q[r] > bv || ba == 0x7FFFFFFF)) { bv = q[r]; ba = a; }
    }
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oa = __shfl_xor(ba, off);
        if (oa != 0x7FFFFFFF && (ba == 0x7FFFFFFF || ov > bv || (ov == bv && oa < ba))) { bv = ov; ba = oa; }
    }
    if (best_val) *best_val = bv;
    return ba;
}

// value Q[a_sel] of batch row j, gathered from the lane that holds it (all 4 lanes of j get it)
__device__ __forceinline__ float gather_rows(const f4& q, int lane, int a_sel) {
    const int g = lane >> 4;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) v += (4 * g + r == a_sel) ? q[r] : 0.f;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

}  // namespace marl
