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
// (layer 1: for ks: for g: k = 4ks+g) - oracle/mlp_exact.c restates this order.
//
// Canonical parameters of one agent = torch parameters() order of FCNetwork:
//   W1[H][D] b1[H] W2[H][H] b2[H] W3[A][H] b3[A]      (row-major, nn.Linear layout)
#pragma once
#include <hip/hip_runtime.h>

namespace marl {

typedef float f4 __attribute__((ext_vector_type(4)));

#define MARL_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

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

__device__ __forceinline__ f4 relu4(f4 v) {
    f4 o;
    o.x = fmaxf(v.x, 0.f); o.y = fmaxf(v.y, 0.f); o.z = fmaxf(v.z, 0.f); o.w = fmaxf(v.w, 0.f);
    return o;
}

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

// One network on TWO 16-row blocks at once: every A operand fetched from LDS feeds two MFMAs, which halves the LDS
// operand traffic per flop (at hidden 128 a single-block forward reads 327 KB of operands per 320 MFMAs - the LDS port
// and the matrix pipe then run at the same rate).  Same pipelining and per-output summation order as mlp_forward_p.
template <class S>
__device__ __forceinline__ void mlp_forward_p2(const float* lds, int lane, const float (&x)[2][S::KS1], f4 (&q)[2]) {
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
#pragma unroll
    for (int k1 = 0; k1 < MT; ++k1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f4 h20 = relu4(acc[0][k1]), h21 = relu4(acc[1][k1]);
            q[0] = MARL_MFMA(op[c3][k1][r], h20[r], q[0]);
            q[1] = MARL_MFMA(op[c3][k1][r], h21[r], q[1]);
        }
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
