// Recurrent Q-networks (`use_rnn: True`): Linear(D, H) -> ReLU -> one-layer GRU(H, H) -> Linear(H, A)
// (marlbase/utils/models.py:51-116), hidden 64 and 128, on the same transposed-activation f32 MFMA scheme as mlp.h.
//
// One agent's block, parameters() order:
//   first_layer.weight [H][D] | .bias [H] | rnn.weight_ih_l0 [3H][H] | rnn.weight_hh_l0 [3H][H] | rnn.bias_ih_l0 [3H] |
//   rnn.bias_hh_l0 [3H] | final_layer.weight [A][H] | .bias [A]            (gate order r, z, n: torch.nn.GRU)
// A wave owns 16 sequences (batch rows) and walks them one step after the other with the hidden state in registers
// (C layout = the next step's B operand, as everywhere in mlp.h); the whole network (108 KB of A-operand packs at
// D = 15) sits in LDS, so the four waves of a workgroup share nothing but the weights.
//   gru_seq_fwd2_kernel: two such passes (online + target networks) in one launch
//   gru_seq_fwd_kernel : q[t] for t = 0..S-1 from h_in (zeros when NULL), optional h_out, optional per-step activation
//                        record (x1, r, z, n, h, W_hn h + b_hn) for the backward pass
#pragma once
#ifndef MARLHIP_GRU_FWD_DBUF
#define MARLHIP_GRU_FWD_DBUF 1
#endif
#include "common.h"
#include "mlp.h"

namespace marl {

template <int D_, int H_, int A_>
struct GruShape {
    static constexpr int D = D_, H = H_, A = A_;
    static constexpr int DP = (D_ + 15) / 16 * 16, KS1 = DP / 4, MT = H_ / 16;
    static_assert(H_ % 16 == 0 && A_ <= 16, "shape");
    // canonical offsets
    static constexpr int oW1 = 0, ob1 = oW1 + H_ * D_, oWih = ob1 + H_, oWhh = oWih + 3 * H_ * H_, obih = oWhh + 3 * H_ * H_,
                         obhh = obih + 3 * H_, oW3 = obhh + 3 * H_, ob3 = oW3 + A_ * H_;
    static constexpr int NPARAM = ob3 + A_;
    // stacked recurrent layers (`layers = [h] * (L + 1)`: nn.GRU(num_layers = L), marlbase/utils/models.py:74-90): parameters() lists layer
    // l's (weight_ih, weight_hh, bias_ih, bias_hh) behind layer l - 1's, the final layer last.  Every layer runs these same kernels as a
    // one-layer network whose block starts l * LAYER floats into the agent's: its gate matrices and biases then sit at oWih .. obhh, the
    // first layer's W1 at oW1 (l = 0) and the final layer at oW3 (l = L - 1); the parts a layer does not own are packed (in bounds) and unused.
    static constexpr int LAYER = 6 * H_ * H_ + 6 * H_;
    static constexpr int nparam(int layers) { return NPARAM + (layers - 1) * LAYER; }
    // forward pack: A1[MT][KS1/4][64][4] | Gi[3][MT][MT][64][4] | Gh[3][MT][MT][64][4] | A3[MT][64][4] | b1[H] | bih[3H] | bhh[3H] | b3[16]
    static constexpr int pA1 = 0, pGi = pA1 + MT * KS1 * 64, pGh = pGi + 3 * MT * MT * 256, pA3 = pGh + 3 * MT * MT * 256,
                         pb1 = pA3 + MT * 256, pbih = pb1 + H_, pbhh = pbih + 3 * H_, pb3 = pbhh + 3 * H_;
    static constexpr int NFWD = pb3 + 16;
    static_assert(NFWD % 4 == 0, "pack is whole float4s");
    // hidden 64: the whole pack is LDS-resident.  hidden 128: the six H x H gate matrices (393 KB) stream through ONE LDS chunk
    // buffer, a gate at a time, every step (they stay L2-resident); W1, W3 and the biases are resident.
    static constexpr int CHUNK = MT * MT * 256;                  // one gate matrix as an A-operand pack
    static constexpr bool STREAM = NFWD * 4 > 156 * 1024;
    static constexpr int TAIL = NFWD - pA3;                      // A3 + biases
    // streamed LDS layout: A1 | A3 + biases | chunk buffer
    // DBUF: a second chunk buffer where it fits (observation widths up to 32) - the next gate matrix lands in it by LDS-DMA while the
    // MFMAs of the current one run (gru_seq_fwd_body); MARLHIP_GRU_FWD_DBUF=0 at build time keeps the single buffer everywhere
    static constexpr int sA1 = 0, sTail = pGi, sChunk = sTail + TAIL;
    static constexpr bool DBUF = STREAM && MARLHIP_GRU_FWD_DBUF && (sChunk + 2 * CHUNK) * 4 <= 158 * 1024;
    static constexpr int LDS_STREAM = sChunk + (DBUF ? 2 : 1) * CHUNK;
    static constexpr int LDS_FLOATS = STREAM ? LDS_STREAM : NFWD;
    static_assert(LDS_FLOATS * 4 <= 158 * 1024, "recurrent network: LDS budget");
    // per (step, 16-row block) activation record for the backward pass: 6 arrays x MT tiles x 64 lanes x f4
    static constexpr int REC_ARRAYS = 6, REC = REC_ARRAYS * MT * 256;
};

template <class S>
__device__ __forceinline__ float gru_fwd_pack_elem(const float* __restrict__ w, int idx) {
    if (idx < S::pGi) {  // A1[mt][ks4][lane][e]: W1[16mt+i][4(4ks4+e)+g]
        const int e = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
        const int ks4 = rest % (S::KS1 / 4), mt = rest / (S::KS1 / 4);
        const int o = 16 * mt + (lane & 15), k = 4 * (4 * ks4 + e) + (lane >> 4);
        return k < S::D ? w[S::oW1 + o * S::D + k] : 0.f;
    } else if (idx < S::pA3) {  // G?[gate][mt2][mt1][lane][r]: W[gate*H + 16mt2+i][16mt1+4g+r]
        const bool hh = idx >= S::pGh;
        const int j = idx - (hh ? S::pGh : S::pGi);
        const int r = j & 3, lane = (j >> 2) & 63, rest = j >> 8;
        const int mt1 = rest % S::MT, mt2 = (rest / S::MT) % S::MT, gate = rest / (S::MT * S::MT);
        return w[(hh ? S::oWhh : S::oWih) + (gate * S::H + 16 * mt2 + (lane & 15)) * S::H + 16 * mt1 + 4 * (lane >> 4) + r];
    } else if (idx < S::pb1) {  // A3[mt1][lane][r]: W3[i][16mt1+4g+r]
        const int j = idx - S::pA3;
        const int r = j & 3, lane = (j >> 2) & 63, mt1 = j >> 8, o = lane & 15;
        return o < S::A ? w[S::oW3 + o * S::H + 16 * mt1 + 4 * (lane >> 4) + r] : 0.f;
    } else if (idx < S::pbih) {
        return w[S::ob1 + idx - S::pb1];
    } else if (idx < S::pbhh) {
        return w[S::obih + idx - S::pbih];
    } else if (idx < S::pb3) {
        return w[S::obhh + idx - S::pbhh];
    }
    const int o = idx - S::pb3;
    return o < S::A ? w[S::ob3 + o] : 0.f;
}

template <class S>
__global__ __launch_bounds__(256) void gru_pack_kernel(const float* __restrict__ params, AgentMap am, float* __restrict__ packs, int block_floats = S::NPARAM,
                                                       int layer_off = 0) {
    const int p = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < S::NFWD) packs[(size_t)p * S::NFWD + idx] = gru_fwd_pack_elem<S>(params + (size_t)am.net[p] * block_floats + layer_off, idx);
}

// Gate nonlinearities on the hardware transcendental units: v_exp_f32 (2^x) and v_rcp_f32, 1 ulp each - 5 / 6 VALU instructions per
// element instead of the ~22 / ~35 of expf + IEEE division / tanhf.  f32 MFMA and VALU share one pipe and a step has
// 3 H of these per row, so at hidden 64 the library forms cost about a quarter of the forward walk.  Absolute error ~1e-7 (the
// reference's own CPU / CUDA libm differ from each other by as much); saturation is exact (exp -> inf gives 0 / 1 / -1).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ f4 sigmoid4(f4 v) {
    f4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = __builtin_amdgcn_rcpf(1.f + fast_exp(-v[r]));
    return o;
}

__device__ __forceinline__ f4 tanh4(f4 v) {
    f4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + fast_exp(2.f * v[r]));
    return o;
}

// gate pre-activations of one gate: out[mt] = bias[16mt+4g..] + sum_{mt1, r} G[gate][mt][mt1][lane][r] * b[mt1][r]
template <class S>
__device__ __forceinline__ void gru_gate(const f4* G /* [MT][MT][64] chunk */, const float* bias /* [H] of this gate */, int lane,
                                         const f4 (&b)[S::MT], f4 (&out)[S::MT]) {
    constexpr int MT = S::MT;
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) out[mt] = *reinterpret_cast<const f4*>(bias + 16 * mt + 4 * g);
#pragma unroll
    for (int k1 = 0; k1 < MT; ++k1) {
        f4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = G[(mt * MT + k1) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) out[mt] = MARL_MFMA(a[mt][r], b[k1][r], out[mt]);
    }
}

// obs: row (t, b) of agent p at obs + p * obs_as + (t * B + b) * obs_rs (dqn/train.py Batch [P][S][B][D]: as = S*B*D, rs = D; the
// ac/train.py Batch [S][B][P*D]: as = D, rs = P*D); q: [P][S][B][A], h_in / h_out: [P][B][H] or NULL,
// rec: [P][S][nblk][REC] or NULL (nblk = ceil(B / 16))
// x_in (stacked layers, layer l >= 1): the record of the layer below - this layer's input x1[t] is that layer's h[t] (no first linear
// layer, no ReLU); q_out NULL: no final layer (every layer but the last)
template <class S>
__device__ __forceinline__ void gru_seq_fwd_body(const float* __restrict__ packs, const float* __restrict__ obs, size_t obs_as, size_t obs_rs,
                                                 int steps, int B, const float* __restrict__ h_in, float* __restrict__ h_out,
                                                 float* __restrict__ q_out, float* __restrict__ rec, const float* __restrict__ x_in = nullptr) {
    constexpr int MT = S::MT, D = S::D, H = S::H, A = S::A;
    constexpr bool STREAM = S::STREAM;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int p = blockIdx.y;
    const float* pack = packs + (size_t)p * S::NFWD;
    if (!STREAM) {
        copy_f4_to_lds(reinterpret_cast<const f4*>(pack), reinterpret_cast<f4*>(lds), S::NFWD / 4, tid, 256);
    } else {
        copy_f4_to_lds(reinterpret_cast<const f4*>(pack + S::pA1), reinterpret_cast<f4*>(lds + S::sA1), S::pGi / 4, tid, 256);
        copy_f4_to_lds(reinterpret_cast<const f4*>(pack + S::pA3), reinterpret_cast<f4*>(lds + S::sTail), S::TAIL / 4, tid, 256);
    }
    __syncthreads();
    // LDS addresses of the resident parts
    const float* lA1 = lds + (STREAM ? S::sA1 : S::pA1);
    const float* tail = lds + (STREAM ? S::sTail : S::pA3);  // A3 | b1 | bih | bhh | b3
    const float* lA3 = tail;
    const float* lb1 = tail + (S::pb1 - S::pA3);
    const float* lbih = tail + (S::pbih - S::pA3);
    const float* lbhh = tail + (S::pbhh - S::pA3);
    const float* lb3 = tail + (S::pb3 - S::pA3);
    const int nblk = (B + 15) >> 4;
    const int blk = blockIdx.x * 4 + wave;
    const bool active = blk < nblk;
    if (!STREAM && !active) return;  // streamed: every wave keeps staging and meeting the barriers
    const int b0 = (active ? blk : nblk - 1) * 16;
    const bool rowok = active && b0 + j < B;
    const int bj = b0 + j < B ? b0 + j : B - 1;
    f4 h[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (h_in != nullptr) h[mt] = *reinterpret_cast<const f4*>(h_in + ((size_t)p * B + bj) * H + 16 * mt + 4 * g);
        else h[mt] = f4{0.f, 0.f, 0.f, 0.f};
    }
    const f4* A1 = reinterpret_cast<const f4*>(lA1);
    const f4* A3 = reinterpret_cast<const f4*>(lA3);
    // DBUF: the i-th gate product of a step (matrices in the order 0, 3, 1, 4, 2, 5) reads chunk buffer i & 1.  A wave requests its 16
    // one-KB pieces of a matrix with global_load_lds (LDS address = wave-uniform base + 16 * lane); they are ordered for the readers
    // by the issuing wave's vmcnt(0) followed by the workgroup barrier in front of the product that uses them, and a buffer is
    // requested again only behind the barrier that follows its last reader.
    constexpr bool DBUF = S::DBUF;
    int t_now = 0;
    auto request = [&](int c, int buf) {
        const char* src = reinterpret_cast<const char*>(pack + S::pGi + (size_t)c * S::CHUNK) + 16 * lane;
        char* dst = reinterpret_cast<char*>(lds + S::sChunk + (size_t)buf * S::CHUNK);
#pragma unroll 1
        for (int k = 0; k < S::CHUNK * 4 / 1024 / 4; ++k) {
            const int piece = 4 * k + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 1024 * piece),
                                             (__attribute__((address_space(3))) void*)(dst + 1024 * piece), 16, 0, 0);
        }
    };
    if (DBUF && steps > 0) request(0, 0);
    // gate matrix `c` (0..2: W_ih r, z, n; 3..5: W_hh r, z, n) as an A-operand chunk in LDS; `i` = its position in the step
    auto gate_chunk = [&](int c, int i) -> const f4* {
        if (!STREAM) return reinterpret_cast<const f4*>(lds + S::pGi) + (size_t)c * MT * MT * 64;
        if (DBUF) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of matrix c have landed
            __syncthreads();                                     // everybody's have; everybody is done with the other buffer
            const int nxt = i == 0 ? 3 : (i == 1 ? 1 : (i == 2 ? 4 : (i == 3 ? 2 : (i == 4 ? 5 : 0))));
            if (i < 5 || t_now + 1 < steps) request(nxt, (i + 1) & 1);  // (nothing may be in flight when the workgroup ends)
            return reinterpret_cast<const f4*>(lds + S::sChunk + (size_t)(i & 1) * S::CHUNK);
        }
        __syncthreads();  // everybody is done with the previous chunk
        copy_f4_to_lds(reinterpret_cast<const f4*>(pack + S::pGi) + (size_t)c * MT * MT * 64, reinterpret_cast<f4*>(lds + S::sChunk), S::CHUNK / 4,
                       tid, 256);
        __syncthreads();
        return reinterpret_cast<const f4*>(lds + S::sChunk);
    };
    for (int t = 0; t < steps; ++t) {
        t_now = t;
        asm volatile("" ::: "memory");  // the packs never change, so the compiler would hoist every weight read out of the time
                                        // loop (and spill ~1 KB per lane): re-read them from LDS each step
        f4 x1[MT];
        if (x_in != nullptr) {
            const f4* Rb = reinterpret_cast<const f4*>(x_in + (((size_t)p * steps + t) * nblk + (active ? blk : nblk - 1)) * S::REC);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) x1[mt] = Rb[(4 * MT + mt) * 64 + lane];
        } else {
        const float* xrow = obs + (size_t)p * obs_as + ((size_t)t * B + bj) * obs_rs;  // row (t, b) of agent p
        float x[S::KS1];
#pragma unroll
        for (int ks = 0; ks < S::KS1; ++ks) {
            const int d = 4 * ks + g;
            x[ks] = (d < D && rowok) ? xrow[d < D ? d : D - 1] : 0.f;
        }
        // x1 = relu(W1 x + b1)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) x1[mt] = *reinterpret_cast<const f4*>(lb1 + 16 * mt + 4 * g);
#pragma unroll
        for (int ks4 = 0; ks4 < S::KS1 / 4; ++ks4) {
            f4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = A1[(mt * (S::KS1 / 4) + ks4) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) x1[mt] = MARL_MFMA(a[mt][e], x[4 * ks4 + e], x1[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) x1[mt] = relu4(x1[mt]);
        }
        // gates (torch.nn.GRU): r, z, n
        f4 gi[MT], gh[MT], rg[MT], zg[MT], ng[MT], ghn[MT];
        gru_gate<S>(gate_chunk(0, 0), lbih + 0 * H, lane, x1, gi);
        gru_gate<S>(gate_chunk(3, 1), lbhh + 0 * H, lane, h, gh);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) rg[mt] = sigmoid4(gi[mt] + gh[mt]);
        gru_gate<S>(gate_chunk(1, 2), lbih + 1 * H, lane, x1, gi);
        gru_gate<S>(gate_chunk(4, 3), lbhh + 1 * H, lane, h, gh);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) zg[mt] = sigmoid4(gi[mt] + gh[mt]);
        gru_gate<S>(gate_chunk(2, 4), lbih + 2 * H, lane, x1, gi);
        gru_gate<S>(gate_chunk(5, 5), lbhh + 2 * H, lane, h, ghn);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ng[mt] = tanh4(gi[mt] + rg[mt] * ghn[mt]);
            h[mt] = (1.f - zg[mt]) * ng[mt] + zg[mt] * h[mt];
        }
        if (rec != nullptr && active) {
            f4* R = reinterpret_cast<f4*>(rec + (((size_t)p * steps + t) * nblk + blk) * S::REC);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                R[(0 * MT + mt) * 64 + lane] = x1[mt];
                R[(1 * MT + mt) * 64 + lane] = rg[mt];
                R[(2 * MT + mt) * 64 + lane] = zg[mt];
                R[(3 * MT + mt) * 64 + lane] = ng[mt];
                R[(4 * MT + mt) * 64 + lane] = h[mt];
                R[(5 * MT + mt) * 64 + lane] = ghn[mt];
            }
        }
        if (q_out == nullptr) continue;  // (uniform: a layer below the last)
        // q = W3 h + b3
        f4 q = *reinterpret_cast<const f4*>(lb3 + 4 * g);
#pragma unroll
        for (int k1 = 0; k1 < MT; ++k1) {
            const f4 a = A3[k1 * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) q = MARL_MFMA(a[r], h[k1][r], q);
        }
        if (rowok) {
            float* qo = q_out + (((size_t)p * steps + t) * B + b0 + j) * A;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < A) qo[4 * g + r] = q[r];
        }
    }
    if (h_out != nullptr && rowok) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f4*>(h_out + ((size_t)p * B + b0 + j) * H + 16 * mt + 4 * g) = h[mt];
    }
}

template <class S>
__global__ __launch_bounds__(256) void gru_seq_fwd_kernel(const float* __restrict__ packs, const float* __restrict__ obs, size_t obs_as,
                                                          size_t obs_rs, int steps, int B, const float* __restrict__ h_in,
                                                          float* __restrict__ h_out, float* __restrict__ q_out, float* __restrict__ rec,
                                                          const float* __restrict__ x_in = nullptr) {
    gru_seq_fwd_body<S>(packs, obs, obs_as, obs_rs, steps, B, h_in, h_out, q_out, rec, x_in);
}

// Two independent passes over the same observations in one launch (blockIdx.z): the online networks (with the activation record) and
// the target networks.  One pass of B = 4096 sequences occupies 512 of the 1024 SIMDs and a sequence cannot be split, so the pair
// costs what one of them does.  Zero initial hidden states.
template <class S>
__global__ __launch_bounds__(256) void gru_seq_fwd2_kernel(const float* __restrict__ packs, const float* __restrict__ packs2,
                                                           const float* __restrict__ obs, size_t obs_as, size_t obs_rs, int steps, int steps2, int B,
                                                           float* __restrict__ q_out, float* __restrict__ q_out2, float* __restrict__ rec,
                                                           float* __restrict__ rec_second = nullptr, const float* __restrict__ x_in = nullptr,
                                                           const float* __restrict__ x_in2 = nullptr) {
    // rec_second / x_in / x_in2: stacked layers - the second pass keeps a record too where a layer above reads its h[t] from it
    const bool second = blockIdx.z == 1;
    gru_seq_fwd_body<S>(second ? packs2 : packs, obs, obs_as, obs_rs, second ? steps2 : steps, B, nullptr, nullptr, second ? q_out2 : q_out,
                        second ? rec_second : rec, second ? x_in2 : x_in);
}

}  // namespace marl
