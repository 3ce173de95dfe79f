// OPT-IN experiment (C-ABI 209, marlhip_*_split16): the independent learner's loss / gradient step with every f32 product formed from
// fp16 halves on the double-rate matrix pipe, fp32 accumulate - a DEVIATION from the exact-f32 default kernels of
// dqn_update_kernels.h, kept out of every default path and gated by the reference's goldens at the default tolerances
// (tests/test_gpu_split16.py).  Replaces the same reference code: QNetwork._compute_loss + loss.backward(), dqn/model.py:118-168.
//
// Why: v_mfma_f32_16x16x4_f32 issues at the f32 VECTOR rate (32 cycles per 16x16x4 on a SIMD, 157 TFLOP/s dense) and shares the
// vector ALU, so the f32 learner sits at 0.58 of that peak with nothing left to schedule.  v_mfma_f32_16x16x32_f16
// does 8x the work in ~17 cycles, and - measured, scripts/mfma_ubench5.hip - VALU instructions interleave with it at ~1.4 cycles
// apiece instead of ~7.
//
// How a product keeps f32 accuracy: x = xh + 2^-11 xl with xh = the top 11 significand bits of x (exact in fp16) and
// xl = fp16((x - xh) * 2^11) (the residual is exact in f32; scaled up so that it stays a NORMAL fp16 for |x| >= 2^-14 - unscaled
// it would fall into fp16's subnormals for |x| < 1/8 and lose its low bits).  Then
//     w * x = wh xh + 2^-11 (wh xl + wl xh) + 2^-22 wl xl          (last term dropped: relative 2^-22)
// with every kept product EXACT in the fp32 accumulator (11 + 11 bits).  Three MFMAs per K = 32 group, two accumulators (main,
// cross), one FMA per output element to combine: value = main + 2^-11 cross.  Weights are split once per update by the pack
// kernel, activations on the fly (8 VALU instructions per pair of values).  |x| >= 65504 overflows fp16: the observations,
// hidden activations and gradients of this path are far inside that range; an overflow shows up as inf / nan in the loss.
//
// Structure: the work decomposition, the backwards walk through time, the one-hot shortcut for dH2 and the partial-record format of
// dqn_lossgrad_kernel (so dqn_reduce_sq_kernel / the clip + Adam launches are shared), with LDS-resident fp16 packs.  IDQN
// (mode 0) only, two hidden layers of 64, observation width <= 32, no action masks; written for clarity first - the compiler
// schedules it.  MFMA operand layouts: a 16x16x32 MFMA contracts over (lane group g, element e) pairs identically for A and B, so
// K index (g, e) of k-group kg is DEFINED as hidden unit 16 (2 kg + (e >> 2)) + 4 g + (e & 3): the C/D layout of one layer (lane (g, j)
// holds outputs 16 mt + 4 g + r of row j) is then the B operand of the next with no cross-lane movement, as in mlp.h.
#pragma once
#include "dqn_update_kernels.h"

namespace marl {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define MARL_MFMA_H32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define MARL_MFMA_H16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16((a), (b), (c), 0, 0, 0)

constexpr float H16_SCALE = 2048.f;           // 2^11: the low half is kept scaled up
constexpr float H16_UNSCALE = 1.f / 2048.f;

// x -> (top 11 significand bits, the scaled residual); both exactly representable in fp16 for 2^-14 <= |x| < 65504
__device__ __forceinline__ void h16_split1(float x, float& hi, float& lo) {
    hi = __uint_as_float(__float_as_uint(x) & 0xFFFFE000u);
    lo = (x - hi) * H16_SCALE;
}
__device__ __forceinline__ h2 h16_pk(float a, float b) { return h2{(_Float16)a, (_Float16)b}; }  // round to nearest even (v_cvt_pk_f16_f32 on gfx950)

__device__ __forceinline__ void h16_split8(const f4& a, const f4& b, h8& hi, h8& lo) {  // elements e < 4 from a, e >= 4 from b
    float fh[8], fl[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { h16_split1(a[e], fh[e], fl[e]); h16_split1(b[e], fh[4 + e], fl[4 + e]); }
    h2 p[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { p[k] = h16_pk(fh[2 * k], fh[2 * k + 1]); q[k] = h16_pk(fl[2 * k], fl[2 * k + 1]); }
    hi = h8{p[0][0], p[0][1], p[1][0], p[1][1], p[2][0], p[2][1], p[3][0], p[3][1]};
    lo = h8{q[0][0], q[0][1], q[1][0], q[1][1], q[2][0], q[2][1], q[3][0], q[3][1]};
}
__device__ __forceinline__ void h16_split4(const f4& a, h4& hi, h4& lo) {
    float fh[4], fl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) h16_split1(a[e], fh[e], fl[e]);
    const h2 p0 = h16_pk(fh[0], fh[1]), p1 = h16_pk(fh[2], fh[3]), q0 = h16_pk(fl[0], fl[1]), q1 = h16_pk(fl[2], fl[3]);
    hi = h4{p0[0], p0[1], p1[0], p1[1]};
    lo = h4{q0[0], q0[1], q1[0], q1[1]};
}

// pack of one agent, in floats (every region 16-byte aligned; fp16 regions hold two halves per float slot):
//   per network (critic, target): A1h A1l [MT][64][8] | A2h A2l [MT][KG][64][8] | A3h A3l [KG][64][8] | b1[H] b2[H] b3[16] | W3f[16][H]
//   critic only: T2h T2l [MT][KG][64][8]   (A operand of dH1 = W2^T dH2: row m = h1 unit, K = h2 units)
template <class S>
struct H16Pack {
    static constexpr int MT = S::MT, KG = S::H / 32, H = S::H;
    static_assert(S::H % 32 == 0 && S::D <= 32 && S::A <= 16, "split16 learner: hidden a multiple of 32, observation width <= 32");
    static constexpr int nA1 = MT * 64 * 8 / 2, nA2 = MT * KG * 64 * 8 / 2, nA3 = KG * 64 * 8 / 2;  // float slots of ONE half-array
    static constexpr int oA1h = 0, oA1l = oA1h + nA1, oA2h = oA1l + nA1, oA2l = oA2h + nA2, oA3h = oA2l + nA2, oA3l = oA3h + nA3;
    static constexpr int ob1 = oA3l + nA3, ob2 = ob1 + H, ob3 = ob2 + H, oW3f = ob3 + 16, NNET = oW3f + 16 * H;
    static constexpr int oT2h = 2 * NNET, oT2l = oT2h + nA2, TOTAL = oT2l + nA2;
    static_assert(NNET % 4 == 0 && TOTAL % 4 == 0, "16-byte regions");
    __host__ __device__ static constexpr int unit(int kg, int g, int e) { return 16 * (2 * kg + (e >> 2)) + 4 * g + (e & 3); }
};

// one pack element (a HALF index inside the fp16 part is handled by the caller): value of weight for (region, position)
template <class S>
__global__ __launch_bounds__(256) void h16_pack_kernel(const float* __restrict__ params, const float* __restrict__ tparams, AgentMap am,
                                                       float* __restrict__ packs) {
    using K = H16Pack<S>;
    constexpr int MT = K::MT, KG = K::KG, H = S::H, D = S::D, A = S::A;
    const int p = blockIdx.y;
    float* out = packs + (size_t)p * K::TOTAL;
    _Float16* outh = reinterpret_cast<_Float16*>(out);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    // thread idx covers: for each net, the hi-arrays' half elements (the lo array is written by the same thread), then f32 parts
    constexpr int HALVES_NET = 2 * (K::nA1 + K::nA2 + K::nA3);  // halves in ONE (hi) set of a net
    constexpr int F32_NET = 2 * H + 16 + 16 * H;
    constexpr int PER_NET = HALVES_NET + F32_NET, T2_HALVES = 2 * K::nA2;
    if (idx >= 2 * PER_NET + T2_HALVES) return;
    auto put = [&](int slot_h, int slot_l, int half_idx, float w) {
        float hi, lo;
        h16_split1(w, hi, lo);
        outh[2 * slot_h + half_idx] = (_Float16)hi;
        outh[2 * slot_l + half_idx] = (_Float16)lo;
    };
    if (idx < 2 * PER_NET) {
        const int net = idx / PER_NET, k = idx - net * PER_NET;
        const float* w = (net == 0 ? params : tparams) + (size_t)am.net[p] * S::NPARAM;
        const int base = net * K::NNET;
        if (k < 2 * K::nA1) {  // A1[mt][lane][e]: W1[16 mt + i][8 g + e]
            const int e = k & 7, lane = (k >> 3) & 63, mt = k >> 9;
            const int o = 16 * mt + (lane & 15), d = 8 * (lane >> 4) + e;
            put(base + K::oA1h, base + K::oA1l, k, d < D ? w[S::oW1 + o * D + d] : 0.f);
        } else if (k < 2 * (K::nA1 + K::nA2)) {  // A2[mt2][kg][lane][e]: W2[16 mt2 + i][unit(kg, g, e)]
            const int q = k - 2 * K::nA1;
            const int e = q & 7, lane = (q >> 3) & 63, rest = q >> 9, kg = rest % KG, mt2 = rest / KG;
            put(base + K::oA2h, base + K::oA2l, q, w[S::oW2 + (16 * mt2 + (lane & 15)) * H + K::unit(kg, lane >> 4, e)]);
        } else if (k < HALVES_NET) {  // A3[kg][lane][e]: W3[i][unit(kg, g, e)], rows i >= A zero
            const int q = k - 2 * (K::nA1 + K::nA2);
            const int e = q & 7, lane = (q >> 3) & 63, kg = q >> 9, i = lane & 15;
            put(base + K::oA3h, base + K::oA3l, q, i < A ? w[S::oW3 + i * H + K::unit(kg, lane >> 4, e)] : 0.f);
        } else {
            const int q = k - HALVES_NET;
            float v;
            if (q < H) v = w[S::ob1 + q];
            else if (q < 2 * H) v = w[S::ob2 + q - H];
            else if (q < 2 * H + 16) v = (q - 2 * H) < A ? w[S::ob3 + q - 2 * H] : 0.f;
            else {
                const int r = q - 2 * H - 16, a = r / H, h = r - a * H;
                v = a < A ? w[S::oW3 + a * H + h] : 0.f;
            }
            out[base + K::ob1 + q] = v;
        }
    } else {  // T2[mt1][kg][lane][e]: A[m = h1 unit 16 mt1 + i][k = h2 unit(kg, g, e)] = W2[unit][16 mt1 + i]
        const int q = idx - 2 * PER_NET;
        const float* w = params + (size_t)am.net[p] * S::NPARAM;
        const int e = q & 7, lane = (q >> 3) & 63, rest = q >> 9, kg = rest % KG, mt1 = rest / KG;
        put(K::oT2h, K::oT2l, q, w[S::oW2 + K::unit(kg, lane >> 4, e) * H + 16 * mt1 + (lane & 15)]);
    }
}

template <class S>
struct H16Lds {
    using K = H16Pack<S>;
    static constexpr int TS = 20;  // tile row stride (floats): ds_read_b128 of 16 lanes x 4 groups conflict-free
    static constexpr int TILE = TS * S::H, PER_WAVE = 3 * TILE + 16 * TS;  // h2 (later dH1) | h1 | dH2 | dQ
    static constexpr int REC = S::NPARAM + 2;
    static constexpr int walk = K::TOTAL + 4 * PER_WAVE, fold = 4 * REC;
    static constexpr int total = walk > fold ? walk : fold;
    static constexpr bool FITS = total * 4 <= 160 * 1024;
};

template <class S, bool REPLAY>
__global__ __launch_bounds__(256, 1) void dqn_lossgrad_h16_kernel(const float* __restrict__ packs, marlhip_batch bt, ReplaySrc rs, float gamma,
                                                                   int double_q, int n_chunks, float* __restrict__ partials) {
    using K = H16Pack<S>;
    using L = H16Lds<S>;
    constexpr int MT = S::MT, KG = K::KG, NT1 = S::DP / 16, D = S::D, H = S::H, A = S::A, TS = L::TS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // the wave index as a SCALAR: task, chunk bounds, the time step and every address term derived from them then live in SGPRs - uniform
    // branches become s_cbranch_scc instead of exec-mask dances, the per-step address arithmetic runs on the scalar ALU
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int p = blockIdx.y, P = gridDim.y;
    const int T = bt.max_len, B = bt.batch;
    copy_f4_to_lds(reinterpret_cast<const f4*>(packs + (size_t)p * K::TOTAL), reinterpret_cast<f4*>(lds), K::TOTAL / 4, tid, 256);
    __syncthreads();
    float* TH2 = lds + K::TOTAL + wave * L::PER_WAVE;
    float* TH1 = TH2 + L::TILE;
    float* TG2 = TH1 + L::TILE;
    float* TQ = TG2 + L::TILE;
    float* TG1 = TH2;  // the dH1 tile re-uses the h2 tile (its only reader, dW3, has run by then)
    const h8* T2h = reinterpret_cast<const h8*>(lds + K::oT2h);
    const h8* T2l = reinterpret_cast<const h8*>(lds + K::oT2l);

    const size_t obs_as = bt.obs_agent_stride > 0 ? (size_t)bt.obs_agent_stride : (bt.obs_agent_stride < 0 ? 0 : (size_t)(T + 1) * B * D);
    const size_t obs_rs = bt.obs_row_stride ? (size_t)bt.obs_row_stride : (size_t)D;
    const float* obs_p = REPLAY ? nullptr : bt.obss + (size_t)p * obs_as;
    const int64_t* act_p = REPLAY ? nullptr : bt.actions + (size_t)p * T * B;
    const float* rew_p = REPLAY ? nullptr : bt.rewards + (size_t)p * T * B;

    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // weight-gradient accumulators: main and cross (value = main + 2^-11 cross)
    f4 dW1m[MT][NT1], dW1c[MT][NT1], dW2m[MT][MT], dW2c[MT][MT], dW3m[MT], dW3c[MT], db1[MT], db2[MT], db3 = zero4;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        dW3m[a] = zero4; dW3c[a] = zero4; db1[a] = zero4; db2[a] = zero4;
#pragma unroll
        for (int b = 0; b < MT; ++b) { dW2m[a][b] = zero4; dW2c[a][b] = zero4; }
#pragma unroll
        for (int b = 0; b < NT1; ++b) { dW1m[a][b] = zero4; dW1c[a][b] = zero4; }
    }
    float loss_acc = 0.f, nfill_acc = 0.f;

    // forward of NN networks (1: the critic; 2: critic and target side by side) on the wave's 16 rows: x split (B operand of layer 1)
    // -> h1, h2 (C layout, f32, post-relu; only network 0's are kept), q[n].  MFMA order inside a layer: every tile's main product,
    // then every tile's first cross product, then the second - consecutive MFMAs never touch the same accumulator (a dependent pair
    // back to back waits out the whole pipeline with one wave per SIMD), and the two networks' chains give the scheduler independent
    // vector work (relu, splits) to place under the other network's MFMAs.
    auto forward = [&](auto nn_c, const float* lds, const h8& xh, const h8& xl, f4 (&h1)[MT], f4 (&h2)[MT], f4 (&q)[2]) {  // `lds`: the first network's pack
        constexpr int NN = decltype(nn_c)::value;
        f4 m[NN][MT], c[NN][MT];
        h8 bh[NN][KG], bl[NN][KG];
        // (Measured alternatives, gpurun r3C: requesting a (layer, k-group)'s A tiles as one batch of reads before its MFMAs spills 36
        // registers and runs 90 us against 85; deferring dW2 / dW1 into the next iteration next to the forward pass, 92 us.)
        // ---- layer 1
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float* pk = lds + n * K::NNET;
                m[n][mt] = MARL_MFMA_H32(reinterpret_cast<const h8*>(pk + K::oA1h)[mt * 64 + lane], xh,
                                         *reinterpret_cast<const f4*>(pk + K::ob1 + 16 * mt + 4 * g));
            }
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) c[n][mt] = MARL_MFMA_H32(reinterpret_cast<const h8*>(lds + n * K::NNET + K::oA1h)[mt * 64 + lane], xl, zero4);
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) c[n][mt] = MARL_MFMA_H32(reinterpret_cast<const h8*>(lds + n * K::NNET + K::oA1l)[mt * 64 + lane], xh, c[n][mt]);
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            f4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                a[mt] = relu4(m[n][mt] + c[n][mt] * H16_UNSCALE);
                if (n == 0) h1[mt] = a[mt];
            }
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) h16_split8(a[2 * kg], a[2 * kg + 1], bh[n][kg], bl[n][kg]);
        }
        // ---- layer 2
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                m[n][mt] = *reinterpret_cast<const f4*>(lds + n * K::NNET + K::ob2 + 16 * mt + 4 * g);
                c[n][mt] = zero4;
            }
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
#pragma unroll
            for (int n = 0; n < NN; ++n)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    m[n][mt] = MARL_MFMA_H32(reinterpret_cast<const h8*>(lds + n * K::NNET + K::oA2h)[(mt * KG + kg) * 64 + lane], bh[n][kg], m[n][mt]);
#pragma unroll
            for (int n = 0; n < NN; ++n)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    c[n][mt] = MARL_MFMA_H32(reinterpret_cast<const h8*>(lds + n * K::NNET + K::oA2h)[(mt * KG + kg) * 64 + lane], bl[n][kg], c[n][mt]);
#pragma unroll
            for (int n = 0; n < NN; ++n)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    c[n][mt] = MARL_MFMA_H32(reinterpret_cast<const h8*>(lds + n * K::NNET + K::oA2l)[(mt * KG + kg) * 64 + lane], bh[n][kg], c[n][mt]);
        }
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            f4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                a[mt] = relu4(m[n][mt] + c[n][mt] * H16_UNSCALE);
                if (n == 0) h2[mt] = a[mt];
            }
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) h16_split8(a[2 * kg], a[2 * kg + 1], bh[n][kg], bl[n][kg]);
        }
        // ---- output layer (one tile per network: the two k-groups and the two networks alternate)
        f4 qm[NN], qc[NN];
#pragma unroll
        for (int n = 0; n < NN; ++n) { qm[n] = *reinterpret_cast<const f4*>(lds + n * K::NNET + K::ob3 + 4 * g); qc[n] = zero4; }
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
#pragma unroll
            for (int n = 0; n < NN; ++n) qm[n] = MARL_MFMA_H32(reinterpret_cast<const h8*>(lds + n * K::NNET + K::oA3h)[kg * 64 + lane], bh[n][kg], qm[n]);
#pragma unroll
            for (int n = 0; n < NN; ++n) qc[n] = MARL_MFMA_H32(reinterpret_cast<const h8*>(lds + n * K::NNET + K::oA3h)[kg * 64 + lane], bl[n][kg], qc[n]);
#pragma unroll
            for (int n = 0; n < NN; ++n) qc[n] = MARL_MFMA_H32(reinterpret_cast<const h8*>(lds + n * K::NNET + K::oA3l)[kg * 64 + lane], bh[n][kg], qc[n]);
        }
#pragma unroll
        for (int n = 0; n < NN; ++n) q[n] = qm[n] + qc[n] * H16_UNSCALE;
        // (Scheduling hints for this block - iglp_opt(0 / 1), sched_group_barrier patterns of 1 MFMA : 1 LDS read : 5 VALU and 2 : 2 : 8 -
        // measured 88.3 / 98.2 / 90.6 / 89.6 us against 88.0 us without, gpurun r3E: none kept.)
    };

    struct Rows {
        float x[8];        // layer-1 B operand: X[row j][8 g + e]
        float bx[NT1][4];  // dW1 B operand:     X[row 4 g + e][16 nt + j]
        int a_sel;
        float rw, dn, fl;
    };
    const int ngroups = (B + 15) >> 4;
    const int ntasks = ngroups * n_chunks;
    for (int task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
        const int grp = task / n_chunks, c = task - grp * n_chunks;
        const int t0 = (c * T) / n_chunks, t1 = ((c + 1) * T) / n_chunks;
        if (t1 <= t0) continue;
        const int b0 = grp * 16;
        const bool rowok = (b0 + j) < B;
        const int bj = rowok ? b0 + j : B - 1;
        int ej = 0, eg[4] = {0, 0, 0, 0};
        if (REPLAY) {
            ej = rs.idx ? rs.idx[bj] : replay_draw(rs, bj);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = b0 + 4 * g + e, rc = row < B ? row : B - 1;
                eg[e] = rs.idx ? rs.idx[rc] : replay_draw(rs, rc);
            }
            if (rs.idx_out != nullptr && p == 0 && c == 0 && g == 0 && rowok) rs.idx_out[bj] = ej;
        }
        // per-task bases (vector part of every address, formed ONCE); a step adds a scalar multiple of t
        const float* xb = REPLAY ? rs.rb.obs + ((size_t)ej * P + p) * (T + 1) * D : obs_p + (size_t)bj * obs_rs;  // + t * (D | B * obs_rs)
        const float* bxb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = b0 + 4 * g + e;
            bxb[e] = REPLAY ? rs.rb.obs + ((size_t)eg[e] * P + p) * (T + 1) * D : obs_p + (size_t)(row < B ? row : B - 1) * obs_rs;
        }
        const size_t xstep = REPLAY ? (size_t)D : (size_t)B * obs_rs;
        int xoff[8], bxoff[NT1];
#pragma unroll
        for (int e = 0; e < 8; ++e) xoff[e] = (8 * g + e) < D ? 8 * g + e : D - 1;
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) bxoff[nt] = (16 * nt + j) < D ? 16 * nt + j : D - 1;
        const uint8_t* actb = REPLAY ? rs.rb.act + ((size_t)ej * P + p) * T : nullptr;
        const float* rewb = REPLAY ? rs.rb.rew + ((size_t)ej * P + p) * T : rew_p + bj;
        const uint8_t* doneb = REPLAY ? rs.rb.done + (size_t)ej * (T + 1) + 1 : nullptr;
        const uint8_t* fillb = REPLAY ? rs.rb.filled + (size_t)ej * T : nullptr;
        auto load_rows = [&](int t, Rows& R) {  // every address clamped in range, loads unconditional; masks at the point of use
            const int tt = t < T ? t : T - 1;
            const float* xrow = xb + (size_t)t * xstep;
#pragma unroll
            for (int e = 0; e < 8; ++e) R.x[e] = xrow[xoff[e]];
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) R.bx[nt][e] = (bxb[e] + (size_t)t * xstep)[bxoff[nt]];
            if (REPLAY) {
                R.a_sel = (int)actb[tt];
                R.rw = rewb[tt];
                R.dn = doneb[tt] ? 1.f : 0.f;
                R.fl = fillb[tt] ? 1.f : 0.f;
            } else {
                R.a_sel = (int)act_p[(size_t)tt * B + bj];
                R.rw = rewb[(size_t)tt * B];
                R.dn = bt.dones[(size_t)(tt + 1) * B + bj];
                R.fl = bt.filled[(size_t)tt * B + bj];
            }
        };
        float tq_next = 0.f;
        Rows cur;
        load_rows(t1, cur);
        for (int t = t1; t >= t0; --t) {
            const bool first = t == t1, last = t == t0;
            Rows nxt;
            if (!last) load_rows(t - 1, nxt);
            // zero the padding of the operands (features >= D, rows >= B) where it matters: x feeds zero-padded weights (finite x is
            // enough), bx feeds dW1 columns the fold drops and rows with dH1 = 0
            const float fl = rowok ? cur.fl : 0.f;
            h8 xh, xl;
            {
                f4 xa = {cur.x[0], cur.x[1], cur.x[2], cur.x[3]}, xb = {cur.x[4], cur.x[5], cur.x[6], cur.x[7]};
#pragma unroll
                for (int e = 0; e < 4; ++e) { xa[e] = (8 * g + e) < D ? xa[e] : 0.f; xb[e] = (8 * g + 4 + e) < D ? xb[e] : 0.f; }
                h16_split8(xa, xb, xh, xl);
            }
            f4 h1[MT], h2[MT], qq[2];
            // critic and target forwards of the SAME rows side by side (the target's value is only needed for transition t - 1, so it is
            // off the critical path of this step's backward pass); the last step of a chunk has nobody waiting for a bootstrap value
            if (!last) forward(IntC<2>{}, lds, xh, xl, h1, h2, qq);
            else forward(IntC<1>{}, lds, xh, xl, h1, h2, qq);
            const f4 q = qq[0];
            if (!first) {
                // ---- TD error of transition t (dqn/model.py:129,152,160-163) and the backward pass of this row block
                const int a_sel = cur.a_sel;
                const float y = cur.rw + gamma * tq_next * (1.f - cur.dn);
                const float delta = gather_rows_pl<A>(q, lane, a_sel) - y;
                loss_acc += fl * delta * delta;
                nfill_acc += fl;
                const float dqs = 2.f * fl * delta;
                f4 dQ[1];
#pragma unroll
                for (int r = 0; r < 4; ++r) dQ[0][r] = (4 * g + r == a_sel) ? dqs : 0.f;
                db3 += dQ[0];
                // dH2 = W3^T dQ: dQ has one non-zero per row, so it is row a_sel of W3 times that value, masked by relu'
                const int ac = a_sel < A ? (a_sel < 0 ? 0 : a_sel) : A - 1;
                f4 dH2[MT], dH1[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f4 w3 = *reinterpret_cast<const f4*>(lds + K::oW3f + ac * H + 16 * mt + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dH2[mt][r] = h2[mt][r] > 0.f ? w3[r] * dqs : 0.f;
                    db2[mt] += dH2[mt];
                }
                // dH1 = W2^T dH2, masked
                {
                    h8 bh[KG], bl[KG];
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) h16_split8(dH2[2 * kg], dH2[2 * kg + 1], bh[kg], bl[kg]);
                    f4 m[MT], cx[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) { m[mt] = zero4; cx[mt] = zero4; }
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) {  // per product: all tiles (no two consecutive MFMAs on one accumulator)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) m[mt] = MARL_MFMA_H32(T2h[(mt * KG + kg) * 64 + lane], bh[kg], m[mt]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) cx[mt] = MARL_MFMA_H32(T2h[(mt * KG + kg) * 64 + lane], bl[kg], cx[mt]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) cx[mt] = MARL_MFMA_H32(T2l[(mt * KG + kg) * 64 + lane], bh[kg], cx[mt]);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f4 v = m[mt] + cx[mt] * H16_UNSCALE;
#pragma unroll
                        for (int r = 0; r < 4; ++r) dH1[mt][r] = h1[mt][r] > 0.f ? v[r] : 0.f;
                        db1[mt] += dH1[mt];
                    }
                }
                // ---- weight gradients: sums over the 16 rows of outer products; operands transposed through the wave's LDS tiles
                wave_lds_fence();
                tile_write_s<TS, MT>(TH2, h2, g, j);
                tile_write_s<TS, MT>(TH1, h1, g, j);
                tile_write_s<TS, MT>(TG2, dH2, g, j);
                tile_write_s<TS, 1>(TQ, dQ, g, j);
                wave_lds_fence();
                h4 qh, ql;
                h16_split4(tile_read_s<TS>(TQ, 0, g, j), qh, ql);
                h4 g2h[MT], g2l[MT];
                {
                    h4 bh[MT], bl[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        h16_split4(tile_read_s<TS>(TH2, mt, g, j), bh[mt], bl[mt]);    // B operand: h2[row 4 g + e][unit 16 mt + j]
                        h16_split4(tile_read_s<TS>(TG2, mt, g, j), g2h[mt], g2l[mt]);  // A operand: dH2[row 4 g + e][unit 16 mt + i]
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) dW3m[mt] = MARL_MFMA_H16(qh, bh[mt], dW3m[mt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) dW3c[mt] = MARL_MFMA_H16(qh, bl[mt], dW3c[mt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) dW3c[mt] = MARL_MFMA_H16(ql, bh[mt], dW3c[mt]);
                }
                wave_lds_fence();  // the h2 tile has been read: dH1 takes its place
                tile_write_s<TS, MT>(TG1, dH1, g, j);
#pragma unroll
                for (int nu = 0; nu < MT; ++nu) {
                    h4 bh, bl;
                    h16_split4(tile_read_s<TS>(TH1, nu, g, j), bh, bl);  // B operand: h1[row 4 g + e][unit 16 nu + j]
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) dW2m[mt][nu] = MARL_MFMA_H16(g2h[mt], bh, dW2m[mt][nu]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) dW2c[mt][nu] = MARL_MFMA_H16(g2h[mt], bl, dW2c[mt][nu]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) dW2c[mt][nu] = MARL_MFMA_H16(g2l[mt], bh, dW2c[mt][nu]);
                }
                wave_lds_fence();
                {
                    h4 xbh[NT1], xbl[NT1], ah[MT], al[MT];
#pragma unroll
                    for (int nt = 0; nt < NT1; ++nt) {
                        f4 v = {cur.bx[nt][0], cur.bx[nt][1], cur.bx[nt][2], cur.bx[nt][3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (16 * nt + j < D && b0 + 4 * g + e < B) ? v[e] : 0.f;
                        h16_split4(v, xbh[nt], xbl[nt]);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) h16_split4(tile_read_s<TS>(TG1, mt, g, j), ah[mt], al[mt]);
#pragma unroll
                    for (int nt = 0; nt < NT1; ++nt) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) dW1m[mt][nt] = MARL_MFMA_H16(ah[mt], xbh[nt], dW1m[mt][nt]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) dW1c[mt][nt] = MARL_MFMA_H16(ah[mt], xbl[nt], dW1c[mt][nt]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) dW1c[mt][nt] = MARL_MFMA_H16(al[mt], xbh[nt], dW1c[mt][nt]);
                    }
                }
                wave_lds_fence();  // the tiles are rewritten by the next step
            }
            if (!last) {
                // ---- bootstrap value for transition t - 1 from the observation of step t (dqn/model.py:131-145)
                const f4 tq = qq[1];
                const int a_p = double_q ? argmax_rows_pl<A>(q, lane) : argmax_rows_pl<A>(tq, lane);
                tq_next = gather_rows_pl<A>(tq, lane, a_p);
            }
            cur = nxt;
        }
    }

    // ---- fold: every wave lays its gradient out in canonical order in its own LDS region, the workgroup sums the four
    __syncthreads();  // packs and tiles are dead from here on
    float* mine = lds + wave * L::REC;
    for (int i = lane; i < L::REC; i += 64) mine[i] = 0.f;
    wave_lds_fence();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 16 * mt + 4 * g + r;
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) {
                const int d = 16 * nt + j;
                if (d < D) mine[S::oW1 + o * D + d] = dW1m[mt][nt][r] + dW1c[mt][nt][r] * H16_UNSCALE;
            }
#pragma unroll
            for (int nu = 0; nu < MT; ++nu) mine[S::oW2 + o * H + 16 * nu + j] = dW2m[mt][nu][r] + dW2c[mt][nu][r] * H16_UNSCALE;
            const int a = 4 * g + r;
            if (a < A) mine[S::oW3 + a * H + 16 * mt + j] = dW3m[mt][r] + dW3c[mt][r] * H16_UNSCALE;
            const float s1 = sum16(db1[mt][r]), s2 = sum16(db2[mt][r]);
            if (j == 0) { mine[S::ob1 + o] = s1; mine[S::ob2 + o] = s2; }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a = 4 * g + r;
        const float s3 = sum16(db3[r]);
        if (j == 0 && a < A) mine[S::ob3 + a] = s3;
    }
    {
        const float ls = sum16(loss_acc), ns = sum16(nfill_acc);  // every lane group g carries the same row sums: g == 0 reports
        if (lane == 0) { mine[S::NPARAM] = ls; mine[S::NPARAM + 1] = p == 0 ? ns : 0.f; }
    }
    __syncthreads();
    float* rec = partials + ((size_t)p * gridDim.x + blockIdx.x) * L::REC;
    for (int i = tid; i < L::REC; i += 256) rec[i] = (lds[i] + lds[L::REC + i]) + (lds[2 * L::REC + i] + lds[3 * L::REC + i]);
}

// loss / gradient (unnormalised partial records -> dqn_reduce*): packs live behind the records in the workspace like the f32 form
template <class S, bool REPLAY>
int launch_lossgrad_h16(const marlhip_net_shape* s, const float* params, const float* tparams, const marlhip_batch* bt, const ReplaySrc& src,
                        float gamma, int double_q, void* ws, int64_t ws_bytes, hipStream_t st, UpdPlan* plan_out, int* rec_out) {
    using K = H16Pack<S>;
    using L = H16Lds<S>;
    static_assert(L::FITS, "split16 learner: packs + tiles exceed the LDS");
    const int P = s->n_agents, T = bt->max_len, B = bt->batch;
    const AgentMap am = agent_map(s);
    const UpdPlan pl = upd_plan(P, T, B);
    constexpr int PACK_F32 = 2 * S::NFWD + S::NBWD;
    constexpr int PACK = K::TOTAL > PACK_F32 ? K::TOTAL : PACK_F32;  // what marlhip_dqn_workspace_bytes reserves per agent (the larger form)
    const WsLayout wl = ws_layout(P, pl.nwg, L::REC, PACK, T, B);
    MARL_REQUIRE(ws_bytes >= wl.total, "dqn_loss_grad_split16: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)wl.total);
    float* packs = reinterpret_cast<float*>(static_cast<char*>(ws) + wl.pack_off);
    constexpr int NPACK_THREADS = 2 * (2 * (K::nA1 + K::nA2 + K::nA3) + 2 * S::H + 16 + 16 * S::H) + 2 * K::nA2;
    hipLaunchKernelGGL((h16_pack_kernel<S>), dim3((NPACK_THREADS + 255) / 256, P), dim3(256), 0, st, params, tparams, am, packs);
    MARL_CHECK_LAUNCH("h16_pack_kernel");
    const size_t lds_bytes = (size_t)L::total * sizeof(float);
    static LdsAttr attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dqn_lossgrad_h16_kernel<S, REPLAY>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_bytes);
        attr_set.done();
    }
    timing_begin(TIMER_LOSSGRAD, st);
    hipLaunchKernelGGL((dqn_lossgrad_h16_kernel<S, REPLAY>), dim3(pl.nwg, P), dim3(256), lds_bytes, st, (const float*)packs, *bt, src, gamma,
                       double_q, pl.n_chunks, (float*)ws);
    timing_end(TIMER_LOSSGRAD, st);
    MARL_CHECK_LAUNCH("dqn_lossgrad_h16_kernel");
    *plan_out = pl;
    *rec_out = L::REC;
    return 0;
}

}  // namespace marl
