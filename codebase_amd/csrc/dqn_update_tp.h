// Hidden-dimension tensor-parallel learner kernels (hidden = 16 * W * TPW; built for hidden 128 = 4 waves x 2 tiles).
// Included by dqn_update.hip after its helpers (tile_write / tile_read / sum16 / ReplaySrc / MixBufs).
//
// Why a second formulation: at hidden 128 the single-wave kernel of dqn_update.hip would need 256 accumulator
// registers for dW2 alone and 240 KB of LDS weight packs.  Here a workgroup of W waves owns a row block
// together: wave w holds the weights and the gradient accumulators of ITS hidden tiles in REGISTERS (the 512 KB
// register file of a CU is the only on-chip store large enough), activations are exchanged through LDS:
//   pass F  : critic + target forward -> chosen_p[t][b], bootstrap value tqsel_p[t][b]       (no accumulators)
//   mixer   : IDQN (per agent) or VDN (sum) -> dq_p[t][b] = dL/dchosen_p, per-row loss        (tiny kernel)
//   pass B  : critic forward again (layers 1-2 only) + backward with dq; every wave writes its own gradient slice
// oracle/mfma_emul_tp.py is the lane-level statement of this file (checked against torch autograd on CPU).
// Layer 3 is split-K over the waves (each wave multiplies its own h2 tiles, partial Q summed in wave order).
#pragma once

namespace marl {

// relu as ONE v_max_f32 per element (round 6).  fmaxf(v, 0) lowers to a canonicalising v_max(v, v) + v_max(v, 0): two VALU instructions per
// element on top of the v_accvgpr_read that brings an accumulator of the builtin (AGPR-destination) MFMA form into a VGPR - 128 of the 514
// VALU instructions of a tp_fwd step (static count, 15-wide rows) next to 320 MFMAs.  The operand reaches the asm through that compiler-
// inserted, hazard-aware read, so no manual wait states are needed here (unlike relu4_settled, whose inputs are VGPR-form MFMA results).
// Equal to fmaxf(v, 0) for every non-NaN input.
__device__ __forceinline__ f4 relu4_one(f4 v) {
    f4 o;
    asm("v_max_f32 %0, 0, %1" : "=v"(o.x) : "v"(v.x));
    asm("v_max_f32 %0, 0, %1" : "=v"(o.y) : "v"(v.y));
    asm("v_max_f32 %0, 0, %1" : "=v"(o.z) : "v"(v.z));
    asm("v_max_f32 %0, 0, %1" : "=v"(o.w) : "v"(v.w));
    return o;
}

// ---- round 6: the forward pass's hidden-layer chains with VGPR accumulators and the weights named as AGPR operands -------------------------
// hipcc gives every MFMA of a > 256-register kernel an AGPR destination, so each pre-activation a relu touches costs a v_accvgpr_read
// (+ the two-instruction fmaxf): 192 of tp_fwd's 514 VALU instructions per step.  Written as inline asm the register classes are ours:
// accumulators in VGPRs (the relu reads them directly: one v_max each), the wave's resident weights as "a" operands (the compiler keeps
// them in AGPRs for the whole kernel), activations "v".  The FIRST k-step of a chain takes its C operand from the bias registers (D != C is
// legal), so no accumulator is ever initialised by VALU moves.  Order per accumulator: k ascending, as in the builtin form - the same bits.
// Hazards the compiler does not see inside asm: `s_nop 1` in front of a group (an operand written just before), tp_settle8 (12 states)
// between the last MFMA of the chains and the first VALU reader.
// MEASURED (round 6, scripts/gpu_runs/r6F.sh, B = 4096, 15-wide rows): the static VALU count of a tp_fwd step goes 514 -> 383 (no
// v_accvgpr_read left in the loop, 64 v_max instead of 128 + 64 reads, no spills: 203 VGPRs + 106 AGPRs) and the kernel goes 180.7 -> 187.4 us
// on a box whose untouched tp_bwd ran its usual 175 us.  The pass is not VALU-issue-bound (WAIT_INST_ANY 0.69: four waves, one per SIMD,
// meeting at two barriers per step around LDS exchanges) - the H64 kernel's price list (7 cycles per VALU next to an MFMA) does not carry
// over to a kernel that waits at barriers anyway, and the volatile asm groups take scheduling freedom from the compiler (operand reads can
// no longer sit between the MFMAs of a group).  OFF by default; kept as the record of the experiment (bitwise the builtin form's results:
// the goldens and the at-size tests pass with it on).
#ifndef MARL_TP_VGPR
#define MARL_TP_VGPR 0
#endif
// MARL_TP_PROF=1 (profiling builds only): s_memtime at the phase boundaries of a pass-F step, averaged per wave and printed by one workgroup at
// the end of the launch - rows + layer 1 | first barrier | layer 2 | relu + stores + layer 3 | second barrier | epilogue.  What it showed on the
// round-5 form (scripts/gpu_runs/r6M.sh, 15-wide rows, cycles per step): 2.7 k | 0.1 k (waves 2, 3: 2.0 k) | 7.35 k | 2.75 k | 0.1 k | 2.3 k on the two
// epilogue waves (0.3 k on the others): 256 of the step's 320 MFMAs sit in the 7.35 k, and the one-wave epilogue - a latency chain of LDS
// reads, sums, permlane argmax and scattered stores - was 2.3 k of exposed time every step, which the other waves spent at the next barrier.
#ifndef MARL_TP_PROF
#define MARL_TP_PROF 0
#endif
// MARL_TP_DEFER_EPI=1 (experiment, round 6; OFF): the epilogue of step t runs in step t - 1's iteration, behind its first barrier and in the
// same scheduling region as its layer-2 MFMAs (the Q partials sit in the LDS set of step t's parity, which nothing writes before step t - 2),
// so that the latency chain hides under matrix work.  Same sums, same order, same bits (goldens and the config-3 at-size test pass with it on).
// MEASURED (scripts/gpu_runs/r6N.sh, both variants of one tree on one box): loss/grad group 358.3 -> 367.4 us.  The phase counters say why: the
// layer-2 region absorbs the epilogue for 1.35 k instead of 2.3 k on its two waves (the other two now wait 1.2 k at the SECOND barrier), but
// the rows + layer-1 region grows 2.7 k -> 3.9 k on all four - the epilogue's scattered stores sit in front of the next step's row loads in
// the in-order memory counter.  Splitting the epilogue over two waves per row block (r6K.sh) changed nothing either: it is a latency
// chain, not work.
#ifndef MARL_TP_DEFER_EPI
#define MARL_TP_DEFER_EPI 0
#endif
struct TpPend {  // what the epilogue of a step needs from its rows
    int a_sel;
    float rw, dn, fl, mk[4];
};
// workgroups of pass B per compute unit.  MEASURED (round 6, scripts/gpu_runs/r6G.sh): 2 (launch_bounds(256, 2): 256 registers, 31 spilled,
// with -DMARL_TP_NB1S_D=0 = one row block per step so that two workgroups' LDS fit) - a second wave per SIMD to hide the barrier / exchange
// latency - runs tp_bwd at 180.7 us against 175 (IDQN 128-128), VDN 15x15-4p 3.96 -> 3.57 M, and the actor-critic FULL form spills badly
// (config 4: 62 -> 33 M).  The matrix pipe is shared by the SIMD's waves and the pass holds its weights in registers: halving the register
// budget costs more than the overlap returns.  1 it stays.
#ifndef MARL_TP_BWD_OCC
#define MARL_TP_BWD_OCC 1
#endif
#define MARL_TP_M8(D0, D1, D2, D3, D4, D5, D6, D7, C0, C1, C2, C3, C4, C5, C6, C7)                      \
    "s_nop 1\n\t"                                                                                      \
    "v_mfma_f32_16x16x4_f32 " D0 ", %[wc0], %[bc0], " C0 "\n\t"                                          \
    "v_mfma_f32_16x16x4_f32 " D4 ", %[wt0], %[bt0], " C4 "\n\t"                                          \
    "v_mfma_f32_16x16x4_f32 " D1 ", %[wc1], %[bc0], " C1 "\n\t"                                          \
    "v_mfma_f32_16x16x4_f32 " D5 ", %[wt1], %[bt0], " C5 "\n\t"                                          \
    "v_mfma_f32_16x16x4_f32 " D2 ", %[wc0], %[bc1], " C2 "\n\t"                                          \
    "v_mfma_f32_16x16x4_f32 " D6 ", %[wt0], %[bt1], " C6 "\n\t"                                          \
    "v_mfma_f32_16x16x4_f32 " D3 ", %[wc1], %[bc1], " C3 "\n\t"                                          \
    "v_mfma_f32_16x16x4_f32 " D7 ", %[wt1], %[bt1], " C7 "\n\t"
// ac[nb][u] / at[nb][u] (critic / target) += w{c,t}[u] x b{c,t}[nb]: 8 MFMAs on 8 chains, issue order (nb, u, net) as the builtin loop
__device__ __forceinline__ void tp_mfma8(f4 (&ac)[2][2], f4 (&at)[2][2], float wc0, float wc1, float wt0, float wt1, float bc0, float bc1, float bt0,
                                         float bt1) {
    asm volatile(MARL_TP_M8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7", "%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7")
                 : "+v"(ac[0][0]), "+v"(ac[0][1]), "+v"(ac[1][0]), "+v"(ac[1][1]), "+v"(at[0][0]), "+v"(at[0][1]), "+v"(at[1][0]), "+v"(at[1][1])
                 : [wc0] "a"(wc0), [wc1] "a"(wc1), [wt0] "a"(wt0), [wt1] "a"(wt1), [bc0] "v"(bc0), [bc1] "v"(bc1), [bt0] "v"(bt0), [bt1] "v"(bt1));
}
// the first k-step: D = A B + bias (cb[u] / tb[u], the wave's resident bias rows in accumulator layout)
__device__ __forceinline__ void tp_mfma8_first(f4 (&ac)[2][2], f4 (&at)[2][2], const f4& cb0, const f4& cb1, const f4& tb0, const f4& tb1, float wc0,
                                               float wc1, float wt0, float wt1, float bc0, float bc1, float bt0, float bt1) {
    asm volatile(MARL_TP_M8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7", "%8", "%9", "%8", "%9", "%10", "%11", "%10", "%11")
                 : "=&v"(ac[0][0]), "=&v"(ac[0][1]), "=&v"(ac[1][0]), "=&v"(ac[1][1]), "=&v"(at[0][0]), "=&v"(at[0][1]), "=&v"(at[1][0]), "=&v"(at[1][1])
                 : "v"(cb0), "v"(cb1), "v"(tb0), "v"(tb1), [wc0] "a"(wc0), [wc1] "a"(wc1), [wt0] "a"(wt0), [wt1] "a"(wt1), [bc0] "v"(bc0), [bc1] "v"(bc1),
                   [bt0] "v"(bt0), [bt1] "v"(bt1));
}
// pass B's two chains whose results the VALU masks next (dH2 = W3^T dQ, dH1 = W2^T dH2): d[nb][u] += w[u] x b[nb], 4 MFMAs on 4 chains in the
// builtin loop's order (nb, u); the first k-step starts from the inline constant 0 instead of VALU-zeroed accumulators
__device__ __forceinline__ void tp_mfma4(f4 (&d)[2][2], float w0, float w1, float b0, float b1) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %4, %6, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %1, %5, %6, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %2, %4, %7, %2\n\t"
                 "v_mfma_f32_16x16x4_f32 %3, %5, %7, %3\n\t"
                 : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1])
                 : "a"(w0), "a"(w1), "v"(b0), "v"(b1));
}
__device__ __forceinline__ void tp_mfma4_zero(f4 (&d)[2][2], float w0, float w1, float b0, float b1) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %4, %6, 0\n\t"
                 "v_mfma_f32_16x16x4_f32 %1, %5, %6, 0\n\t"
                 "v_mfma_f32_16x16x4_f32 %2, %4, %7, 0\n\t"
                 "v_mfma_f32_16x16x4_f32 %3, %5, %7, 0\n\t"
                 : "=&v"(d[0][0]), "=&v"(d[0][1]), "=&v"(d[1][0]), "=&v"(d[1][1])
                 : "a"(w0), "a"(w1), "v"(b0), "v"(b1));
}
__device__ __forceinline__ void tp_settle4(f4 (&d)[2][2]) { asm volatile("s_nop 11" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1])); }
__device__ __forceinline__ void tp_settle8(f4 (&ac)[2][2], f4 (&at)[2][2]) {
    asm volatile("s_nop 11" : "+v"(ac[0][0]), "+v"(ac[0][1]), "+v"(ac[1][0]), "+v"(ac[1][1]), "+v"(at[0][0]), "+v"(at[0][1]), "+v"(at[1][0]), "+v"(at[1][1]));
}

struct TpMix {
    float* chosen;  // [P][T][B]
    float* tqsel;   // [P][T][B]
    float* rew;     // [P][T][B]
    float* dn;      // [T][B]
    float* fl;      // [T][B]
    float* dq;      // [P][T][B]
    float* lrow;    // [T][B]
    const float* dout;  // FULL pass B: [P][T][B][A] external gradient w.r.t. every network output
};

// register image of the weights of one owned hidden tile
template <class S, int NT>
struct TpTileW {
    float a1[S::KS1];  // W1[16tau+i][4ks+g]
    f4 a2[NT];         // W2[16tau+i][16kap+4g+r]
    f4 b1s, b2s;       // b1[16tau+4g+r], b2[...]
};

template <class S, int NT>
__device__ __forceinline__ void tp_load_fwd(const float* __restrict__ w, int tau, int lane, TpTileW<S, NT>& t) {
    const int g = lane >> 4, i = lane & 15;
#pragma unroll
    for (int ks = 0; ks < S::KS1; ++ks) {
        const int k = 4 * ks + g;
        t.a1[ks] = k < S::D ? w[S::oW1 + (16 * tau + i) * S::D + k] : 0.f;
    }
#pragma unroll
    for (int kap = 0; kap < NT; ++kap)
#pragma unroll
        for (int r = 0; r < 4; ++r) t.a2[kap][r] = w[S::oW2 + (16 * tau + i) * S::H + 16 * kap + 4 * g + r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        t.b1s[r] = w[S::ob1 + 16 * tau + 4 * g + r];
        t.b2s[r] = w[S::ob2 + 16 * tau + 4 * g + r];
    }
}

// rows of one 16-episode block at one time step
template <class S, bool REPLAY>
struct TpRows {
    static constexpr int NT1 = S::DP / 16;
    float x[S::KS1];
    float bx[NT1][4];
    int a_sel;
    float rw, dn, fl, dq, lr;
    float dqv[4];  // FULL: dL/d(output 4g+r)
    float mk[4];   // pass F: batch.action_mask of outputs 4g+r (see marlhip_batch)
};

template <class S, bool REPLAY>
struct TpSrc {  // everything load_rows needs, per block
    const float* mask_p;  // batch.action_mask of this agent ([T+1][B][A]) or nullptr
    const float* obs_p;
    const int64_t* act_p;
    const float* rew_p;
    const float* dones;
    const float* filled;
    ReplaySrc rs;
    int P, p, T, B;
    size_t obs_rs;  // row stride of obs_p (D in the dqn/train.py Batch)
};

// NOX: the rows' MFMA-A-side copy x is not needed (pass B with BOTH hidden layers read back: no layer-1 recompute)
template <class S, bool REPLAY, bool BWD, bool FULL = false, bool NOX = false>
__device__ __forceinline__ void tp_load_rows(const TpSrc<S, REPLAY>& s, const TpMix& mix, int t, int b0, int g, int j, int ej,
                                             const int (&eg)[4], TpRows<S, REPLAY>& R) {
    constexpr int D = S::D, NT1 = S::DP / 16;
    const int T = s.T, B = s.B, p = s.p;
    const int bj = (b0 + j) < B ? b0 + j : B - 1;
    const int tt = t < T ? t : T - 1;
    if (REPLAY) {
        const float* xrow = s.rs.rb.obs + (((size_t)ej * s.P + p) * (T + 1) + t) * D;
        if (!NOX) {
#pragma unroll
            for (int ks = 0; ks < S::KS1; ++ks) {
                const int d = 4 * ks + g;
                R.x[ks] = xrow[d < D ? d : D - 1];
            }
        }
        if (BWD) {
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int d = 16 * nt + j;
                    R.bx[nt][ks] = s.rs.rb.obs[(((size_t)eg[ks] * s.P + p) * (T + 1) + t) * D + (d < D ? d : D - 1)];
                }
        }
        R.a_sel = (int)s.rs.rb.act[((size_t)ej * s.P + p) * T + tt];
        R.rw = s.rs.rb.rew[((size_t)ej * s.P + p) * T + tt];
        R.dn = s.rs.rb.done[(size_t)ej * (T + 1) + tt + 1] ? 1.f : 0.f;
        R.fl = s.rs.rb.filled[(size_t)ej * T + tt] ? 1.f : 0.f;
    } else {
        const float* xrow = s.obs_p + ((size_t)t * B + bj) * s.obs_rs;
        if (!NOX) {
#pragma unroll
            for (int ks = 0; ks < S::KS1; ++ks) {
                const int d = 4 * ks + g;
                R.x[ks] = xrow[d < D ? d : D - 1];
            }
        }
        if (BWD) {
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int row = b0 + 4 * g + ks, d = 16 * nt + j;
                    R.bx[nt][ks] = s.obs_p[((size_t)t * B + (row < B ? row : B - 1)) * s.obs_rs + (d < D ? d : D - 1)];
                }
        }
        if (!FULL) {
            R.a_sel = (int)s.act_p[(size_t)tt * B + bj];
            R.rw = s.rew_p[(size_t)tt * B + bj];
            R.dn = s.dones[(size_t)(tt + 1) * B + bj];
        }
        R.fl = s.filled[(size_t)tt * B + bj];
    }
    if (!BWD && !REPLAY && s.mask_p != nullptr) {
        const float* mrow = s.mask_p + ((size_t)t * B + bj) * S::A;
#pragma unroll
        for (int r = 0; r < 4; ++r) R.mk[r] = mrow[4 * g + r < S::A ? 4 * g + r : S::A - 1];
    }
    if (BWD) {
        if (FULL) {
            const float* drow = mix.dout + (((size_t)p * T + tt) * B + bj) * S::A;
#pragma unroll
            for (int r = 0; r < 4; ++r) R.dqv[r] = drow[4 * g + r < S::A ? 4 * g + r : S::A - 1];
        } else {
            R.dq = mix.dq[((size_t)p * T + tt) * B + bj];
        }
        R.lr = mix.lrow[(size_t)tt * B + bj];
    }
}

template <class S, bool REPLAY, bool BWD, bool NOX = false>
__device__ __forceinline__ void tp_mask_rows(TpRows<S, REPLAY>& R, int b0, int B, int g, int j) {
    constexpr int D = S::D, NT1 = S::DP / 16;
    const bool rowok = (b0 + j) < B;
    if (!NOX) {
#pragma unroll
        for (int ks = 0; ks < S::KS1; ++ks) R.x[ks] = (4 * ks + g < D && rowok) ? R.x[ks] : 0.f;
    }
    if (BWD) {
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) R.bx[nt][ks] = (b0 + 4 * g + ks < B && 16 * nt + j < D) ? R.bx[nt][ks] : 0.f;
        R.dq = rowok ? R.dq : 0.f;
        R.lr = rowok ? R.lr : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) R.dqv[r] = (rowok && 4 * g + r < S::A) ? R.dqv[r] : 0.f;
    }
    R.fl = rowok ? R.fl : 0.f;
}

template <bool REPLAY>
__device__ __forceinline__ void tp_episode_ids(const ReplaySrc& rs, int b0, int B, int g, int j, int& ej, int (&eg)[4]) {
    ej = 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) eg[ks] = 0;
    if (REPLAY) {
        const int bj = (b0 + j) < B ? b0 + j : B - 1;
        ej = rs.idx ? rs.idx[bj] : replay_draw(rs, bj);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int row = b0 + 4 * g + ks, rc = row < B ? row : B - 1;
            eg[ks] = rs.idx ? rs.idx[rc] : replay_draw(rs, rc);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// pass F: forward of critic and target, workgroup = W waves, NB row blocks per step
// LDS (floats): Hc[NB][NT][256] | Ht[NB][NT][256] | Qp[NB][W][256] | Tp[NB][W][256]
// ---------------------------------------------------------------------------------------------------------
// h2_out (may be null): the critic's second hidden layer of every TRANSITION row (t < T), relu applied, in C layout -
// h2_out[(((p T + t) tp_h2_blocks(B) + b0 / 16) NT + tau) 64 + lane] - so that pass B reads it back instead of recomputing layer 2 (the
// largest GEMM of the step: 256 of pass B's 896 MFMAs per row block at hidden 128).  H f32 per row-step, written once and read once.
// Row blocks are counted in PAIRS (a pass walks 1 or 2 blocks per step): the padding block of an odd batch is written too (its rows
// are zero inputs -> finite values), so pass B never meets uninitialised memory there (0 * NaN would poison dW3).
__host__ __device__ inline int tp_h2_blocks(int B) { return ((B + 31) / 32) * 2; }
template <class S, int W, int TPW, bool REPLAY, int NB>
__global__ __launch_bounds__(64 * W, 1) void tp_fwd_kernel(const float* __restrict__ params, const float* __restrict__ tparams,
                                                           AgentMap am, marlhip_batch bt, ReplaySrc rs, TpMix mix, int double_q, int n_chunks,
                                                           f4* __restrict__ h2_out) {
    constexpr int NT = W * TPW, A = S::A;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // TWO sets of the exchange regions, alternating with the time step: the next step writes the other set, so the barrier that used to
    // protect this step's regions from the next step's writes is gone (two barriers per step instead of three; a set is rewritten two
    // steps later, behind both barriers of the step in between)
    constexpr int SETF = (2 * NB * NT + 2 * NB * W) * 64;  // f4 per set
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int p = blockIdx.y, P = gridDim.y;
    const int T = bt.max_len, B = bt.batch;

    TpTileW<S, NT> cw[TPW], tw[TPW];
    f4 ca3[TPW], ta3[TPW], cb3, tb3;
    {
        const float* wc = params + (size_t)am.net[p] * S::NPARAM;
        const float* wt = tparams + (size_t)am.net[p] * S::NPARAM;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int tau = wave * TPW + u;
            tp_load_fwd<S, NT>(wc, tau, lane, cw[u]);
            tp_load_fwd<S, NT>(wt, tau, lane, tw[u]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // split-K layer 3: A operand W3[i][16tau+4g+r]
                ca3[u][r] = j < A ? wc[S::oW3 + j * S::H + 16 * tau + 4 * g + r] : 0.f;
                ta3[u][r] = j < A ? wt[S::oW3 + j * S::H + 16 * tau + 4 * g + r] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = 4 * g + r;
            cb3[r] = (wave == 0 && a < A) ? wc[S::ob3 + a] : 0.f;
            tb3[r] = (wave == 0 && a < A) ? wt[S::ob3 + a] : 0.f;
        }
    }
    TpSrc<S, REPLAY> src;
    src.obs_p = REPLAY ? nullptr : bt.obss + (size_t)p * (bt.obs_agent_stride > 0 ? (size_t)bt.obs_agent_stride : (bt.obs_agent_stride < 0 ? 0 : (size_t)(T + 1) * B * S::D));
    src.mask_p = (REPLAY || bt.action_mask == nullptr) ? nullptr : bt.action_mask + (size_t)p * (T + 1) * B * S::A;
    src.obs_rs = bt.obs_row_stride ? (size_t)bt.obs_row_stride : (size_t)S::D;
    src.act_p = REPLAY ? nullptr : bt.actions + (size_t)p * T * B;
    src.rew_p = REPLAY ? nullptr : bt.rewards + (size_t)p * T * B;
    src.dones = bt.dones; src.filled = bt.filled; src.rs = rs; src.P = P; src.p = p; src.T = T; src.B = B;

    const int nsets = (B + 16 * NB - 1) / (16 * NB);
    const int ntasks = nsets * n_chunks;
#if MARL_TP_PROF
    unsigned long long tp_sp[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const int set = task / n_chunks, c = task - set * n_chunks;
        const int t0 = (c * T) / n_chunks, t1 = ((c + 1) * T) / n_chunks;
        if (t1 <= t0) continue;  // uniform over the workgroup
        int ej[NB], eg[NB][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) tp_episode_ids<REPLAY>(rs, (set * NB + nb) * 16, B, g, j, ej[nb], eg[nb]);
        if (REPLAY && rs.idx_out != nullptr && p == 0 && c == 0 && wave == 0 && g == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                if ((set * NB + nb) * 16 + j < B) rs.idx_out[(set * NB + nb) * 16 + j] = ej[nb];
        }
        TpRows<S, REPLAY> cur[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) tp_load_rows<S, REPLAY, false>(src, mix, t1, (set * NB + nb) * 16, g, j, ej[nb], eg[nb], cur[nb]);
        // the epilogue of step `tt` on the Q partials of its LDS set: the block's wave finishes Q (partials summed in wave order), publishes the
        // mixer inputs of transition tt and the bootstrap value of transition tt - 1
        TpPend pend[NB];
        int pend_t = -1;
        auto epilogue = [&](int tt, const TpPend (&pr)[NB]) {
            const f4* Qe = reinterpret_cast<const f4*>(lds) + (tt & 1) * SETF + 2 * NB * NT * 64;
            const f4* Te = Qe + NB * W * 64;
            const bool need_tt = tt > t0;  // the target value of this step feeds transition tt - 1
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (wave != nb % W) continue;
                const int b0 = (set * NB + nb) * 16;
                const bool rowok = (b0 + j) < B;
                const int bj = rowok ? b0 + j : B - 1;
                f4 q = Qe[(nb * W + 0) * 64 + lane], tq = Te[(nb * W + 0) * 64 + lane];
#pragma unroll
                for (int w2 = 1; w2 < W; ++w2) {
                    q += Qe[(nb * W + w2) * 64 + lane];
                    tq += Te[(nb * W + w2) * 64 + lane];
                }
                if (tt < t1) {
                    const float ch = gather_rows_pl<A>(q, lane, pr[nb].a_sel);
                    if (g == 0 && rowok) {
                        mix.chosen[((size_t)p * T + tt) * B + bj] = ch;
                        mix.rew[((size_t)p * T + tt) * B + bj] = pr[nb].rw;
                        if (p == 0) {
                            mix.dn[(size_t)tt * B + bj] = pr[nb].dn;
                            mix.fl[(size_t)tt * B + bj] = pr[nb].fl;
                        }
                    }
                }
                if (need_tt) {
                    if (!REPLAY && src.mask_p != nullptr) {  // dqn/model.py:136-142
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (pr[nb].mk[r] == 0.f) { q[r] = -1e8f; tq[r] = -1e8f; }
                        }
                    }
                    const int a_p = double_q ? argmax_rows_pl<A>(q, lane) : argmax_rows_pl<A>(tq, lane);
                    const float tv = gather_rows_pl<A>(tq, lane, a_p);
                    if (g == 0 && rowok) mix.tqsel[((size_t)p * T + (tt - 1)) * B + bj] = tv;
                }
            }
        };
#if MARL_TP_PROF
#define MARL_TPP(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tp_sp[k] += now_ - tp_c; tp_c = now_; }
#else
#define MARL_TPP(k)
#endif
        for (int t = t1; t >= t0; --t) {
            TpRows<S, REPLAY> nxt[NB];
#if MARL_TP_PROF
            unsigned long long tp_c = __builtin_readcyclecounter();
            tp_sp[6] += 1;
#endif
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                tp_load_rows<S, REPLAY, false>(src, mix, t > t0 ? t - 1 : t0, (set * NB + nb) * 16, g, j, ej[nb], eg[nb], nxt[nb]);
                tp_mask_rows<S, REPLAY, false>(cur[nb], (set * NB + nb) * 16, B, g, j);
            }
            f4* Hc = reinterpret_cast<f4*>(lds) + (t & 1) * SETF;
            f4* Ht = Hc + NB * NT * 64;
            f4* Qp = Ht + NB * NT * 64;
            f4* Tp = Qp + NB * W * 64;
            const bool st_rec = h2_out != nullptr && t < t1;  // a transition row of this chunk leaves its hidden layers for pass B (t1 itself only bootstraps here)
            // ---- layer 1 of my tiles, dumped for everybody (the 2 NB TPW chains advance together)
            {
                f4 a1c[NB][TPW], a1t[NB][TPW];
                constexpr bool VG = MARL_TP_VGPR && NB == 2 && TPW == 2;
                if constexpr (VG) {
                    tp_mfma8_first(a1c, a1t, cw[0].b1s, cw[1].b1s, tw[0].b1s, tw[1].b1s, cw[0].a1[0], cw[1].a1[0], tw[0].a1[0], tw[1].a1[0], cur[0].x[0],
                                   cur[1].x[0], cur[0].x[0], cur[1].x[0]);
#pragma unroll
                    for (int ks = 1; ks < S::KS1; ++ks)
                        tp_mfma8(a1c, a1t, cw[0].a1[ks], cw[1].a1[ks], tw[0].a1[ks], tw[1].a1[ks], cur[0].x[ks], cur[1].x[ks], cur[0].x[ks], cur[1].x[ks]);
                    tp_settle8(a1c, a1t);
                } else {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int u = 0; u < TPW; ++u) { a1c[nb][u] = cw[u].b1s; a1t[nb][u] = tw[u].b1s; }
#pragma unroll
                    for (int ks = 0; ks < S::KS1; ++ks)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int u = 0; u < TPW; ++u) {
                                a1c[nb][u] = MARL_MFMA(cw[u].a1[ks], cur[nb].x[ks], a1c[nb][u]);
                                a1t[nb][u] = MARL_MFMA(tw[u].a1[ks], cur[nb].x[ks], a1t[nb][u]);
                            }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int u = 0; u < TPW; ++u) {
                        const f4 h1c = VG ? relu4_settled(a1c[nb][u]) : relu4_one(a1c[nb][u]);
                        Hc[(nb * NT + wave * TPW + u) * 64 + lane] = h1c;
                        Ht[(nb * NT + wave * TPW + u) * 64 + lane] = VG ? relu4_settled(a1t[nb][u]) : relu4_one(a1t[nb][u]);
                        if (st_rec)  // the critic's first hidden layer too (behind the h2 record, same layout): pass B recomputes nothing
                            h2_out[(size_t)P * T * tp_h2_blocks(B) * NT * 64 + ((((size_t)p * T + t) * tp_h2_blocks(B) + (set * NB + nb)) * NT + wave * TPW + u) * 64 + lane] = h1c;
                    }
            }
            MARL_TPP(0)
            __syncthreads();
            MARL_TPP(1)
            if (MARL_TP_DEFER_EPI && pend_t >= 0) epilogue(pend_t, pend);  // (no scheduling barrier towards the MFMAs below)
            // ---- layer 2 of my tiles + my split-K share of layer 3.  The (row block, owned tile, network) chains - 2 NB TPW of them - advance
            // together, k ascending in each (bitwise the order of the one-chain-at-a-time form): no MFMA waits for its own predecessor
            {
                f4 acc[NB][TPW], acct[NB][TPW];
                constexpr bool VG = MARL_TP_VGPR && NB == 2 && TPW == 2;
                if constexpr (!VG) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int u = 0; u < TPW; ++u) { acc[nb][u] = cw[u].b2s; acct[nb][u] = tw[u].b2s; }
                }
#pragma unroll
                for (int kap = 0; kap < NT; ++kap) {
                    f4 hk[NB], gk[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) { hk[nb] = Hc[(nb * NT + kap) * 64 + lane]; gk[nb] = Ht[(nb * NT + kap) * 64 + lane]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (VG) {
                            if (kap == 0 && r == 0)
                                tp_mfma8_first(acc, acct, cw[0].b2s, cw[1].b2s, tw[0].b2s, tw[1].b2s, cw[0].a2[0][0], cw[1].a2[0][0], tw[0].a2[0][0],
                                               tw[1].a2[0][0], hk[0][0], hk[1][0], gk[0][0], gk[1][0]);
                            else
                                tp_mfma8(acc, acct, cw[0].a2[kap][r], cw[1].a2[kap][r], tw[0].a2[kap][r], tw[1].a2[kap][r], hk[0][r], hk[1][r], gk[0][r],
                                         gk[1][r]);
                        } else {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                for (int u = 0; u < TPW; ++u) {
                                    acc[nb][u] = MARL_MFMA(cw[u].a2[kap][r], hk[nb][r], acc[nb][u]);
                                    acct[nb][u] = MARL_MFMA(tw[u].a2[kap][r], gk[nb][r], acct[nb][u]);
                                }
                        }
                    }
                }
                if constexpr (VG) tp_settle8(acc, acct);
                MARL_TPP(2)
                f4 q[NB], tq[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    q[nb] = cb3; tq[nb] = tb3;
#pragma unroll
                    for (int u = 0; u < TPW; ++u) {
                        acc[nb][u] = VG ? relu4_settled(acc[nb][u]) : relu4_one(acc[nb][u]);
                        acct[nb][u] = VG ? relu4_settled(acct[nb][u]) : relu4_one(acct[nb][u]);
                        if (st_rec)
                            h2_out[((((size_t)p * T + t) * tp_h2_blocks(B) + (set * NB + nb)) * NT + wave * TPW + u) * 64 + lane] = acc[nb][u];
                    }
                }
#pragma unroll
                for (int u = 0; u < TPW; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            q[nb] = MARL_MFMA(ca3[u][r], acc[nb][u][r], q[nb]);
                            tq[nb] = MARL_MFMA(ta3[u][r], acct[nb][u][r], tq[nb]);
                        }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    Qp[(nb * W + wave) * 64 + lane] = q[nb];
                    Tp[(nb * W + wave) * 64 + lane] = tq[nb];
                }
            }
            MARL_TPP(3)
            __syncthreads();
            MARL_TPP(4)
            // ---- the step's epilogue: deferred into the next iteration (MARL_TP_DEFER_EPI), or here
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                pend[nb].a_sel = cur[nb].a_sel; pend[nb].rw = cur[nb].rw; pend[nb].dn = cur[nb].dn; pend[nb].fl = cur[nb].fl;
#pragma unroll
                for (int r = 0; r < 4; ++r) pend[nb].mk[r] = cur[nb].mk[r];
            }
            if (MARL_TP_DEFER_EPI) {
                pend_t = t;
            } else {
                epilogue(t, pend);
            }
            // (no barrier: the next step works on the other LDS set)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) cur[nb] = nxt[nb];
            MARL_TPP(5)
        }
        if (MARL_TP_DEFER_EPI && pend_t >= 0) epilogue(pend_t, pend);  // the chunk's last step
        __syncthreads();  // the next task may start on either set
    }
#if MARL_TP_PROF
    if (blockIdx.x == 7 && blockIdx.y == 0 && lane == 0 && tp_sp[6] > 0)
        printf("TPPROF wave %d steps %llu: rows+L1 %llu | barrier1 %llu | (epilogue+)L2 %llu | relu+store+L3 %llu | barrier2 %llu | tail %llu (cycles per step)\n", wave,
               tp_sp[6], tp_sp[0] / tp_sp[6], tp_sp[1] / tp_sp[6], tp_sp[2] / tp_sp[6], tp_sp[3] / tp_sp[6], tp_sp[4] / tp_sp[6], tp_sp[5] / tp_sp[6]);
#endif
#undef MARL_TPP
}

// dynamic LDS of tp_bwd_kernel in floats: Hc | [HcT | G2] | per-wave tiles | (STORED) a second [HcT | G2] set
template <class S, int W, int TPW, int NB, bool STORED>
constexpr int tp_bwd_lds_floats() {
    return NB * W * TPW * 256 + (STORED ? 2 : 1) * (NB * S::H * 16 + NB * W * TPW * 256) + W * 256 * NB * (1 + 3 * TPW);
}

// mixers: dq_p[t][b] = dL/dchosen_p (unnormalised), lrow[t][b] = per-row loss (dqn/model.py:152,160-163 / 254-269)
static __global__ __launch_bounds__(256) void tp_mix_kernel(TpMix mix, int P, int T, int B, float gamma, int vdn, const float* __restrict__ ret) {
    const int n = T * B;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float fl = mix.fl[i], nd = 1.f - mix.dn[i];
        if (vdn) {
            float ch = 0.f, tq = 0.f;
            for (int p = 0; p < P; ++p) {
                ch += mix.chosen[(size_t)p * n + i];
                tq += mix.tqsel[(size_t)p * n + i];
            }
            const float delta = ch - (ret != nullptr ? ret[i] : mix.rew[i] + gamma * tq * nd);
            for (int p = 0; p < P; ++p) mix.dq[(size_t)p * n + i] = 2.f * fl * delta;
            mix.lrow[i] = fl * delta * delta;
        } else {
            float l = 0.f;
            for (int p = 0; p < P; ++p) {
                const float delta = mix.chosen[(size_t)p * n + i] - (mix.rew[(size_t)p * n + i] + gamma * mix.tqsel[(size_t)p * n + i] * nd);
                mix.dq[(size_t)p * n + i] = 2.f * fl * delta;
                l += delta * delta;  // sum over agents (mse_loss(...).sum(dim=0), model.py:160-162)
            }
            mix.lrow[i] = fl * l;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// pass B: critic layers 1-2 forward + backward with the external dq; gradient slices stay with their wave
// LDS (floats): Hc[NB][NT][256] | HcT[NB][H][16] | G2[NB][NT][256] | per wave: PQ[256] + (P2,PH2,P1)[TPW][256]
// ---------------------------------------------------------------------------------------------------------
// STORED: layer 2 is not recomputed - the wave reads ITS h2 tiles of the row block back from pass F's h2_out (one step ahead, next
// to the rows), so the C-layout dump of h1 and the barrier behind it go away too: layer 1 -> [dH2, dW3 from the stored h2] ->
// barrier -> [dH1, dW2, dW1] -> barrier.  Same arithmetic per element as the recomputing form (pass F ran the identical MFMA chain).
// STORED1 (with STORED; round 4, the actor-critic step): layer 1 is not recomputed either - the wave reads ITS h1 tiles of the row block from
// the forward-rows pass (h1_in, the layout of h2_in): no layer-1 weights, no A-side copy of the rows in registers, 2 KS1 fewer MFMAs per block.
template <class S, int W, int TPW, bool REPLAY, int NB, bool FULL = false, bool STORED = false, bool STORED1 = false>
__global__ __launch_bounds__(64 * W, MARL_TP_BWD_OCC) void tp_bwd_kernel(const float* __restrict__ params, AgentMap am, marlhip_batch bt, ReplaySrc rs, TpMix mix,
                                                           int n_chunks, float* __restrict__ partials, const f4* __restrict__ h2_in = nullptr,
                                                           const f4* __restrict__ h1_in = nullptr) {
    static_assert(!STORED1 || STORED, "the stored first layer comes with the stored second");
    constexpr int NT = W * TPW, A = S::A, D = S::D, H = S::H, NT1 = S::DP / 16;
    constexpr int PRIV = 256 * NB * (1 + 3 * TPW);  // per wave: PQ[NB] | PH2[NB][TPW] | P2[NB][TPW] | P1[NB][TPW] tiles of 256 floats
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // STORED: [HcT | G2] exist TWICE (behind the per-wave regions) and alternate with the time step, so the barrier that protected them from
    // the next step's writes is gone: one barrier per step.  The recomputing form keeps the single set and its barriers.
    f4* Hc = reinterpret_cast<f4*>(lds);
    float* HcT0 = lds + NB * NT * 256;
    float* priv = HcT0 + NB * H * 16 + NB * NT * 256;
    float* set1 = priv + W * 256 * NB * (1 + 3 * TPW);  // second [HcT | G2] set (STORED only; tp_bwd_lds_floats)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    float* PQ = priv + wave * PRIV;
    float* P2 = PQ + 256 * NB;
    float* PH2 = P2 + 256 * NB * TPW;
    float* P1 = PH2 + 256 * NB * TPW;
    const int p = blockIdx.y, P = gridDim.y;
    const int T = bt.max_len, B = bt.batch;

    TpTileW<S, NT> cw[TPW];
    f4 t3[TPW], t2[TPW][NT];
    {
        const float* wc = params + (size_t)am.net[p] * S::NPARAM;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int tau = wave * TPW + u;
            tp_load_fwd<S, NT>(wc, tau, lane, cw[u]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 4 * g + r;
                t3[u][r] = a < A ? wc[S::oW3 + a * H + 16 * tau + j] : 0.f;  // A[i=h2 16tau+i][k=a]
#pragma unroll
                for (int kap = 0; kap < NT; ++kap) t2[u][kap][r] = wc[S::oW2 + (16 * kap + 4 * g + r) * H + 16 * tau + j];
            }
        }
    }
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4 dW2[TPW][NT], dW1[TPW][NT1], dW3[TPW], db1[TPW], db2[TPW], db3 = zero4;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        dW3[u] = zero4; db1[u] = zero4; db2[u] = zero4;
#pragma unroll
        for (int k = 0; k < NT; ++k) dW2[u][k] = zero4;
#pragma unroll
        for (int k = 0; k < NT1; ++k) dW1[u][k] = zero4;
    }
    float loss_acc = 0.f, nfill_acc = 0.f;

    TpSrc<S, REPLAY> src;
    src.obs_p = REPLAY ? nullptr : bt.obss + (size_t)p * (bt.obs_agent_stride > 0 ? (size_t)bt.obs_agent_stride : (bt.obs_agent_stride < 0 ? 0 : (size_t)(T + 1) * B * D));
    src.mask_p = nullptr;
    src.obs_rs = bt.obs_row_stride ? (size_t)bt.obs_row_stride : (size_t)D;
    src.act_p = (REPLAY || FULL) ? nullptr : bt.actions + (size_t)p * T * B;
    src.rew_p = (REPLAY || FULL) ? nullptr : bt.rewards + (size_t)p * T * B;
    src.dones = bt.dones; src.filled = bt.filled; src.rs = rs; src.P = P; src.p = p; src.T = T; src.B = B;

    const int nsets = (B + 16 * NB - 1) / (16 * NB);
    const int ntasks = nsets * n_chunks;
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const int set = task / n_chunks, c = task - set * n_chunks;
        const int t0 = (c * T) / n_chunks, t1 = ((c + 1) * T) / n_chunks;
        if (t1 <= t0) continue;
        int ej[NB], eg[NB][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) tp_episode_ids<REPLAY>(rs, (set * NB + nb) * 16, B, g, j, ej[nb], eg[nb]);
        TpRows<S, REPLAY> cur[NB];
        f4 h2c[STORED ? NB : 1][TPW], h1c[STORED1 ? NB : 1][TPW];
        auto h2_at = [&](int t, int nb, int u) -> f4 {
            return h2_in[((((size_t)p * T + t) * tp_h2_blocks(B) + (set * NB + nb)) * NT + wave * TPW + u) * 64 + lane];
        };
        auto h1_at = [&](int t, int nb, int u) -> f4 {
            return h1_in[((((size_t)p * T + t) * tp_h2_blocks(B) + (set * NB + nb)) * NT + wave * TPW + u) * 64 + lane];
        };
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            tp_load_rows<S, REPLAY, true, FULL, STORED1>(src, mix, t1 - 1, (set * NB + nb) * 16, g, j, ej[nb], eg[nb], cur[nb]);
            if constexpr (STORED) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) h2c[nb][u] = h2_at(t1 - 1, nb, u);
            }
            if constexpr (STORED1) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) h1c[nb][u] = h1_at(t1 - 1, nb, u);
            }
        }
        for (int t = t1 - 1; t >= t0; --t) {
            TpRows<S, REPLAY> nxt[NB];
            f4 h2n[STORED ? NB : 1][TPW], h1n[STORED1 ? NB : 1][TPW];
            f4 h1[NB][TPW];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                tp_load_rows<S, REPLAY, true, FULL, STORED1>(src, mix, t > t0 ? t - 1 : t0, (set * NB + nb) * 16, g, j, ej[nb], eg[nb], nxt[nb]);
                if constexpr (STORED) {
#pragma unroll
                    for (int u = 0; u < TPW; ++u) h2n[nb][u] = h2_at(t > t0 ? t - 1 : t0, nb, u);
                }
                if constexpr (STORED1) {
#pragma unroll
                    for (int u = 0; u < TPW; ++u) h1n[nb][u] = h1_at(t > t0 ? t - 1 : t0, nb, u);
                }
                tp_mask_rows<S, REPLAY, true, STORED1>(cur[nb], (set * NB + nb) * 16, B, g, j);
            }
            float* HcT = (STORED && (t & 1)) ? set1 : HcT0;
            f4* G2 = reinterpret_cast<f4*>(HcT + NB * H * 16);
            // ---- layer 1 of my tiles: C-layout dump (layer-2 operand) + [h][row] tile (dW2 operand)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    f4 acc;
                    if constexpr (STORED1) {
                        acc = h1c[nb][u];  // relu'd by the forward-rows pass (a padding row block holds the finite values of zero rows)
                    } else {
                        acc = cw[u].b1s;
#pragma unroll
                        for (int ks = 0; ks < S::KS1; ++ks) acc = MARL_MFMA(cw[u].a1[ks], cur[nb].x[ks], acc);
                        acc = relu4_one(acc);
                    }
                    h1[nb][u] = acc;
                    const int tau = wave * TPW + u;
                    if constexpr (!STORED) Hc[(nb * NT + tau) * 64 + lane] = acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) HcT[(nb * H + 16 * tau + 4 * g + r) * 16 + j] = acc[r];
                }
            if constexpr (!STORED) __syncthreads();  // (STORED: nothing of another wave is read before the barrier behind dH2)
            if constexpr (STORED) {
                // The two phases in BATCHED form: every private tile of the step (per row block nb and owned hidden tile u) has its own LDS
                // slot, so all of a phase's tile writes go out together, ONE wave fence later all reads come back together, and the
                // (nb, u) MFMA chains are interleaved instead of each waiting out its own LDS round trips (12 fences per step before).
                // Per accumulator the order of products is unchanged (nb ascending, k ascending): bitwise the same gradient.
                f4 dQ[NB][1], d2[NB][TPW];
                wave_lds_fence();  // the previous step's reads of these slots have retired
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int a_sel = cur[nb].a_sel;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dQ[nb][0][r] = FULL ? cur[nb].dqv[r] : ((4 * g + r == a_sel) ? cur[nb].dq : 0.f);
                    if (wave == 0) {
                        db3 += dQ[nb][0];
                        if (g == 0 && p == 0) { loss_acc += cur[nb].lr; nfill_acc += cur[nb].fl; }
                    }
                    tile_write<1>(PQ + 256 * nb, dQ[nb], g, j);
#pragma unroll
                    for (int u = 0; u < TPW; ++u) {
                        f4 hh[1] = {h2c[nb][u]};
                        tile_write<1>(PH2 + 256 * (nb * TPW + u), hh, g, j);
                        if constexpr (!(MARL_TP_VGPR && NB == 2 && TPW == 2)) d2[nb][u] = zero4;
                    }
                }
                if constexpr (MARL_TP_VGPR && NB == 2 && TPW == 2) {
                    tp_mfma4_zero(d2, t3[0][0], t3[1][0], dQ[0][0][0], dQ[1][0][0]);
#pragma unroll
                    for (int r = 1; r < 4; ++r) tp_mfma4(d2, t3[0][r], t3[1][r], dQ[0][0][r], dQ[1][0][r]);
                    tp_settle4(d2);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int u = 0; u < TPW; ++u) d2[nb][u] = MARL_MFMA(t3[u][r], dQ[nb][0][r], d2[nb][u]);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int u = 0; u < TPW; ++u) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) d2[nb][u][r] = h2c[nb][u][r] > 0.f ? d2[nb][u][r] : 0.f;
                        db2[u] += d2[nb][u];
                        G2[(nb * NT + wave * TPW + u) * 64 + lane] = d2[nb][u];
                    }
                wave_lds_fence();
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const f4 aQ = tile_read(PQ + 256 * nb, 0, g, j);
                    f4 bH2[TPW];
#pragma unroll
                    for (int u = 0; u < TPW; ++u) bH2[u] = tile_read(PH2 + 256 * (nb * TPW + u), 0, g, j);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int u = 0; u < TPW; ++u) dW3[u] = MARL_MFMA(aQ[ks], bH2[u][ks], dW3[u]);
                }
                __syncthreads();
                // ---- dH1 = W2^T dH2 (mask): NB x TPW independent chains, k ascending in each
                f4 d1[NB][TPW];
                constexpr bool VGB = MARL_TP_VGPR && NB == 2 && TPW == 2;
                if constexpr (!VGB) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int u = 0; u < TPW; ++u) d1[nb][u] = zero4;
                }
#pragma unroll
                for (int kap = 0; kap < NT; ++kap) {
                    f4 gk[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) gk[nb] = G2[(nb * NT + kap) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (VGB) {
                            if (kap == 0 && r == 0) tp_mfma4_zero(d1, t2[0][0][0], t2[1][0][0], gk[0][0], gk[1][0]);
                            else tp_mfma4(d1, t2[0][kap][r], t2[1][kap][r], gk[0][r], gk[1][r]);
                        } else {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                for (int u = 0; u < TPW; ++u) d1[nb][u] = MARL_MFMA(t2[u][kap][r], gk[nb][r], d1[nb][u]);
                        }
                    }
                }
                if constexpr (VGB) tp_settle4(d1);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int u = 0; u < TPW; ++u) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) d1[nb][u][r] = h1[nb][u][r] > 0.f ? d1[nb][u][r] : 0.f;
                        db1[u] += d1[nb][u];
                        f4 own[1] = {G2[(nb * NT + wave * TPW + u) * 64 + lane]}, mine[1] = {d1[nb][u]};
                        tile_write<1>(P2 + 256 * (nb * TPW + u), own, g, j);   // transposed A operands: my dH2 tile (from its C dump) ...
                        tile_write<1>(P1 + 256 * (nb * TPW + u), mine, g, j);  // ... and my dH1 tile
                    }
                wave_lds_fence();
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f4 aG2[TPW], aG1[TPW];
#pragma unroll
                    for (int u = 0; u < TPW; ++u) {
                        aG2[u] = tile_read(P2 + 256 * (nb * TPW + u), 0, g, j);
                        aG1[u] = tile_read(P1 + 256 * (nb * TPW + u), 0, g, j);
                    }
#pragma unroll
                    for (int nu = 0; nu < NT; ++nu) {
                        const f4 bH1 = tile_read(HcT + nb * H * 16, nu, g, j);
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                            for (int u = 0; u < TPW; ++u) dW2[u][nu] = MARL_MFMA(aG2[u][ks], bH1[ks], dW2[u][nu]);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                            for (int u = 0; u < TPW; ++u) dW1[u][nt] = MARL_MFMA(aG1[u][ks], cur[nb].bx[nt][ks], dW1[u][nt]);
                }
            } else {
            // ---- layer 2, dH2 = W3^T dQ (mask), dW3, dumps of dH2
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int a_sel = cur[nb].a_sel;
                f4 dQ[1];
#pragma unroll
                for (int r = 0; r < 4; ++r) dQ[0][r] = FULL ? cur[nb].dqv[r] : ((4 * g + r == a_sel) ? cur[nb].dq : 0.f);
                if (wave == 0) {
                    db3 += dQ[0];
                    if (g == 0 && p == 0) { loss_acc += cur[nb].lr; nfill_acc += cur[nb].fl; }
                }
                wave_lds_fence();
                tile_write<1>(PQ, dQ, g, j);
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    f4 h2[1];
                    if constexpr (STORED) {
                        h2[0] = h2c[nb][u];
                    } else {
                        f4 acc = cw[u].b2s;
#pragma unroll
                        for (int kap = 0; kap < NT; ++kap) {
                            const f4 hk = Hc[(nb * NT + kap) * 64 + lane];
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc = MARL_MFMA(cw[u].a2[kap][r], hk[r], acc);
                        }
                        h2[0] = relu4_one(acc);
                    }
                    f4 d2[1];
                    d2[0] = zero4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) d2[0] = MARL_MFMA(t3[u][r], dQ[0][r], d2[0]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) d2[0][r] = h2[0][r] > 0.f ? d2[0][r] : 0.f;
                    db2[u] += d2[0];
                    G2[(nb * NT + wave * TPW + u) * 64 + lane] = d2[0];
                    wave_lds_fence();
                    tile_write<1>(PH2 + 256 * u, h2, g, j);
                    wave_lds_fence();
                    const f4 aQ = tile_read(PQ, 0, g, j);
                    const f4 bH2 = tile_read(PH2 + 256 * u, 0, g, j);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) dW3[u] = MARL_MFMA(aQ[ks], bH2[ks], dW3[u]);
                }
                wave_lds_fence();  // PQ / PH2 are rewritten by the next block
            }
            __syncthreads();
            // ---- dH1 = W2^T dH2 (mask), dW2, dW1
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const int tau = wave * TPW + u;
                    f4 d1[1];
                    d1[0] = zero4;
#pragma unroll
                    for (int kap = 0; kap < NT; ++kap) {
                        const f4 gk = G2[(nb * NT + kap) * 64 + lane];
#pragma unroll
                        for (int r = 0; r < 4; ++r) d1[0] = MARL_MFMA(t2[u][kap][r], gk[r], d1[0]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) d1[0][r] = h1[nb][u][r] > 0.f ? d1[0][r] : 0.f;
                    db1[u] += d1[0];
                    // transposed A operands: my dH2 tile (from its C dump) and my dH1 tile
                    f4 d2[1];
                    d2[0] = G2[(nb * NT + tau) * 64 + lane];
                    wave_lds_fence();
                    tile_write<1>(P2 + 256 * u, d2, g, j);
                    tile_write<1>(P1 + 256 * u, d1, g, j);
                    wave_lds_fence();
                    const f4 aG2 = tile_read(P2 + 256 * u, 0, g, j);
                    const f4 aG1 = tile_read(P1 + 256 * u, 0, g, j);
#pragma unroll
                    for (int nu = 0; nu < NT; ++nu) {
                        const f4 bH1 = tile_read(HcT + nb * H * 16, nu, g, j);
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) dW2[u][nu] = MARL_MFMA(aG2[ks], bH1[ks], dW2[u][nu]);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) dW1[u][nt] = MARL_MFMA(aG1[ks], cur[nb].bx[nt][ks], dW1[u][nt]);
                }
            }
            __syncthreads();  // Hc / HcT / G2 are rewritten by the next step (STORED: the next step uses the other set, no barrier)
            }  // !STORED
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                cur[nb] = nxt[nb];
                if constexpr (STORED) {
#pragma unroll
                    for (int u = 0; u < TPW; ++u) h2c[nb][u] = h2n[nb][u];
                }
                if constexpr (STORED1) {
#pragma unroll
                    for (int u = 0; u < TPW; ++u) h1c[nb][u] = h1n[nb][u];
                }
            }
        }
        if constexpr (STORED) __syncthreads();  // the next task may start on either set
    }

    // ---- every wave owns disjoint slices of the gradient: straight to the workgroup's partial record
    float* rec = partials + ((size_t)p * gridDim.x + blockIdx.x) * (S::NPARAM + 2);
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int tau = wave * TPW + u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 16 * tau + 4 * g + r;
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) {
                const int d = 16 * nt + j;
                if (d < D) rec[S::oW1 + o * D + d] = dW1[u][nt][r];
            }
#pragma unroll
            for (int nu = 0; nu < NT; ++nu) rec[S::oW2 + o * H + 16 * nu + j] = dW2[u][nu][r];
            const int a = 4 * g + r;
            if (a < A) rec[S::oW3 + a * H + 16 * tau + j] = dW3[u][r];
            const float s1 = sum16(db1[u][r]), s2 = sum16(db2[u][r]);
            if (j == 0) {
                rec[S::ob1 + o] = s1;
                rec[S::ob2 + o] = s2;
            }
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = 4 * g + r;
            const float s3 = sum16(db3[r]);
            if (j == 0 && a < A) rec[S::ob3 + a] = s3;
        }
        const float ls = sum16(loss_acc), ns = sum16(nfill_acc);
        if (lane == 0) {
            rec[S::NPARAM] = ls;
            rec[S::NPARAM + 1] = ns;
        }
    }
}

}  // namespace marl
