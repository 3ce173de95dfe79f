// Kernels and launchers of the actor-critic learner step, shared by a2c.hip (feed-forward networks) and gru_ac.hip (recurrent).
#pragma once
#include "dqn_update_kernels.h"
#include "collect_common.h"
#include "gru_rows.h"
#include "wide_mlp.h"
#include "wide_critic.h"

#ifndef MARL_TP_NB1_D
#define MARL_TP_NB1_D 48  // observation rows wider than this walk ONE row block per step in tp_bwd_kernel (register budget; re-measured in round 4 on the
                         // 71-wide warehouse rows: two blocks spill 42 - 102 registers and the IA2C update goes 18.0 -> 19.0 ms, scripts/gpu_runs/r4AA.sh)
#endif

#ifndef MARL_TP_NB1S_D
#define MARL_TP_NB1S_D 80  // the same threshold for the form that reads both hidden layers back (no layer-1 registers)
#endif

namespace marl {

// A centralised critic whose input is too wide for the register-resident kernels (wide_critic.h: layer 1 streamed through LDS, the three
// layers fused): hidden width H fixed by the type, the input width set by the entry point for the duration of its call (thread-local:
// calls on different host threads never meet).
template <int H_>
struct WideCritic {
    static constexpr int H = H_, A = 1;
    static inline thread_local int D = 0;
    static WideNet net() { return WideNet{D, H, A, 2}; }
};
// both widths, inputs and outputs at run time (actors AND critics without a fused kernel: hidden > 128, other observation widths);
// TAG keeps the actor's and the critic's thread-local shapes apart
template <int TAG>
struct WideRt {
    static inline thread_local int D = 0, H = 0, A = 0, L = 2;
    static WideNet net() { return WideNet{D, H, A, L}; }
    static void set(int d, int h, int a, int l) { D = d; H = h; A = a; L = l; }
};
template <class S>
struct IsWide : std::false_type {};
template <int H>
struct IsWide<WideCritic<H>> : std::true_type {};
template <int TAG>
struct IsWide<WideRt<TAG>> : std::true_type {};
template <class S>
struct IsWideCritic : std::false_type {};
template <int H>
struct IsWideCritic<WideCritic<H>> : std::true_type {};

// bytes of forward-pack scratch (collect_pack_scratch) a forward-rows launch of shape S over n_rows rows wants
// (B, L: recurrent stacks - L pack sets and the chain records of a pass that keeps none, gru_rows.h)
template <class S>
int64_t forward_scratch_bytes(int P, int n_rows, int B = 0, int L = 1) {
    if constexpr (IsWideCritic<S>::value) return wc_pack_bytes(P, S::D, S::H) + 16;
    else if constexpr (IsWide<S>::value) return wide_ws(S::net(), P, n_rows, false).total + 16;
    else if constexpr (IsGru<S>::value) return gru_forward_scratch_bytes<S>(P, n_rows / (B > 0 ? B : 1), B, L);
    else return (int64_t)P * S::NFWD * 4 + 16;
}

// every fused MLP shape keeps hidden layers from its forward-rows pass for its backward-rows pass: h2 in the tensor-parallel layout
// (hidden 128), h1 | h2 in the LDS-resident learner's (hidden 64: dqn_lossgrad_kernel<MODE 4, STORED>)
template <class S>
constexpr bool mlp_stored_shape() { return !IsGru<S>::value && !IsWide<S>::value; }
template <class S>
inline int64_t mlp_stored_floats(int P, int T, int B) {
    if constexpr (!mlp_stored_shape<S>()) return 0;
    else if constexpr (use_tp<S>()) return 2 * tp_h2_floats(P, T, B, S::H);  // h2 | h1, both in tp_bwd_kernel's layout
    else return lds_h_floats(P, T, B, S::H);
}

// ---- forward rows ------------------------------------------------------------------------------------------
// H2: also store the second hidden layer of every row block in the layout tp_bwd_kernel<STORED> reads (rows = t * B + b, B % 16 == 0:
// row block bk = (t, b0 / 16)): h2_out[(((p T + t) tp_h2_blocks(B) + blk) MT + tile) 64 + lane]
// H2 = 2: the LDS-resident learner's layout instead, both hidden layers: h_out[(((p T + t) (B / 16) + block) 2 MT + layer MT + tile) 64 + lane]
template <class S, int H2 = 0>
__global__ __launch_bounds__(256) void mlp_rows_fwd_kernel(const float* __restrict__ packs /* pre-packed [P][NFWD] */, const float* __restrict__ obs,
                                                           size_t agent_stride, size_t row_stride, int n_rows,
                                                           float* __restrict__ out, f4* __restrict__ h2_out = nullptr, int T = 0, int B = 0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int p = blockIdx.y;
    stage_packed<S>(packs + (size_t)p * S::NFWD, lds, tid, 256);
    __syncthreads();
    const float* obs_p = obs + (size_t)p * agent_stride;
    const int nblk = (n_rows + 15) >> 4, npair = (nblk + 1) >> 1;
    for (int pr = blockIdx.x * 4 + wave; pr < npair; pr += gridDim.x * 4) {  // two row blocks per wave and step
        float x[2][S::KS1];
        int row[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            row[h] = (2 * pr + h) * 16 + j;
            const bool ok = row[h] < n_rows;
            const float* xrow = obs_p + (size_t)(ok ? row[h] : n_rows - 1) * row_stride;
#pragma unroll
            for (int ks = 0; ks < S::KS1; ++ks) {
                const int d = 4 * ks + g;
                const float v = xrow[d < S::D ? d : S::D - 1];
                x[h][ks] = (d < S::D && ok) ? v : 0.f;
            }
        }
        f4 q[2];
        if constexpr (H2 == 2) {
            f4 h2[2][S::MT], h1[2][S::MT];
            mlp_forward_p2<S, true, true>(lds, lane, x, q, h2, h1);
            const int bpt = B >> 4;  // row blocks per time step
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int bk = 2 * pr + h;
                if (bk < nblk) {
                    const int t = bk / bpt, blk = bk - t * bpt;
                    f4* dst = h2_out + ((((size_t)p * T + t) * bpt + blk) * (2 * S::MT)) * 64 + lane;
#pragma unroll
                    for (int mt = 0; mt < S::MT; ++mt) {
                        dst[mt * 64] = h1[h][mt];
                        dst[(S::MT + mt) * 64] = h2[h][mt];
                    }
                }
            }
        } else if constexpr (H2 == 1) {
            f4 h2[2][S::MT], h1[2][S::MT];
            mlp_forward_p2<S, true, true>(lds, lane, x, q, h2, h1);
            const int bpt = B >> 4;  // row blocks per time step
            const size_t h1_off = (size_t)gridDim.y * T * tp_h2_blocks(B) * S::MT * 64;  // f4 units: the h1 record sits behind the h2 record
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int bk = 2 * pr + h;
                if (bk < nblk) {
                    const int t = bk / bpt, blk = bk - t * bpt;
                    f4* dst = h2_out + ((((size_t)p * T + t) * tp_h2_blocks(B) + blk) * S::MT) * 64 + lane;
#pragma unroll
                    for (int mt = 0; mt < S::MT; ++mt) {
                        dst[mt * 64] = h2[h][mt];
                        dst[h1_off + mt * 64] = h1[h][mt];
                    }
                }
            }
        } else {
            mlp_forward_p2<S>(lds, lane, x, q);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (row[h] < n_rows) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < S::A) out[((size_t)p * n_rows + row[h]) * S::A + 4 * g + r] = q[h][r];
            }
        }
    }
    if constexpr (H2 == 1) {
        // tp_bwd_kernel walks row blocks in PAIRS: with an odd number of blocks per time step the pair's second slot holds no rows - its
        // gradients are masked, but 0 x (whatever the workspace held) must stay 0: the slot is zero-filled (both records)
        const int bpt = B >> 4;
        if ((bpt & 1) && tp_h2_blocks(B) > bpt) {
            const size_t h1_off = (size_t)gridDim.y * T * tp_h2_blocks(B) * S::MT * 64;
            const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
            for (int t = blockIdx.x * 4 + wave; t < T; t += gridDim.x * 4) {
                f4* dst = h2_out + ((((size_t)p * T + t) * tp_h2_blocks(B) + bpt) * S::MT) * 64 + lane;
#pragma unroll
                for (int mt = 0; mt < S::MT; ++mt) {
                    dst[mt * 64] = zero4;
                    dst[h1_off + mt * 64] = zero4;
                }
            }
        }
    }
}

template <class S>
int launch_forward_rows(int P, const AgentMap& am, const float* params, const marlhip_batch* bt, int n_rows, float* out, hipStream_t st,
                        float* rec = nullptr) {
    if constexpr (IsGru<S>::value) {
        return gru_forward_rows<S>(P, am, params, bt, n_rows / bt->batch, out, st, rec);
    } else if constexpr (IsWideCritic<S>::value) {
        MARL_REQUIRE(S::D > 0, "wide critic: input width not set");
        const int64_t as = bt->obs_agent_stride > 0 ? bt->obs_agent_stride : (bt->obs_agent_stride < 0 ? 0 : (int64_t)(bt->max_len + 1) * bt->batch * S::D);
        const int64_t rs = bt->obs_row_stride ? bt->obs_row_stride : S::D;
        MARL_REQUIRE(rec == nullptr || n_rows == bt->max_len * bt->batch, "wide critic forward: a hidden-layer record needs n_rows == T*B");
        return wc_forward_rows<S::H>(P, am, params, S::D, bt->obss, as, rs, n_rows, out, rec, st);
    } else if constexpr (IsWide<S>::value) {
        (void)rec;
        const WideNet s = S::net();
        MARL_REQUIRE(s.D > 0, "wide critic: input width not set");
        const int64_t as = bt->obs_agent_stride > 0 ? bt->obs_agent_stride : (bt->obs_agent_stride < 0 ? 0 : (int64_t)(bt->max_len + 1) * bt->batch * s.D);
        const int64_t rs = bt->obs_row_stride ? bt->obs_row_stride : s.D;
        // as many rows per pass as the caller's scratch holds activations for (the learner step sizes it for all rows at once)
        const int64_t avail = scratch_avail(), per_row = (int64_t)s.L * s.H * 4;
        int chunk = (int)((avail - 1024) / per_row < n_rows ? (avail - 1024) / per_row : n_rows);
        chunk &= ~63;
        if (chunk >= n_rows || n_rows < 64) chunk = n_rows;
        MARL_REQUIRE(chunk >= 64 || chunk == n_rows, "wide critic forward: the workspace holds no 64-row slice of activations (%lld bytes)", (long long)avail);
        float* scratch = collect_pack_scratch((size_t)wide_ws(s, P, chunk, false).total, st);
        if (scratch == nullptr) return -1;
        for (int r0 = 0; r0 < n_rows; r0 += chunk) {
            const int nr = n_rows - r0 < chunk ? n_rows - r0 : chunk;
            // out is [P][n_rows][A]: a slice keeps the full agent stride, so slices go agent by agent through one-agent views
            for (int p = 0; p < P; ++p) {
                AgentMap one = am;
                one.net[0] = am.net[p];
                const int rc = wide_forward_rows(s, 1, one, params, bt->obss + (int64_t)p * as + (int64_t)r0 * rs, 0, rs, nr,
                                                 out + ((int64_t)p * n_rows + r0) * s.A, scratch, st);
                if (rc != 0) return rc;
            }
        }
        return 0;
    } else {
    const int T = bt->max_len, B = bt->batch;
    // the stored-hidden-layer record is [T*B/16 blocks]: a longer row range (the target critic's T*B + B) would write a time step past it
    MARL_REQUIRE(rec == nullptr || (n_rows == T * B && B % 16 == 0), "forward rows: a hidden-layer record needs n_rows == T*B (%d vs %d x %d) and B %% 16 == 0", n_rows, T, B);
    const size_t as = bt->obs_agent_stride > 0 ? (size_t)bt->obs_agent_stride : (bt->obs_agent_stride < 0 ? 0 : (size_t)(T + 1) * B * S::D);
    const size_t rs = bt->obs_row_stride ? (size_t)bt->obs_row_stride : (size_t)S::D;
    constexpr int LDSB = S::NFWD * (int)sizeof(float);
    static LdsAttr attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_rows_fwd_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        attr_set.done();
    }
    const int npair = ((n_rows + 15) / 16 + 1) / 2;
    int gx = (npair + 3) / 4;
    // one resident workgroup per CU when the pack fills most of the LDS (hidden 128), two otherwise: every workgroup stages once
    const int per_cu = LDSB > 80 * 1024 ? 1 : 2;
    const int cap = (256 * per_cu) / P > 1 ? (256 * per_cu) / P : 1;
    if (gx > cap) gx = cap;
    float* packs = nullptr;
    if (launch_fwd_pack<S>(P, am, params, &packs, st) != 0) return -1;
    if (rec != nullptr) {  // leave the hidden layers for the backward-rows pass (the caller checked n_rows == T * B and B % 16 == 0)
        constexpr int HS = use_tp<S>() ? 1 : 2;  // tp_bwd_kernel<STORED>'s h2 | dqn_lossgrad_kernel<MODE 4, STORED>'s h1 | h2
        static LdsAttr attr_h2;
        if (attr_h2.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_rows_fwd_kernel<S, HS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
            attr_h2.done();
        }
        hipLaunchKernelGGL((mlp_rows_fwd_kernel<S, HS>), dim3(gx, P), dim3(256), LDSB, st, (const float*)packs, bt->obss, as, rs, n_rows, out,
                           reinterpret_cast<f4*>(rec), T, B);
        MARL_CHECK_LAUNCH("mlp_rows_fwd_kernel<hidden layers kept>");
        return 0;
    }
    hipLaunchKernelGGL((mlp_rows_fwd_kernel<S>), dim3(gx, P), dim3(256), LDSB, st, (const float*)packs, bt->obss, as, rs, n_rows, out);
    MARL_CHECK_LAUNCH("mlp_rows_fwd_kernel");
    return 0;
    }
}

// ---- backward rows -----------------------------------------------------------------------------------------
template <class S>
int64_t backward_ws_bytes(int P, int T, int B, int L = 1) {
    if constexpr (IsGru<S>::value) {
        return gru_rows_ws<S>(P, T, B, false, L).total;  // the step's forward passes write the records (AcWs::rec_a / rec_c)
    } else if constexpr (IsWideCritic<S>::value) {
        return wc_ws(P, T * B, S::D, S::H).total;
    } else if constexpr (IsWide<S>::value) {
        return wide_ws(S::net(), P, T * B, true).total;
    } else if constexpr (use_tp<S>()) {
        const UpdPlan a = upd_plan_tp(P, T, B, S::D > MARL_TP_NB1_D ? 1 : 2), b = upd_plan_tp(P, T, B, S::D > MARL_TP_NB1S_D ? 1 : 2, MARL_TP_BWD_OCC);  // either form of the pass
        return ws_layout(P, a.nwg > b.nwg ? a.nwg : b.nwg, S::NPARAM + 2, 0, T, B).total;
    } else {
        const UpdPlan pl = upd_plan(P, T, B);
        return ws_layout(P, pl.nwg, UpdLds<S>::REC, 2 * S::NFWD + S::NBWD, T, B).total;
    }
}

// grad[P][NPARAM] = d(sum_rows lrow-loss)/dparams / sum(filled) from dout[P][T][B][A]; loss[0] = sum(lrow)/sum(filled)
template <class S>
int launch_backward_rows(int P, const AgentMap& am, const float* params, const marlhip_batch* bt, const float* dout, float* lrow, void* ws,
                         int64_t ws_bytes, float* grad, float* loss, hipStream_t st, const float* rec = nullptr) {
    if constexpr (IsGru<S>::value) {
        MARL_REQUIRE(rec != nullptr, "ac backward: the recurrent networks need the forward record");
        return gru_backward_rows<S>(P, am, params, bt, bt->max_len, dout, lrow, ws, ws_bytes, grad, loss, st, rec);
    } else if constexpr (IsWideCritic<S>::value) {
        const int T = bt->max_len, B = bt->batch;
        MARL_REQUIRE(rec != nullptr, "ac backward: the wide critics need the hidden layers of this step's forward pass");
        MARL_REQUIRE(ws_bytes >= backward_ws_bytes<S>(P, T, B), "ac backward: workspace %lld too small", (long long)ws_bytes);
        const int64_t as = bt->obs_agent_stride > 0 ? bt->obs_agent_stride : (bt->obs_agent_stride < 0 ? 0 : (int64_t)(T + 1) * B * S::D);
        const int64_t rs = bt->obs_row_stride ? bt->obs_row_stride : S::D;
        return wc_backward_rows<S::H>(P, am, params, S::D, bt->obss, as, rs, T * B, bt->filled, dout, lrow, ws, grad, loss, st, rec);
    } else if constexpr (IsWide<S>::value) {
        (void)rec;
        const WideNet s = S::net();
        const int T = bt->max_len, B = bt->batch;
        MARL_REQUIRE(ws_bytes >= backward_ws_bytes<S>(P, T, B), "ac backward: workspace %lld too small", (long long)ws_bytes);
        const int64_t as = bt->obs_agent_stride > 0 ? bt->obs_agent_stride : (bt->obs_agent_stride < 0 ? 0 : (int64_t)(T + 1) * B * s.D);
        const int64_t rs = bt->obs_row_stride ? bt->obs_row_stride : s.D;
        return wide_backward_rows(s, P, am, params, bt->obss, as, rs, T * B, bt->filled, dout, (int64_t)T * B * s.A, lrow, ws, grad, loss, st);
    } else {
    const int T = bt->max_len, B = bt->batch;
    MARL_REQUIRE(ws_bytes >= backward_ws_bytes<S>(P, T, B), "ac backward: workspace %lld too small", (long long)ws_bytes);
    ReplaySrc none = {};
    int nwg;
    if constexpr (use_tp<S>()) {
        // row blocks per step: the recomputing form keeps the layer-1 weights and both copies of the rows in registers and walks ONE block
        // per step on rows wider than 48 (two spill); with both hidden layers read back (STORED1) those registers are free and two blocks
        // fit up to 80-wide rows (the 71-wide warehouse rows: 503 registers, no scratch)
        constexpr int W = MARL_TP_W, TPW = S::H / (16 * W), NBR = S::D > MARL_TP_NB1_D ? 1 : 2, NBS = S::D > MARL_TP_NB1S_D ? 1 : 2;
        TpMix mix = {};
        mix.lrow = lrow;
        mix.dout = dout;
        const size_t ldsB = (size_t)tp_bwd_lds_floats<S, W, TPW, NBR, false>() * sizeof(float);
        const size_t ldsS = (size_t)tp_bwd_lds_floats<S, W, TPW, NBS, true>() * sizeof(float);
        static_assert(tp_bwd_lds_floats<S, W, TPW, NBS, true>() * 4 <= 160 * 1024, "tp_bwd_kernel<STORED>: LDS");
        static LdsAttr attr_set;
        if (attr_set.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_bwd_kernel<S, W, TPW, false, NBR, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_bwd_kernel<S, W, TPW, false, NBS, true, true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsS);
            attr_set.done();
        }
        if (rec != nullptr) {  // both hidden layers come back from the forward-rows pass of this step: h2 | h1 (mlp_rows_fwd_kernel<S, 1>)
            const UpdPlan pl = upd_plan_tp(P, T, B, NBS, MARL_TP_BWD_OCC);
            nwg = pl.nwg;
            hipLaunchKernelGGL((tp_bwd_kernel<S, W, TPW, false, NBS, true, true, true>), dim3(pl.nwg, P), dim3(64 * W), ldsS, st, params, am, *bt, none, mix,
                               pl.n_chunks, (float*)ws, reinterpret_cast<const f4*>(rec), reinterpret_cast<const f4*>(rec) + tp_h2_floats(P, T, B, S::H) / 4);
        } else {
            const UpdPlan pl = upd_plan_tp(P, T, B, NBR);
            nwg = pl.nwg;
            hipLaunchKernelGGL((tp_bwd_kernel<S, W, TPW, false, NBR, true>), dim3(pl.nwg, P), dim3(64 * W), ldsB, st, params, am, *bt, none, mix,
                               pl.n_chunks, (float*)ws);
        }
        MARL_CHECK_LAUNCH("tp_bwd_kernel<FULL>");
    } else {
        using L = UpdLds<S>;
        constexpr int PACK = 2 * S::NFWD + S::NBWD;
        const UpdPlan pl = upd_plan(P, T, B);
        nwg = pl.nwg;
        const WsLayout wl = ws_layout(P, pl.nwg, L::REC, PACK, T, B);
        float* packs = reinterpret_cast<float*>(static_cast<char*>(ws) + wl.pack_off);
        const size_t lds_bytes = (size_t)L::total(4) * sizeof(float);
        static LdsAttr attr_set;
        if (attr_set.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dqn_lossgrad_kernel<S, 4, false, 4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dqn_lossgrad_kernel<S, 4, false, 4, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            attr_set.done();
        }
        hipLaunchKernelGGL((dqn_pack_kernel<S>), dim3((PACK + 255) / 256, P), dim3(256), 0, st, params, params, am, packs);
        MixBufs mix = {};
        mix.lrow = lrow;
        mix.dout = dout;
        if (rec != nullptr)  // the forward-rows pass of this step left h1 | h2: no second forward (dqn_lossgrad_kernel<..., STORED>)
            hipLaunchKernelGGL((dqn_lossgrad_kernel<S, 4, false, 4, true>), dim3(pl.nwg, P), dim3(256), lds_bytes, st, (const float*)packs, *bt, none,
                               mix, 0.f, 0, pl.n_chunks, (float*)ws, (unsigned long long*)nullptr,
                               reinterpret_cast<f4*>(const_cast<float*>(rec)));
        else
            hipLaunchKernelGGL((dqn_lossgrad_kernel<S, 4, false, 4>), dim3(pl.nwg, P), dim3(256), lds_bytes, st, (const float*)packs, *bt, none,
                               mix, 0.f, 0, pl.n_chunks, (float*)ws, (unsigned long long*)nullptr);
        MARL_CHECK_LAUNCH("dqn_lossgrad_kernel<MODE 4>");
    }
    const int n = am.nblk * S::NPARAM;
    hipLaunchKernelGGL(dqn_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, (const float*)ws, P, nwg, S::NPARAM, am, grad, loss);
    MARL_CHECK_LAUNCH("dqn_reduce_kernel");
    return 0;
    }
}

// ---- the elementwise stage ---------------------------------------------------------------------------------
struct AcArgs {
    // mode 0: A2C; 1: PPO prepare (returns + old log-probs, no gradients); 2: PPO epoch (stored returns);
    // 3: A2C on stored returns; 4: returns only (3 and 4 bracket the statistics update of standardise_returns)
    int P, T, B, A, n_steps, mode;
    int standardise;                // returns are de-standardised / standardised with st (ac/model.py:195-204)
    float gk[18];                   // (float)(gamma ** k), k = 0..n_steps, formed in fp64 like python's gamma**step
    float ent_coef, vlc, ppo_clip;
};

struct AcBufs {
    const float* logits;  // [P][T*B][A]
    const float* v;       // [P][T*B]
    const float* vnext;   // [P][(T+1)*B] target-critic values of every observation
    float* dlogits;       // [P][T*B][A]
    float* dv;            // [P][T*B]
    float* lrow_a;        // [T*B] filled * actor-loss row
    float* lrow_v;        // [T*B] filled * value-loss row
    float* ent;           // [T*B] filled * sum_p entropy
    float* ret;           // [P][T*B]  PPO: returns kept across epochs
    float* oldlogp;       // [P][T*B]  PPO: log-prob under the pre-update policy
    float* partial;       // [blocks][4] per-block sums of (actor row, value row, entropy row, filled)
    float* rpartial;      // [blocks][P][2] per-block sums of the raw returns and their squares (standardise_returns)
    RetStats st;
};

static __global__ __launch_bounds__(256) void ac_elem_kernel(AcArgs a, marlhip_batch bt, AcBufs w) {
    const int TB = a.T * a.B;
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    const bool inside = i0 < TB;
    const int i = inside ? i0 : TB - 1;  // out-of-range threads shadow the last row and contribute nothing
    const int t = i / a.B, b = i - t * a.B;
    const size_t aas = bt.act_agent_stride ? (size_t)bt.act_agent_stride : (size_t)TB;
    const size_t ars = bt.act_row_stride ? (size_t)bt.act_row_stride : 1;
    const float fl = inside ? bt.filled[i] : 0.f;
    float la = 0.f, lv = 0.f, es = 0.f;
    for (int p = 0; p < a.P; ++p) {
        const size_t pi = (size_t)p * TB + i;
        float ret;
        if (a.mode == 2 || a.mode == 3) {
            ret = w.ret[pi];
            if (a.standardise) ret = (ret - w.st.mean[p]) / sqrtf(w.st.var[p]);  // with the statistics updated from this batch
        } else {  // compute_nstep_returns (utils/utils.py:38-63)
            ret = 0.f;
            for (int k = 0; k <= a.n_steps; ++k) {
                const int tt = t + k;
                if (tt >= a.T) break;
                const float nd = 1.f - bt.dones[(size_t)tt * a.B + b];
                if (k == a.n_steps) {
                    float vn = w.vnext[(size_t)p * (TB + a.B) + (size_t)tt * a.B + b];
                    if (a.standardise) vn = vn * sqrtf(w.st.var[p]) + w.st.mean[p];  // model.py:195-196, statistics BEFORE the update
                    ret += a.gk[k] * vn * nd;
                } else {
                    ret += a.gk[k] * bt.rewards[p * aas + ((size_t)tt * a.B + b) * ars] * nd;
                }
            }
            if ((a.mode == 1 || a.mode == 4) && inside) w.ret[pi] = ret;
            if (a.standardise && (a.mode == 1 || a.mode == 4)) {
                __shared__ float shr[8];
                ret_block_partials(inside ? ret : 0.f, shr, w.rpartial + ((size_t)blockIdx.x * a.P + p) * 2);
            }
            if (a.mode == 4) continue;
        }
        const float* lraw = w.logits + pi * a.A;
        // get_dist (ac/model.py:135-145): logits * mask + (1 - mask) * -1e8 with batch.action_masks[t] ([T+1][B][P][A])
        const float* mrow = bt.action_mask != nullptr ? bt.action_mask + (((size_t)t * a.B + b) * a.P + p) * a.A : nullptr;
        auto l = [&](int k) { return mrow != nullptr ? lraw[k] * mrow[k] + (1.f - mrow[k]) * -1e8f : lraw[k]; };  // re-read, no local array
        float m = l(0);
        for (int k = 1; k < a.A; ++k) m = fmaxf(m, l(k));
        float s = 0.f;
        for (int k = 0; k < a.A; ++k) s += expf(l(k) - m);
        const float lse = m + logf(s);
        const int act = (int)bt.actions[p * aas + (size_t)i * ars];
        const float logp = l(act) - lse;
        float H = 0.f;
        for (int k = 0; k < a.A; ++k) H -= expf(l(k) - lse) * (l(k) - lse);
        if (a.mode == 1) {
            if (inside) w.oldlogp[pi] = logp;
            continue;
        }
        const float val = w.v[pi], adv = ret - val;
        float coef;  // d(actor row loss) / d logp
        if (a.mode == 0 || a.mode == 3) {
            la += -logp * adv - a.ent_coef * H;
            coef = -adv;
        } else {  // clipped surrogate (model.py:318-327); min() ties split the gradient, clamp passes it inside the range
            const float ratio = expf(logp - w.oldlogp[pi]);
            const float rc = fminf(fmaxf(ratio, 1.f - a.ppo_clip), 1.f + a.ppo_clip);
            const float s1 = ratio * adv, s2 = rc * adv;
            la += -fminf(s1, s2) - a.ent_coef * H;
            const bool in_clip = ratio >= 1.f - a.ppo_clip && ratio <= 1.f + a.ppo_clip;
            const float share = s1 < s2 ? 1.f : (s1 == s2 ? (in_clip ? 1.f : 0.5f) : (0.f));
            coef = -adv * ratio * share;
        }
        if (inside) {
            for (int k = 0; k < a.A; ++k) {
                const float lp = l(k) - lse, pk = expf(lp);
                const float dl = fl * (coef * ((k == act ? 1.f : 0.f) - pk) + a.ent_coef * pk * (lp + H));
                w.dlogits[pi * a.A + k] = mrow != nullptr ? dl * mrow[k] : dl;  // d(masked logit)/d(logit) = mask
            }
            w.dv[pi] = fl * (-2.f * a.vlc * (ret - val));
        }
        lv += (ret - val) * (ret - val);
        es += H;
    }
    if (a.mode == 1 || a.mode == 4) return;
    if (inside) {
        w.lrow_a[i] = fl * la;
        w.lrow_v[i] = fl * lv;
        w.ent[i] = fl * es;
    }
    // per-block sums for the metrics (fixed order: wave butterfly, then waves 0..3)
    __shared__ float sh[4][4];
    float s4[4] = {fl * la, fl * lv, fl * es, fl};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s4[k] += __shfl_xor(s4[k], off);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = s4[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) w.partial[blockIdx.x * 4 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// metrics[0..4] = loss, actor_loss, value_loss, entropy, sum(filled)  (fixed-order tree: reproducible)
static __global__ __launch_bounds__(1024) void ac_metrics_kernel(int nblocks, float vlc, const float* __restrict__ partial,
                                                                 float* __restrict__ metrics) {
    __shared__ float sh[4][16];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nblocks; i += 1024) {
        s[0] += partial[4 * i]; s[1] += partial[4 * i + 1]; s[2] += partial[4 * i + 2]; s[3] += partial[4 * i + 3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[k] += __shfl_xor(s[k], off);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = s[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot[4];
        for (int k = 0; k < 4; ++k) {
            float acc = 0.f;
            for (int wv = 0; wv < 16; ++wv) acc += sh[k][wv];
            tot[k] = acc;
        }
        const float al = tot[0] / tot[3], vl = tot[1] / tot[3];
        metrics[0] = al + vlc * vl;
        metrics[1] = al;
        metrics[2] = vl;
        metrics[3] = tot[2] / tot[3];
        metrics[4] = tot[3];
    }
}

// The recurrent step overlaps the critics' sequence passes with the actors' on a second stream: one pass of the bench batch fills half
// of the SIMDs and the two are different kernels, so they overlap through a fork / join on events instead of a shared grid.  The
// stream is the CALLER's (marlhip_ac_config.side_stream; NULL = everything on the call's stream); the two events live for the
// duration of the call (created here, destroyed on return - a destroyed event's pending record still completes), and the join is
// waited for on every exit path so that no side-stream kernel outlives the caller's ordering of the buffers it reads.
struct SideStream {
    hipStream_t s = nullptr, main = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool forked = false;
    SideStream(void* side, hipStream_t main_st) : main(main_st) {
        if (side == nullptr || side == (void*)main_st) return;
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) { fork = nullptr; return; }
        if (hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(fork); fork = join = nullptr; return; }
        s = (hipStream_t)side;
    }
    bool ok() const { return s != nullptr; }
    void do_fork() {  // the side stream continues behind everything queued on the main stream so far
        (void)hipEventRecord(fork, main);
        (void)hipStreamWaitEvent(s, fork, 0);
        forked = true;
    }
    void do_join() {  // the main stream continues behind everything queued on the side stream so far
        if (!forked) return;
        (void)hipEventRecord(join, s);
        (void)hipStreamWaitEvent(main, join, 0);
        forked = false;
    }
    ~SideStream() {
        if (s == nullptr) return;
        do_join();
        (void)hipEventDestroy(fork);
        (void)hipEventDestroy(join);
    }
    SideStream(const SideStream&) = delete;
    SideStream& operator=(const SideStream&) = delete;
};

// workspace: [vnext | v | logits | dlogits | dv | lrow_a | lrow_v | ent | ret | oldlogp | loss scratch 4][backward workspace]
struct AcWs {
    int64_t vnext, v, logits, dlogits, dv, lrow_a, lrow_v, ent, ret, oldlogp, partial, rpartial, scratch, packs, packs_bytes, rec_a, rec_c, bwd, bwd_c, total;
};

template <class SA, class SC>
AcWs ac_ws_layout(int P, int T, int B, int L = 1, int Lc_ = 0) {  // L / Lc: stacked GRU layers of recurrent actors / critics (AgentMap::depth; Lc 0 = L)
    const int Lc = Lc_ > 0 ? Lc_ : L;
    const int64_t TB = (int64_t)T * B;
    AcWs w;
    int64_t o = 0;
    auto take = [&](int64_t nfloat) { const int64_t at = o; o = (o + nfloat * 4 + 255) & ~(int64_t)255; return at; };
    w.vnext = take((int64_t)P * (TB + B));
    w.v = take(P * TB);
    w.logits = take(P * TB * SA::A);
    w.dlogits = take(P * TB * SA::A);
    w.dv = take(P * TB);
    w.lrow_a = take(TB);
    w.lrow_v = take(TB);
    w.ent = take(TB);
    w.ret = take(P * TB);
    w.oldlogp = take(P * TB);
    w.partial = take(4 * ((TB + 255) / 256));
    w.rpartial = take(2 * P * ((TB + 255) / 256));
    w.scratch = take(8);
    // weight packs of the forward-rows launches (two networks at once for the paired recurrent pass): collect_pack_scratch's region
    {
        const int64_t fa = forward_scratch_bytes<SA>(P, (int)(TB + B), B, L), fc = forward_scratch_bytes<SC>(P, (int)(TB + B), B, Lc);
        w.packs_bytes = 2 * (fa > fc ? fa : fc);
    }
    w.packs = take(w.packs_bytes / 4 + 1);
    w.rec_a = w.rec_c = o;  // recurrent networks: the activation records of this step's actor / critic forward passes
    if constexpr (IsGru<SA>::value) w.rec_a = take(gru_rec_floats<SA>(P, T, B, L));
    else if constexpr (mlp_stored_shape<SA>()) w.rec_a = take(mlp_stored_floats<SA>(P, T, B));  // hidden layers of the actors' rows for their backward pass
    if constexpr (IsGru<SC>::value) w.rec_c = take(gru_rec_floats<SC>(P, T, B, Lc));
    else if constexpr (mlp_stored_shape<SC>()) w.rec_c = take(mlp_stored_floats<SC>(P, T, B));
    else if constexpr (IsWideCritic<SC>::value) w.rec_c = take(wc_rec_floats(P, (int)TB, SC::H));  // both hidden layers of the critics' rows
    w.bwd = o;
    const int64_t ba = backward_ws_bytes<SA>(P, T, B, L), bc = backward_ws_bytes<SC>(P, T, B, Lc);
    // recurrent networks: the two backward passes run side by side (side_stream) and need a workspace each
    w.bwd_c = IsGru<SA>::value && IsGru<SC>::value ? o + ((ba + 255) & ~(int64_t)255) : o;
    w.total = w.bwd_c != o ? w.bwd_c + bc : o + (ba > bc ? ba : bc);
    return w;
}

// ---- the actors' forward pass kept by the rollout (AcKeep, common.h) ----------------------------------------------------------
// tp_bwd_kernel walks row blocks in pairs: the empty second slot of an odd count per time step is zero-filled, as mlp_rows_fwd_kernel<S, 1>
// does for its own record
static __global__ __launch_bounds__(64) void ac_keep_pad_kernel(AcKeep k, int P, int MT) {
    const int lane = threadIdx.x, t = blockIdx.x, p = blockIdx.y;
    f4* hid = reinterpret_cast<f4*>(k.hid);
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const size_t slot = ((((size_t)p * k.T + t) * k.bpt + (k.B >> 4)) * k.stride) * 64 + lane;
    for (int mt = 0; mt < MT; ++mt) {
        hid[k.off_h1 + slot + mt * 64] = zero4;
        hid[k.off_h2 + slot + mt * 64] = zero4;
    }
}

// where a rollout of (T, B) leaves the actors' logits and hidden layers inside the learner step's workspace: AcWs::logits / AcWs::rec_a in the
// layout launch_backward_rows<SA> reads (the record mlp_rows_fwd_kernel<SA, HS> would write)
template <class SA, class SC>
int ac_keep_layout(int P, int T, int B, void* ws, int64_t ws_bytes, AcKeep* k, hipStream_t st) {
    if constexpr (!mlp_stored_shape<SA>()) {
        set_error("ac_collect_keep: the actors of this shape keep no forward pass (recurrent / GEMM-path networks)");
        return -1;
    } else {
        MARL_REQUIRE(T > 0 && B > 0 && B % 16 == 0, "ac_collect_keep: the hidden-layer record is laid out in blocks of 16 envs (%d envs)", B);
        const AcWs wl = ac_ws_layout<SA, SC>(P, T, B);
        MARL_REQUIRE(ws_bytes >= wl.total, "ac_collect_keep: learner workspace %lld < %lld bytes (marlhip_ac_workspace_bytes)", (long long)ws_bytes,
                     (long long)wl.total);
        char* base = static_cast<char*>(ws);
        k->logits = reinterpret_cast<float*>(base + wl.logits);
        k->hid = reinterpret_cast<float*>(base + wl.rec_a);
        k->T = T;
        k->B = B;
        if constexpr (use_tp<SA>()) {  // h2 record | h1 record (mlp_rows_fwd_kernel<S, 1>)
            k->bpt = tp_h2_blocks(B);
            k->stride = SA::MT;
            k->off_h2 = 0;
            k->off_h1 = (int64_t)P * T * k->bpt * SA::MT * 64;
            if (((B >> 4) & 1) && k->bpt > (B >> 4)) {
                hipLaunchKernelGGL(ac_keep_pad_kernel, dim3(T, P), dim3(64), 0, st, *k, P, (int)SA::MT);
                MARL_CHECK_LAUNCH("ac_keep_pad_kernel");
            }
        } else {  // h1 | h2 inside a row block's slot (mlp_rows_fwd_kernel<S, 2>)
            k->bpt = B >> 4;
            k->stride = 2 * SA::MT;
            k->off_h1 = 0;
            k->off_h2 = SA::MT * 64;
        }
        return 0;
    }
}

// SA / SC: the actors' and critics' shapes (MlpShape, or GruShape for recurrent networks)
template <class SA, class SC>
int ac_step_t(int P, const AgentMap& am, const float* actor, const float* critic, const float* target, const marlhip_batch* bt, const marlhip_ac_config* c,
              int mode, void* ws, int64_t ws_bytes, float* actor_grad, float* critic_grad, float* metrics, hipStream_t st) {
    const int D = SA::D, A = SA::A, DC = SC::D;  // (run-time values for the wide networks)
    const int T = bt->max_len, B = bt->batch, TB = T * B;
    // critic.parameter_sharing is its own setting (ac/model.py:68-97): the critics and target critics follow marlhip_ac_config's map when it is given
    AgentMap amc = am;
    if (c->critic_n_networks > 0) {
        MARL_REQUIRE(c->critic_n_networks <= P, "ac_loss_grad: %d critic networks for %d agents", c->critic_n_networks, P);
        amc.nblk = c->critic_n_networks;
        for (int i = 0; i < 16; ++i) {
            const int k = i < P ? c->critic_net_of[i] : 0;
            MARL_REQUIRE(k >= 0 && k < c->critic_n_networks, "ac_loss_grad: critic_net_of[%d] = %d out of range", i, k);
            amc.net[i] = (int8_t)k;
        }
    }
    marlhip_batch btc = *bt;  // the critics' view of the batch
    if (DC != D) btc.obs_agent_stride = -1;
    const marlhip_batch* bc = &btc;
    if constexpr (IsGru<SC>::value) {  // recurrent critics of their own depth (marlhip_ac_config.critic_n_hidden = len(critic.layers); 0: as the actors)
        if (c->critic_n_hidden > 0) amc.depth = (int8_t)(c->critic_n_hidden - 1);
    }
    const bool any_gru = IsGru<SA>::value || IsGru<SC>::value;
    const AcWs wl = ac_ws_layout<SA, SC>(P, T, B, any_gru ? am.depth : 1, any_gru ? amc.depth : 0);
    MARL_REQUIRE(ws_bytes >= wl.total, "ac_loss_grad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)wl.total);
    char* base = static_cast<char*>(ws);
    ScratchScope pack_scope(base + wl.packs, wl.packs_bytes);
    auto f = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
    AcBufs w;
    w.logits = f(wl.logits); w.v = f(wl.v); w.vnext = f(wl.vnext); w.dlogits = f(wl.dlogits); w.dv = f(wl.dv);
    w.lrow_a = f(wl.lrow_a); w.lrow_v = f(wl.lrow_v); w.ent = f(wl.ent); w.ret = f(wl.ret); w.oldlogp = f(wl.oldlogp);
    w.partial = f(wl.partial);
    w.rpartial = f(wl.rpartial);
    const bool std_on = c->ret_mean != nullptr;
    w.st.mean = c->ret_mean; w.st.var = c->ret_var; w.st.count = c->ret_count;
    w.st.exchange = c->ret_exchange; w.st.exchange_ctx = c->ret_exchange_ctx; w.st.moments = c->ret_moments;
    AcArgs a;
    a.P = P; a.T = T; a.B = B; a.A = A; a.n_steps = c->n_steps; a.mode = mode;
    for (int k = 0; k <= c->n_steps; ++k) a.gk[k] = (float)pow(c->gamma, (double)k);
    a.ent_coef = c->entropy_coef; a.vlc = c->value_loss_coef; a.ppo_clip = c->ppo_clip;
    a.standardise = std_on ? 1 : 0;
    int rc;
    timing_begin(TIMER_LOSSGRAD, st);
    float* rec_a = (IsGru<SA>::value || (mlp_stored_shape<SA>() && B % 16 == 0)) && mode != 1 ? f(wl.rec_a) : nullptr;
    float* rec_c = (IsGru<SC>::value || (mlp_stored_shape<SC>() && B % 16 == 0 && mode != 1) || (IsWideCritic<SC>::value && mode != 1)) ? f(wl.rec_c) : nullptr;
    bool v_done = false;
    // PPO's passes with recurrent networks (prepare: target critics + actors; epochs: actors + critics): the critics' sequence pass
    // goes to the side stream next to the actors' (each fills half of the SIMDs), with its packs in the critics' backward workspace
    SideStream side(IsGru<SA>::value && IsGru<SC>::value ? c->side_stream : nullptr, st);  // joins on every return path
    auto side_forward = [&](const float* prm, int steps, float* out, float* rec) -> int {  // -2: not available
        if constexpr (IsGru<SA>::value && IsGru<SC>::value) {
            if (!side.ok()) return -2;
            float* pk = reinterpret_cast<float*>(base + wl.bwd_c + gru_rows_ws<SC>(P, T, B, false, amc.depth).packF);
            side.do_fork();
            // (a stack's record-less pass - marlhip_gru_ppo_prepare's target critics - chains its layers through the critics' record space)
            return gru_forward_rows<SC>(P, amc, prm, bc, steps, out, side.s, rec, pk, rec == nullptr ? reinterpret_cast<float*>(base + wl.rec_c) : nullptr);
        } else {
            (void)prm; (void)steps; (void)out; (void)rec;
            return -2;
        }
    };
    if (mode == 1) {
        rc = side_forward(target, T + 1, f(wl.vnext), nullptr);
        if (rc == -2) rc = launch_forward_rows<SC>(P, amc, target, bc, TB + B, f(wl.vnext), st);
        if (rc != 0) return rc;
    } else if (mode != 2) {  // target-critic values of all T+1 observations (model.py:190-193); PPO reuses the returns across epochs
        if constexpr (IsGru<SC>::value) {  // recurrent critics: the critics' own pass rides in the same launch (each fills half the chip)
            rc = gru_forward_rows_pair<SC>(P, amc, critic, target, bc, T, T + 1, f(wl.v), f(wl.vnext), st, rec_c);
            if (rc != 0) return rc;
            v_done = true;
        }
        if (!v_done) {
            rc = launch_forward_rows<SC>(P, amc, target, bc, TB + B, f(wl.vnext), st);
            if (rc != 0) return rc;
        }
    }
    if (std_on && mode == 0) {  // A2C with standardise_returns: raw returns -> statistics update -> A2C on the stored returns
        a.mode = 4;
        hipLaunchKernelGGL(ac_elem_kernel, dim3((TB + 255) / 256), dim3(256), 0, st, a, *bt, w);
        if (launch_stats_update(w.st, w.rpartial, (TB + 255) / 256, P, TB, st) != 0) return -1;
        a.mode = 3;
    }
    if (mode != 1 && !v_done) {  // (forked before the actors' pass is queued, or there is nothing to overlap with)
        rc = side_forward(critic, T, f(wl.v), rec_c);
        if (rc == -2) rc = launch_forward_rows<SC>(P, amc, critic, bc, TB, f(wl.v), st, rec_c);
        if (rc != 0) return rc;
    }
    if (c->actor_forward_kept != 0) {
        // the rollout's collector left the logits and the hidden-layer record of exactly these rows (marlhip_*_ac_collect_keep on this
        // workspace, the same actor parameters): A2C's one update per rollout needs no second pass
        // (PPO: marlhip_ppo_prepare's old log-probs, and the FIRST epoch's step - the parameters have not moved yet)
        MARL_REQUIRE((mode == 1 || rec_a != nullptr) && mlp_stored_shape<SA>() && bt->batch % 16 == 0,
                     "ac_loss_grad: actor_forward_kept goes with fused feed-forward actors and whole blocks of 16 envs");
    } else {
        rc = launch_forward_rows<SA>(P, am, actor, bt, TB, f(wl.logits), st, rec_a);
        if (rc != 0) return rc;
    }
    side.do_join();  // before the elementwise stage
    hipLaunchKernelGGL(ac_elem_kernel, dim3((TB + 255) / 256), dim3(256), 0, st, a, *bt, w);
    MARL_CHECK_LAUNCH("ac_elem_kernel");
    if (mode == 1) {
        if (std_on && launch_stats_update(w.st, w.rpartial, (TB + 255) / 256, P, TB, st) != 0) return -1;
        timing_end(TIMER_LOSSGRAD, st);
        return 0;
    }
    float* scratch = f(wl.scratch);
    if (c->defer_critic_backward != 0) {
        // The critics' backward pass leaves the call's stream (marlhip_ac_config.defer_critic_backward): without a joint gradient clip the
        // optimiser step is elementwise, so the ACTORS' step - all the next rollout needs - does not wait for the critics' gradient.  The
        // actors' pass goes first (the two passes share the backward workspace), the critics' is enqueued on side_stream behind it and is
        // NOT joined: the caller steps the critics there and orders the next call on this workspace behind that (an event of its own).
        if constexpr (IsGru<SA>::value || IsGru<SC>::value) {
            set_error("ac_loss_grad: defer_critic_backward is for feed-forward networks (the recurrent passes already share the chip)");
            return -1;
        } else {
            MARL_REQUIRE(mode == 0 && c->side_stream != nullptr && c->side_stream != (void*)st,
                         "ac_loss_grad: defer_critic_backward goes with marlhip_a2c_loss_grad and a side_stream other than the call's");
            const int rc_a = launch_backward_rows<SA>(P, am, actor, bt, w.dlogits, w.lrow_a, base + wl.bwd, ws_bytes - wl.bwd, actor_grad, scratch, st, rec_a);
            if (rc_a != 0) return rc_a;
            hipEvent_t ev;
            MARL_REQUIRE(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "ac_loss_grad: hipEventCreate failed");
            const bool ok = hipEventRecord(ev, st) == hipSuccess && hipStreamWaitEvent((hipStream_t)c->side_stream, ev, 0) == hipSuccess;
            (void)hipEventDestroy(ev);  // (released once it has completed)
            MARL_REQUIRE(ok, "ac_loss_grad: could not order the critics' backward pass behind the actors'");
            rc = launch_backward_rows<SC>(P, amc, critic, bc, w.dv, w.lrow_v, base + wl.bwd, ws_bytes - wl.bwd, critic_grad, scratch + 2,
                                          (hipStream_t)c->side_stream, rec_c);
            if (rc != 0) return rc;
        }
    } else {
    hipStream_t st_c = st;
    if (wl.bwd_c != wl.bwd && side.ok()) {  // fork: the critics' backward on the side stream, behind everything queued so far
        side.do_fork();
        st_c = side.s;
    }
    rc = launch_backward_rows<SC>(P, amc, critic, bc, w.dv, w.lrow_v, base + wl.bwd_c, ws_bytes - wl.bwd_c, critic_grad, scratch + 2, st_c, rec_c);
    const int rc_a = launch_backward_rows<SA>(P, am, actor, bt, w.dlogits, w.lrow_a, base + wl.bwd, wl.bwd_c != wl.bwd ? wl.bwd_c - wl.bwd : ws_bytes - wl.bwd,
                                              actor_grad, scratch, st, rec_a);
    side.do_join();
    if (rc != 0 || rc_a != 0) return rc != 0 ? rc : rc_a;
    }
    hipLaunchKernelGGL(ac_metrics_kernel, dim3(1), dim3(1024), 0, st, (TB + 255) / 256, c->value_loss_coef, (const float*)w.partial,
                       metrics);
    timing_end(TIMER_LOSSGRAD, st);
    MARL_CHECK_LAUNCH("ac_metrics_kernel");
    return 0;
}

// DC = the critics' input width: D (independent critics) or P * D (critic.centralised: every critic reads the whole row)
template <int D, int H, int A, int DC>
int ac_step(int P, const AgentMap& am, const float* actor, const float* critic, const float* target, const marlhip_batch* bt, const marlhip_ac_config* c,
            int mode, void* ws, int64_t ws_bytes, float* actor_grad, float* critic_grad, float* metrics, hipStream_t st) {
    return ac_step_t<MlpShape<D, H, A>, MlpShape<DC, H, 1>>(P, am, actor, critic, target, bt, c, mode, ws, ws_bytes, actor_grad, critic_grad, metrics, st);
}

}  // namespace marl

