// Row-level launchers of the recurrent kernels for the actor-critic learner step (a2c_core.h): the same two calls the
// feed-forward networks offer - forward of every row (here: of every sequence, step by step) and backward with an external
// output gradient - so that A2CNetwork.update / PPONetwork.update run unchanged around them.
#pragma once
#include <type_traits>

#include "collect_common.h"
#include "gru_stack.h"

namespace marl {

template <class S>
struct IsGru : std::false_type {};
template <int D, int H, int A>
struct IsGru<GruShape<D, H, A>> : std::true_type {};

struct GruRowsWs {
    int64_t q, rec, rec2, partials, packF, packB, total;
    int nwg;
};

// own_rec: the backward also runs the recording forward; false when the caller's forward already wrote the record elsewhere.
// L: stacked GRU layers (gru_stack.h; AgentMap::depth) - L records, L backward records, L pack sets, one partial record of the whole block
template <class S>
GruRowsWs gru_rows_ws(int P, int steps, int B, bool own_rec = true, int L = 1) {
    GruRowsWs w;
    int64_t off = 0;
    auto take = [&](int64_t floats) { const int64_t o = off; off += (floats * 4 + 255) / 256 * 256; return o; };
    w.q = take((int64_t)P * steps * B * S::A);
    w.rec = take(own_rec ? L * gru_layer_rec<S>(P, steps, B) : 0);
    w.rec2 = take(L * gru_layer_rec2<S>(P, steps, B));
    const int64_t items = (int64_t)steps * ((B + 15) / 16);
    const int cap = 256 / P > 1 ? 256 / P : 1;
    w.nwg = (int)(items < cap ? items : cap);
    w.partials = take((int64_t)P * w.nwg * (S::nparam(L) + 2));
    w.packF = take((int64_t)L * P * S::NFWD);
    w.packB = take((int64_t)L * P * GruBwd<S>::NBWD);
    w.total = off;
    return w;
}

inline void gru_obs_strides(const marlhip_batch* bt, int D, size_t* as, size_t* rs) {
    *as = bt->obs_agent_stride > 0 ? (size_t)bt->obs_agent_stride : (bt->obs_agent_stride < 0 ? 0 : (size_t)(bt->max_len + 1) * bt->batch * D);
    *rs = bt->obs_row_stride ? (size_t)bt->obs_row_stride : (size_t)D;
}

template <class S>
void gru_set_attrs() {
    static LdsAttr done;
    if (!done.need()) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_fwd_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(S::LDS_FLOATS * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_fwd2_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(S::LDS_FLOATS * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_bwd_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(GruBwd<S>::LDS_FLOATS * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_wgrad_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)gru_wgrad_lds_bytes<S>());
    done.done();
}

// the activation records of a family's T-step passes.  A stack (L > 1) sizes them for T + 1 steps: the region also serves as the chain
// scratch of the record-less (T + 1)-step pass of marlhip_gru_ppo_prepare (gru_forward_rows: chain_buf)
template <class S>
int64_t gru_rec_floats(int P, int steps, int B, int L = 1) {
    return L * gru_layer_rec<S>(P, L > 1 ? steps + 1 : steps, B);
}

// bytes of pack scratch (collect_pack_scratch) one forward-rows launch of a stack wants: the L pack sets and, for a pass that keeps no
// record, the chain records of layers 0 .. L-2
template <class S>
int64_t gru_forward_scratch_bytes(int P, int steps, int B, int L) {
    return (((int64_t)L * P * S::NFWD * 4 + 255) & ~(int64_t)255) + (L > 1 ? (L - 1) * gru_layer_rec<S>(P, steps, B) * 4 : 0) + 256;
}

// out[p][t][b][:] for t < steps, sequences from zero hidden states (`hiddens=None`, ac/model.py:191,206-207); rec (optional)
// receives the activation record gru_backward_rows would otherwise recompute (L records for a stack of L = am.depth layers)
template <class S>
int gru_forward_rows(int P, const AgentMap& am, const float* params, const marlhip_batch* bt, int steps, float* out, hipStream_t st,
                     float* rec = nullptr, float* packs_buf = nullptr, float* chain_buf = nullptr) {
    // packs_buf: the caller's own [L][P][NFWD] pack space (a pass on a side stream must not share the per-process scratch); chain_buf:
    // with it, (L - 1) records' worth of space for a stack's pass that keeps no record
    const int L = am.depth, B = bt->batch;
    const size_t pack_bytes = ((size_t)L * P * S::NFWD * sizeof(float) + 255) & ~(size_t)255;
    const size_t chain_bytes = (rec == nullptr && L > 1 && chain_buf == nullptr) ? (size_t)(L - 1) * gru_layer_rec<S>(P, steps, B) * sizeof(float) : 0;
    MARL_REQUIRE(packs_buf == nullptr || chain_bytes == 0, "recurrent forward rows: a stack's record-less pass with its own packs needs chain space");
    float* packs = packs_buf != nullptr ? packs_buf : collect_pack_scratch(pack_bytes + chain_bytes, st);
    if (packs == nullptr) return -1;  // error text set by collect_pack_scratch
    float* chain = rec != nullptr ? rec : (chain_buf != nullptr ? chain_buf : (chain_bytes ? reinterpret_cast<float*>(reinterpret_cast<char*>(packs) + pack_bytes) : nullptr));
    gru_set_attrs<S>();
    size_t as, rs;
    gru_obs_strides(bt, S::D, &as, &rs);
    gru_pack_fwd_layers<S>(P, L, params, am, packs, st);
    gru_fwd_layers<S>(P, L, packs, bt->obss, as, rs, steps, B, nullptr, nullptr, out, chain, rec != nullptr, st);
    MARL_CHECK_LAUNCH("gru_seq_fwd_kernel (rows)");
    return 0;
}

// Two networks of one shape over the same batch in one launch: out[p][t][b][:] for t < steps from `params` (with the record) and
// out2 for t < steps2 from `params2` (the critics and their targets, ac/model.py:190-193,206-207)
template <class S>
int gru_forward_rows_pair(int P, const AgentMap& am, const float* params, const float* params2, const marlhip_batch* bt, int steps, int steps2,
                          float* out, float* out2, hipStream_t st, float* rec) {
    const int L = am.depth, B = bt->batch;
    const size_t pack_bytes = ((size_t)2 * L * P * S::NFWD * sizeof(float) + 255) & ~(size_t)255;
    const size_t chain_bytes = L > 1 ? (size_t)(L - 1) * gru_layer_rec<S>(P, steps2, B) * sizeof(float) : 0;  // the second network's chain
    float* packs = collect_pack_scratch(pack_bytes + chain_bytes, st);
    if (packs == nullptr) return -1;  // error text set by collect_pack_scratch
    float* packs2 = packs + (size_t)L * P * S::NFWD;
    float* chain2 = chain_bytes ? reinterpret_cast<float*>(reinterpret_cast<char*>(packs) + pack_bytes) : nullptr;
    MARL_REQUIRE(L == 1 || rec != nullptr, "recurrent forward rows (pair): a stack chains the first network through its record");
    gru_set_attrs<S>();
    size_t as, rs;
    gru_obs_strides(bt, S::D, &as, &rs);
    gru_pack_fwd_layers<S>(P, L, params, am, packs, st);
    gru_pack_fwd_layers<S>(P, L, params2, am, packs2, st);
    gru_fwd2_layers<S>(P, L, packs, packs2, bt->obss, as, rs, steps, steps2, B, out, out2, rec, chain2, st);
    MARL_CHECK_LAUNCH("gru_seq_fwd2_kernel (rows)");
    return 0;
}

// grad[P][NPARAM] = d(sum of row losses)/dparams / sum(filled) from dout[P][steps][B][A]; loss[0] = sum(lrow)/sum(filled), loss[1] = sum(filled)
template <class S>
int gru_backward_rows(int P, const AgentMap& am, const float* params, const marlhip_batch* bt, int steps, const float* dout, const float* lrow,
                      void* ws, int64_t ws_bytes, float* grad, float* loss, hipStream_t st, const float* rec_in = nullptr) {
    const int B = bt->batch, L = am.depth;
    const GruRowsWs wl = gru_rows_ws<S>(P, steps, B, rec_in == nullptr, L);
    MARL_REQUIRE(ws_bytes >= wl.total, "gru_backward_rows: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)wl.total);
    char* base = static_cast<char*>(ws);
    auto f = [&](int64_t o) { return reinterpret_cast<float*>(base + o); };
    gru_set_attrs<S>();
    size_t as, rs;
    gru_obs_strides(bt, S::D, &as, &rs);
    gru_pack_bwd_layers<S>(P, L, params, am, f(wl.packB), st);
    const float* rec = rec_in;
    if (rec == nullptr) {  // no record from the caller's forward: recompute it
        gru_pack_fwd_layers<S>(P, L, params, am, f(wl.packF), st);
        gru_fwd_layers<S>(P, L, f(wl.packF), bt->obss, as, rs, steps, B, nullptr, nullptr, f(wl.q), f(wl.rec), true, st);
        rec = f(wl.rec);
    }
    gru_bwd_layers<S>(P, L, f(wl.packB), steps, B, rec, dout, f(wl.rec2), st, /*alone=*/false);
    gru_wgrad_layers<S>(P, L, wl.nwg, steps, B, bt->obss, as, rs, rec, f(wl.rec2), dout, lrow, bt->filled, steps, f(wl.partials), st);
    MARL_CHECK_LAUNCH("gru backward rows");
    const int n = am.nblk * S::nparam(L);  // one gradient block per NETWORK: a shared network's agents are summed by the reduce
    hipLaunchKernelGGL(dqn_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, (const float*)f(wl.partials), P, wl.nwg, S::nparam(L), am, grad, loss);
    MARL_CHECK_LAUNCH("dqn_reduce_kernel");
    return 0;
}

}  // namespace marl
