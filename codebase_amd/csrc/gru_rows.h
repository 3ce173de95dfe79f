// Row-level launchers of the recurrent kernels for the actor-critic learner step (a2c_core.h): the same two calls the
// feed-forward networks offer - forward of every row (here: of every sequence, step by step) and backward with an external
// output gradient - so that A2CNetwork.update / PPONetwork.update run unchanged around them.
#pragma once
#include <type_traits>

#include "collect_common.h"
#include "gru_bwd.h"

namespace marl {

template <class S>
struct IsGru : std::false_type {};
template <int D, int H, int A>
struct IsGru<GruShape<D, H, A>> : std::true_type {};

struct GruRowsWs {
    int64_t q, rec, rec2, partials, packF, packB, total;
    int nwg;
};

// own_rec: the backward also runs the recording forward; false when the caller's forward already wrote the record elsewhere
template <class S>
GruRowsWs gru_rows_ws(int P, int steps, int B, bool own_rec = true) {
    const int64_t nblk = (B + 15) / 16;
    GruRowsWs w;
    int64_t off = 0;
    auto take = [&](int64_t floats) { const int64_t o = off; off += (floats * 4 + 255) / 256 * 256; return o; };
    w.q = take((int64_t)P * steps * B * S::A);
    w.rec = take(own_rec ? (int64_t)P * steps * nblk * S::REC : 0);
    w.rec2 = take((int64_t)P * steps * nblk * GruBwd<S>::REC2);
    const int64_t items = (int64_t)steps * nblk;
    const int cap = 256 / P > 1 ? 256 / P : 1;
    w.nwg = (int)(items < cap ? items : cap);
    w.partials = take((int64_t)P * w.nwg * (S::NPARAM + 2));
    w.packF = take((int64_t)P * S::NFWD);
    w.packB = take((int64_t)P * GruBwd<S>::NBWD);
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

template <class S>
int64_t gru_rec_floats(int P, int steps, int B) {
    return (int64_t)P * steps * ((B + 15) / 16) * S::REC;
}

// out[p][t][b][:] for t < steps, sequences from zero hidden states (`hiddens=None`, ac/model.py:191,206-207); rec (optional)
// receives the activation record gru_backward_rows would otherwise recompute
template <class S>
int gru_forward_rows(int P, const AgentMap& am, const float* params, const marlhip_batch* bt, int steps, float* out, hipStream_t st,
                     float* rec = nullptr, float* packs_buf = nullptr) {
    // packs_buf: the caller's own [P][NFWD] pack space (a pass on a side stream must not share the per-process scratch)
    float* packs = packs_buf != nullptr ? packs_buf : collect_pack_scratch((size_t)P * S::NFWD * sizeof(float), st);
    if (packs == nullptr) return -1;  // error text set by collect_pack_scratch
    gru_set_attrs<S>();
    size_t as, rs;
    gru_obs_strides(bt, S::D, &as, &rs);
    hipLaunchKernelGGL((gru_pack_kernel<S>), dim3((S::NFWD + 255) / 256, P), dim3(256), 0, st, params, am, packs);
    hipLaunchKernelGGL((gru_seq_fwd_kernel<S>), dim3((bt->batch + 63) / 64, P), dim3(256), S::LDS_FLOATS * sizeof(float), st, (const float*)packs, bt->obss,
                       as, rs, steps, bt->batch, (const float*)nullptr, (float*)nullptr, out, rec);
    MARL_CHECK_LAUNCH("gru_seq_fwd_kernel (rows)");
    return 0;
}

// Two networks of one shape over the same batch in one launch: out[p][t][b][:] for t < steps from `params` (with the record) and
// out2 for t < steps2 from `params2` (the critics and their targets, ac/model.py:190-193,206-207)
template <class S>
int gru_forward_rows_pair(int P, const AgentMap& am, const float* params, const float* params2, const marlhip_batch* bt, int steps, int steps2,
                          float* out, float* out2, hipStream_t st, float* rec) {
    float* packs = collect_pack_scratch((size_t)2 * P * S::NFWD * sizeof(float), st);
    if (packs == nullptr) return -1;  // error text set by collect_pack_scratch
    float* packs2 = packs + (size_t)P * S::NFWD;
    gru_set_attrs<S>();
    size_t as, rs;
    gru_obs_strides(bt, S::D, &as, &rs);
    hipLaunchKernelGGL((gru_pack_kernel<S>), dim3((S::NFWD + 255) / 256, P), dim3(256), 0, st, params, am, packs);
    hipLaunchKernelGGL((gru_pack_kernel<S>), dim3((S::NFWD + 255) / 256, P), dim3(256), 0, st, params2, am, packs2);
    hipLaunchKernelGGL((gru_seq_fwd2_kernel<S>), dim3((bt->batch + 63) / 64, P, 2), dim3(256), S::LDS_FLOATS * sizeof(float), st, (const float*)packs,
                       (const float*)packs2, bt->obss, as, rs, steps, steps2, bt->batch, out, out2, rec);
    MARL_CHECK_LAUNCH("gru_seq_fwd2_kernel (rows)");
    return 0;
}

// grad[P][NPARAM] = d(sum of row losses)/dparams / sum(filled) from dout[P][steps][B][A]; loss[0] = sum(lrow)/sum(filled), loss[1] = sum(filled)
template <class S>
int gru_backward_rows(int P, const AgentMap& am, const float* params, const marlhip_batch* bt, int steps, const float* dout, const float* lrow,
                      void* ws, int64_t ws_bytes, float* grad, float* loss, hipStream_t st, const float* rec_in = nullptr) {
    using Bk = GruBwd<S>;
    const int B = bt->batch;
    const GruRowsWs wl = gru_rows_ws<S>(P, steps, B, rec_in == nullptr);
    MARL_REQUIRE(ws_bytes >= wl.total, "gru_backward_rows: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)wl.total);
    char* base = static_cast<char*>(ws);
    auto f = [&](int64_t o) { return reinterpret_cast<float*>(base + o); };
    gru_set_attrs<S>();
    size_t as, rs;
    gru_obs_strides(bt, S::D, &as, &rs);
    hipLaunchKernelGGL((gru_bwd_pack_kernel<S>), dim3((Bk::NBWD + 255) / 256, P), dim3(256), 0, st, params, am, f(wl.packB));
    const dim3 gridS((B + 63) / 64, P);
    const float* rec = rec_in;
    if (rec == nullptr) {  // no record from the caller's forward: recompute it
        hipLaunchKernelGGL((gru_pack_kernel<S>), dim3((S::NFWD + 255) / 256, P), dim3(256), 0, st, params, am, f(wl.packF));
        hipLaunchKernelGGL((gru_seq_fwd_kernel<S>), gridS, dim3(256), S::LDS_FLOATS * sizeof(float), st, (const float*)f(wl.packF), bt->obss, as, rs, steps,
                           B, (const float*)nullptr, (float*)nullptr, f(wl.q), f(wl.rec));
        rec = f(wl.rec);
    }
    gru_launch_seq_bwd<S>(P, B, (const float*)f(wl.packB), steps, rec, dout, f(wl.rec2), st, /*alone=*/false);
    hipLaunchKernelGGL((gru_wgrad_kernel<S>), dim3(wl.nwg, P, gru_wgrad_roles<S>()), dim3(256), gru_wgrad_lds_bytes<S>(), st, steps, B, bt->obss, as, rs, rec,
                       (const float*)f(wl.rec2), dout, lrow, bt->filled, steps, f(wl.partials));
    MARL_CHECK_LAUNCH("gru backward rows");
    const int n = am.nblk * S::NPARAM;  // one gradient block per NETWORK: a shared network's agents are summed by the reduce
    hipLaunchKernelGGL(dqn_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, (const float*)f(wl.partials), P, wl.nwg, S::NPARAM, am, grad, loss);
    MARL_CHECK_LAUNCH("dqn_reduce_kernel");
    return 0;
}

}  // namespace marl
