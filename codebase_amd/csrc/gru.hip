// extern "C" entry points of the recurrent (use_rnn) Q-network path; kernels in gru.h / gru_bwd.h
#include "gru_stack.h"
#include "collect_common.h"

using namespace marl;

// (obs dim, hidden, actions) with compiled recurrent kernels: the LBF widths and the warehouse, hidden 64 (whole network LDS-resident)
// and 128 (gate matrices streamed)
#define MARL_GRU_SHAPES(X)                                                                                   \
    X(12, 64, 6) X(15, 64, 6) X(18, 64, 6) X(21, 64, 6) X(24, 64, 6) X(27, 64, 6) X(39, 64, 6) X(71, 64, 5) \
    X(12, 128, 6) X(15, 128, 6) X(18, 128, 6) X(21, 128, 6) X(24, 128, 6) X(27, 128, 6) X(39, 128, 6) X(71, 128, 5) \
    X(14, 64, 6) X(17, 64, 6) X(25, 64, 6) X(31, 64, 6) X(47, 64, 6) X(14, 128, 6) X(17, 128, 6) X(25, 128, 6) X(31, 128, 6) X(47, 128, 6) /* env.observe_id */

static int gru_check(const marlhip_net_shape* s) {
    MARL_REQUIRE(s != nullptr, "net shape is NULL");
    if (agent_map_validate(s, true) != 0) return -1;
    MARL_REQUIRE(gru_depth(s) >= 1 && gru_depth(s) <= GRU_MAX_LAYERS, "recurrent networks: n_hidden %d = %d stacked GRU layers (1..%d: layers = [h] * 2 .. [h] * %d)",
                 s->n_hidden, gru_depth(s), GRU_MAX_LAYERS, GRU_MAX_LAYERS + 1);
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return 0;
    MARL_GRU_SHAPES(X)
#undef X
    set_error("no recurrent kernels for obs_dim %d, hidden %d, %d actions (MARL_GRU_SHAPES)", s->obs_dim, s->hidden, s->n_actions);
    return -1;
}

extern "C" int marlhip_gru_nparams(const marlhip_net_shape* s) {
    if (gru_check(s) != 0) return -1;
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return GruShape<d, h, a>::nparam(gru_depth(s));
    MARL_GRU_SHAPES(X)
#undef X
    return -1;
}

extern "C" int64_t marlhip_gru_record_floats(const marlhip_net_shape* s, int32_t steps, int32_t batch) {
    if (gru_check(s) != 0) return -1;
    return (int64_t)gru_depth(s) * s->n_agents * steps * ((batch + 15) / 16) * (s->hidden == 64 ? GruShape<15, 64, 6>::REC : GruShape<15, 128, 6>::REC);  // REC depends on H only
}

// floats of scratch behind the packs: the chain records of a stack (layers 0 .. L-2) when the caller keeps no record of its own
template <class S>
static int64_t gru_forward_chain_floats(int P, int L, int steps, int B) { return L > 1 ? (L - 1) * gru_layer_rec<S>(P, steps, B) : 0; }

template <class S>
static int gru_forward(const marlhip_net_shape* s, const float* params, const float* obs, int steps, int B, const float* h_in, float* h_out,
                       float* q_out, float* rec, hipStream_t st) {
    const int P = s->n_agents, L = gru_depth(s);
    const size_t pack_bytes = ((size_t)L * P * S::NFWD * sizeof(float) + 255) & ~(size_t)255;
    const size_t chain_bytes = rec != nullptr ? 0 : (size_t)gru_forward_chain_floats<S>(P, L, steps, B) * sizeof(float);
    float* packs = collect_pack_scratch(pack_bytes + chain_bytes, st);
    if (packs == nullptr) return -1;  // error text set by collect_pack_scratch
    float* chain = rec != nullptr ? rec : (chain_bytes ? reinterpret_cast<float*>(reinterpret_cast<char*>(packs) + pack_bytes) : nullptr);
    gru_pack_fwd_layers<S>(P, L, params, agent_map(s), packs, st);
    MARL_CHECK_LAUNCH("gru_pack_kernel");
    const size_t lds = (size_t)S::LDS_FLOATS * sizeof(float);
    static LdsAttr attr;
    if (attr.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_fwd_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr.done();
    }
    gru_fwd_layers<S>(P, L, packs, obs, (size_t)steps * B * S::D, (size_t)S::D, steps, B, h_in, h_out, q_out, chain, rec != nullptr, st);
    MARL_CHECK_LAUNCH("gru_seq_fwd_kernel");
    return 0;
}

extern "C" int64_t marlhip_gru_forward_workspace_bytes(const marlhip_net_shape* s, int32_t steps, int32_t batch) {
    if (s == nullptr || steps < 1 || batch < 1) {
        set_error("gru_forward_workspace_bytes: bad argument");
        return -1;
    }
    if (agent_map_validate(s, true) != 0) return -1;
    const int L = gru_depth(s);
    MARL_REQUIRE(L >= 1 && L <= GRU_MAX_LAYERS, "gru_forward_workspace_bytes: %d stacked GRU layers (1..%d)", L, GRU_MAX_LAYERS);
    const int64_t packs = marlhip_forward_workspace_bytes(s);  // one layer's packs (two sets, own or concatenated observations)
    if (packs < 0) return -1;
    const int64_t rec = s->hidden <= 64 ? GruShape<15, 64, 6>::REC : GruShape<15, 128, 6>::REC;  // REC depends on H only
    return L * packs + (int64_t)(L - 1) * s->n_agents * steps * ((batch + 15) / 16) * rec * 4 + 1024;
}

extern "C" int marlhip_gru_forward(const marlhip_net_shape* s, const float* params, const float* obs, int32_t steps, int32_t batch,
                                   const float* h_in, float* h_out, float* q_out, float* record, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
    if (gru_check(s) != 0) return -1;
    ScratchScope scratch(workspace, workspace_bytes);
    MARL_REQUIRE(params && obs && q_out && steps > 0 && batch > 0, "gru_forward: bad argument");
#define X(d, h, a) \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return gru_forward<GruShape<d, h, a>>(s, params, obs, steps, batch, h_in, h_out, q_out, record, (hipStream_t)stream);
    MARL_GRU_SHAPES(X)
#undef X
    return -1;
}

// ---- learner step: QNetwork._compute_loss / VDNetwork._compute_loss + backward with recurrent networks ----------------------
namespace {
struct GruWs {
    int64_t q, tq, dq, lrow, rec, recT, rec2, partials, packC, packT, packB, total;
    int nwg;
};

template <class S>
GruWs gru_ws_layout(int P, int T, int B, int L) {
    const int64_t steps = T + 1, nblk = (B + 15) / 16;
    GruWs w;
    int64_t off = 0;
    auto take = [&](int64_t floats) { const int64_t o = off; off += (floats * 4 + 255) / 256 * 256; return o; };
    w.q = take(P * steps * B * S::A);
    w.tq = take(P * steps * B * S::A);
    w.dq = take(P * steps * B * S::A);
    w.lrow = take((int64_t)T * B);
    w.rec = take(L * P * steps * nblk * S::REC);
    w.recT = take((L - 1) * P * steps * nblk * S::REC);  // the target networks' chain through a stack
    w.rec2 = take(L * P * steps * nblk * GruBwd<S>::REC2);
    const int64_t items = steps * nblk;
    const int cap = 256 / P > 1 ? 256 / P : 1;
    w.nwg = (int)(items < cap ? items : cap);
    w.partials = take((int64_t)P * w.nwg * (S::nparam(L) + 2));
    w.packC = take((int64_t)L * P * S::NFWD);
    w.packT = take((int64_t)L * P * S::NFWD);
    w.packB = take((int64_t)L * P * GruBwd<S>::NBWD);
    w.total = off;
    return w;
}

template <class S>
int gru_loss_grad(const marlhip_net_shape* s, const float* params, const float* target, const marlhip_batch* bt, float gamma, int double_q,
                  int mode, void* ws, int64_t ws_bytes, float* grad, float* loss, hipStream_t st, const RetStats* rst = nullptr) {
    using Bk = GruBwd<S>;
    const int P = s->n_agents, T = bt->max_len, B = bt->batch, steps = T + 1, L = gru_depth(s);
    const GruWs wl = gru_ws_layout<S>(P, T, B, L);
    MARL_REQUIRE(ws_bytes >= wl.total, "gru_loss_grad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)wl.total);
    char* base = static_cast<char*>(ws);
    auto f = [&](int64_t o) { return reinterpret_cast<float*>(base + o); };
    const AgentMap am = agent_map(s);
    gru_pack_fwd_layers<S>(P, L, params, am, f(wl.packC), st);
    gru_pack_fwd_layers<S>(P, L, target, am, f(wl.packT), st);
    gru_pack_bwd_layers<S>(P, L, params, am, f(wl.packB), st);
    MARL_CHECK_LAUNCH("gru pack kernels");
    const size_t ldsF = (size_t)S::LDS_FLOATS * sizeof(float), ldsB = (size_t)Bk::LDS_FLOATS * sizeof(float);
    const size_t ldsW = gru_wgrad_lds_bytes<S>();
    static LdsAttr attr;
    if (attr.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_fwd2_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsF);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_bwd_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_wgrad_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsW);
        attr.done();
    }
    const dim3 gridS((B + 63) / 64, P);
    timing_begin(TIMER_LOSSGRAD, st);
    gru_fwd2_layers<S>(P, L, f(wl.packC), f(wl.packT), bt->obss, (size_t)steps * B * S::D, (size_t)S::D, steps, steps, B, f(wl.q), f(wl.tq), f(wl.rec), f(wl.recT), st);
    MARL_CHECK_LAUNCH("gru_seq_fwd_kernel");
    (void)hipMemsetAsync(f(wl.dq), 0, (size_t)P * steps * B * S::A * sizeof(float), st);
    if (rst != nullptr) {
        // standardise_returns (dqn/model.py:146-158): chosen / bootstrap values -> the standardising mixer of the feed-forward path
        // (returns from the de-standardised bootstrap, running statistics update, dL/dchosen) -> dense rows.  The scratch arrays
        // borrow the backward record's space, which is written only afterwards.
        const int64_t R = (int64_t)T * B;
        float* chosen = f(wl.rec2);
        float* tqsel = chosen + P * R;
        float* dqm = tqsel + P * R;
        float* r0 = dqm + P * R;
        float* dn = r0 + R;
        float* fl = dn + R;
        float* partial = fl + R;
        float* rbuf = partial + 2 * P * ((R + 255) / 256);  // VDN: the standardised returns [R]
        MARL_REQUIRE((3 * P + 4) * R + 2 * P * ((R + 255) / 256) <= (int64_t)P * steps * ((B + 15) / 16) * Bk::REC2, "gru_loss_grad: scratch");
        const dim3 gridR((unsigned)((R + 255) / 256));
        hipLaunchKernelGGL(gru_qsel_kernel, gridR, dim3(256), 0, st, P, T, B, S::A, (const float*)f(wl.q), (const float*)f(wl.tq), *bt, double_q, chosen,
                           tqsel, r0, dn, fl);
        if (mode == 1) {  // VDNetwork (dqn/model.py:256-264): per-batch-column statistics on the summed bootstrap, then the VDN rows
            int rc = launch_colstd(T, B, gamma, *rst, tqsel, P, (size_t)R, r0, dn, rbuf, st);
            if (rc != 0) return rc;
            MixBufs mix = {};
            mix.chosen = chosen; mix.tqsel = tqsel; mix.r0 = r0; mix.dn = dn; mix.fl = fl; mix.dq = dqm; mix.lrow = f(wl.lrow);
            hipLaunchKernelGGL(vdn_mix_kernel, dim3((unsigned)(gridR.x > 1024 ? 1024 : gridR.x)), dim3(256), 0, st, mix, P, T, B, gamma, (const float*)rbuf);
            for (int p = 1; p < P; ++p)  // one dL/dchosen for every agent
                (void)hipMemcpyAsync(dqm + (size_t)p * R, dqm, (size_t)R * sizeof(float), hipMemcpyDeviceToDevice, st);
        } else {
            const int rc = launch_std_mixer(P, (int)R, gamma, *rst, chosen, tqsel, bt->rewards, dn, fl, dqm, f(wl.lrow), partial, st);
            if (rc != 0) return rc;
        }
        hipLaunchKernelGGL(gru_expand_dq_kernel, gridR, dim3(256), 0, st, P, T, B, S::A, (const float*)dqm, *bt, f(wl.dq));
    } else {
        hipLaunchKernelGGL(gru_td_kernel, dim3((T * B + 255) / 256), dim3(256), 0, st, P, T, B, S::A, (const float*)f(wl.q), (const float*)f(wl.tq), *bt,
                           gamma, double_q, mode == 1 ? 1 : 0, f(wl.dq), f(wl.lrow));
    }
    MARL_CHECK_LAUNCH("gru_td_kernel");
    gru_bwd_layers<S>(P, L, f(wl.packB), steps, B, f(wl.rec), f(wl.dq), f(wl.rec2), st, true);
    MARL_CHECK_LAUNCH("gru_seq_bwd_kernel");
    gru_wgrad_layers<S>(P, L, wl.nwg, steps, B, bt->obss, (size_t)steps * B * S::D, (size_t)S::D, f(wl.rec), f(wl.rec2), f(wl.dq), f(wl.lrow), bt->filled, T,
                        f(wl.partials), st);
    MARL_CHECK_LAUNCH("gru_wgrad_kernel");
    const int n = am.nblk * S::nparam(L);  // one gradient block per NETWORK: a shared network's agents are summed by the reduce
    hipLaunchKernelGGL(dqn_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, (const float*)f(wl.partials), P, wl.nwg, S::nparam(L), am, grad, loss);
    timing_end(TIMER_LOSSGRAD, st);
    MARL_CHECK_LAUNCH("dqn_reduce_kernel");
    return 0;
}
}  // namespace

extern "C" int64_t marlhip_gru_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch) {
    if (gru_check(s) != 0) return -1;
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return gru_ws_layout<GruShape<d, h, a>>(s->n_agents, max_len, batch, gru_depth(s)).total;
    MARL_GRU_SHAPES(X)
#undef X
    return -1;
}

extern "C" int marlhip_gru_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* batch,
                                     float gamma, int32_t double_q, int32_t mode, void* workspace, int64_t workspace_bytes, float* grad,
                                     float* loss, void* stream) {
    if (gru_check(s) != 0) return -1;
    MARL_REQUIRE(params && target_params && batch && workspace && grad && loss, "gru_loss_grad: NULL pointer");
    MARL_REQUIRE(mode == 0 || mode == 1, "gru_loss_grad: mode %d (0 = IDQN, 1 = VDN; the recurrent QMIX path is not built)", mode);
    MARL_REQUIRE(batch->obss && batch->actions && batch->rewards && batch->dones && batch->filled && batch->max_len > 0 && batch->batch > 0,
                 "gru_loss_grad: bad batch");
    MARL_REQUIRE(batch->obs_agent_stride == 0 && batch->obs_row_stride == 0, "gru_loss_grad: the dqn/train.py Batch layout only");
#define X(d, h, a)                                               \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a)  \
        return gru_loss_grad<GruShape<d, h, a>>(s, params, target_params, batch, gamma, double_q, mode, workspace, workspace_bytes, grad, loss, \
                                                 (hipStream_t)stream);
    MARL_GRU_SHAPES(X)
#undef X
    return -1;
}

static __global__ __launch_bounds__(256) void act_from_q_kernel(int P, int N, int A, const float* __restrict__ q, float eps, uint64_t seed,
                                                         const uint32_t* __restrict__ episode, const int32_t* __restrict__ ep_length,
                                                         int32_t* __restrict__ actions) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const uint32_t epi = episode[n], t = (uint32_t)ep_length[n];
    const bool explore = eps > u01_f32(act_noise_word(seed, (uint32_t)n, epi, t, 0));
    for (int p = 0; p < P; ++p) {
        const float* row = q + ((size_t)p * N + n) * A;
        int best = 0;
        float bv = row[0];
        for (int a = 1; a < A; ++a)
            if (row[a] > bv) { bv = row[a]; best = a; }
        const int ra = (int)bounded_nr(act_noise_word(seed, (uint32_t)n, epi, t, 1 + p), (uint32_t)A);
        actions[(size_t)p * N + n] = explore ? ra : best;
    }
}

extern "C" int marlhip_act_from_q(int32_t n_agents, int32_t n_envs, int32_t n_actions, const float* q, float epsilon, uint64_t seed,
                                  const uint32_t* episode, const int32_t* ep_length, int32_t* actions, void* stream) {
    MARL_REQUIRE(q && episode && ep_length && actions && n_agents > 0 && n_envs > 0 && n_actions > 0, "act_from_q: bad argument");
    hipLaunchKernelGGL(act_from_q_kernel, dim3((n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_agents, n_envs, n_actions, q, epsilon, seed,
                       episode, ep_length, actions);
    MARL_CHECK_LAUNCH("act_from_q_kernel");
    return 0;
}

// ---- recurrent QMIX: the agent networks here, the mixer stage of qmix.h through dqn_update.hip ------------------------------
namespace marl {
int qmix_mix_stage(const marlhip_net_shape* s, const QmixCtx* qx, const marlhip_batch* bt, const QmixIo* io, float gamma, int phase,
                   const float* loss, hipStream_t stream);
int64_t qmix_mixer_ws_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch);
int64_t qmix_mixer_ws_bytes_mx(const marlhip_net_shape* s, int embed_dim, int hypernet_layers, int hypernet_embed, int32_t max_len, int32_t batch);
void qmix_ctx_mixing(QmixCtx& qx, const marlhip_net_shape* s, const marlhip_qmix_mixer* mx);
}

namespace {
template <class S>
int gru_qmix_loss_grad(const marlhip_net_shape* s, const float* params, const float* target, const marlhip_qmix_mixer* mx, const marlhip_batch* bt,
                       float gamma, int double_q, void* ws, int64_t ws_bytes, float* grad, float* loss, hipStream_t st) {
    using Bk = GruBwd<S>;
    const int P = s->n_agents, T = bt->max_len, B = bt->batch, steps = T + 1, L = gru_depth(s);
    const int64_t R = (int64_t)T * B;
    const GruWs wl = gru_ws_layout<S>(P, T, B, L);
    const int64_t extra = ((3 * P + 3) * R * 4 + 255) / 256 * 256;  // chosen, tqsel, dqm [P][R]; r0, dn, fl [R]
    const int64_t mixws = qmix_mixer_ws_bytes_mx(s, mx->embed_dim, mx->hypernet_layers, mx->hypernet_embed, T, B);
    if (mixws < 0) return -1;
    MARL_REQUIRE(ws_bytes >= wl.total + extra + mixws, "gru_qmix_loss_grad: workspace %lld < %lld bytes", (long long)ws_bytes,
                 (long long)(wl.total + extra + mixws));
    char* base = static_cast<char*>(ws);
    auto f = [&](int64_t o) { return reinterpret_cast<float*>(base + o); };
    float* chosen = f(wl.total);
    float* tqsel = chosen + P * R;
    float* dqm = tqsel + P * R;
    float* r0 = dqm + P * R;
    float* dn = r0 + R;
    float* fl = dn + R;
    QmixCtx qx;
    qx.mixer = mx->mixer; qx.tmixer = mx->target_mixer; qx.mgrad = mx->mixer_grad;
    qx.ws = base + wl.total + extra; qx.ws_bytes = mixws;
    qx.l1_fp16 = mx->l1_fp16 != 0;
    qmix_ctx_mixing(qx, s, mx);
    RetStats rst;
    if (mx->ret_stats != nullptr) {  // standardise_returns: the mixer stage standardises the target mixer's output per batch column
        const marlhip_ret_stats* stt = mx->ret_stats;
        MARL_REQUIRE(stt->mean && stt->var && stt->count && stt->columns == B, "gru_qmix_loss_grad: return statistics need columns = batch (%d), got %d", B,
                     stt->columns);
        ret_stats_fill(rst, stt);
        qx.rst = &rst;
    }
    const AgentMap am = agent_map(s);
    gru_pack_fwd_layers<S>(P, L, params, am, f(wl.packC), st);
    gru_pack_fwd_layers<S>(P, L, target, am, f(wl.packT), st);
    gru_pack_bwd_layers<S>(P, L, params, am, f(wl.packB), st);
    MARL_CHECK_LAUNCH("gru pack kernels");
    const size_t ldsF = (size_t)S::LDS_FLOATS * sizeof(float), ldsB = (size_t)Bk::LDS_FLOATS * sizeof(float);
    const size_t ldsW = gru_wgrad_lds_bytes<S>();
    static LdsAttr attr;
    if (attr.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_fwd2_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsF);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_bwd_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_wgrad_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsW);
        attr.done();
    }
    const dim3 gridS((B + 63) / 64, P), gridR((unsigned)((R + 255) / 256));
    timing_begin(TIMER_LOSSGRAD, st);
    gru_fwd2_layers<S>(P, L, f(wl.packC), f(wl.packT), bt->obss, (size_t)steps * B * S::D, (size_t)S::D, steps, steps, B, f(wl.q), f(wl.tq), f(wl.rec), f(wl.recT), st);
    hipLaunchKernelGGL(gru_qsel_kernel, gridR, dim3(256), 0, st, P, T, B, S::A, (const float*)f(wl.q), (const float*)f(wl.tq), *bt, double_q, chosen,
                       tqsel, r0, dn, fl);
    MARL_CHECK_LAUNCH("gru forward / qsel");
    QmixIo io = {chosen, tqsel, r0, dn, fl, dqm, f(wl.lrow), nullptr};
    int rc = qmix_mix_stage(s, &qx, bt, &io, gamma, 0, nullptr, st);
    if (rc != 0) return rc;
    (void)hipMemsetAsync(f(wl.dq), 0, (size_t)P * steps * B * S::A * sizeof(float), st);
    hipLaunchKernelGGL(gru_expand_dq_kernel, gridR, dim3(256), 0, st, P, T, B, S::A, (const float*)dqm, *bt, f(wl.dq));
    gru_bwd_layers<S>(P, L, f(wl.packB), steps, B, f(wl.rec), f(wl.dq), f(wl.rec2), st, true);
    gru_wgrad_layers<S>(P, L, wl.nwg, steps, B, bt->obss, (size_t)steps * B * S::D, (size_t)S::D, f(wl.rec), f(wl.rec2), f(wl.dq), f(wl.lrow), bt->filled, T,
                        f(wl.partials), st);
    MARL_CHECK_LAUNCH("gru backward");
    const int n = am.nblk * S::nparam(L);  // one gradient block per NETWORK: a shared network's agents are summed by the reduce
    hipLaunchKernelGGL(dqn_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, (const float*)f(wl.partials), P, wl.nwg, S::nparam(L), am, grad, loss);
    timing_end(TIMER_LOSSGRAD, st);
    MARL_CHECK_LAUNCH("dqn_reduce_kernel");
    return qmix_mix_stage(s, &qx, bt, &io, gamma, 1, loss, st);
}
}  // namespace

extern "C" int64_t marlhip_gru_qmix_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch) {
    const int64_t a = marlhip_gru_workspace_bytes(s, max_len, batch), m = a < 0 ? -1 : qmix_mixer_ws_bytes(s, max_len, batch);
    if (a < 0 || m < 0) return -1;
    return a + (((int64_t)(3 * s->n_agents + 3) * max_len * batch * 4 + 255) / 256 * 256) + m;
}

extern "C" int64_t marlhip_gru_qmix_workspace_bytes_mx(const marlhip_net_shape* s, const marlhip_qmix_mixer* mx, int32_t max_len, int32_t batch) {
    MARL_REQUIRE(mx != nullptr, "gru_qmix_workspace_bytes: NULL mixer");
    const int64_t a = marlhip_gru_workspace_bytes(s, max_len, batch),
                  m = a < 0 ? -1 : qmix_mixer_ws_bytes_mx(s, mx->embed_dim, mx->hypernet_layers, mx->hypernet_embed, max_len, batch);
    if (a < 0 || m < 0) return -1;
    return a + (((int64_t)(3 * s->n_agents + 3) * max_len * batch * 4 + 255) / 256 * 256) + m;
}

extern "C" int marlhip_gru_qmix_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_qmix_mixer* mixer,
                                          const marlhip_batch* batch, float gamma, int32_t double_q, void* workspace, int64_t workspace_bytes,
                                          float* grad, float* loss, void* stream) {
    if (gru_check(s) != 0) return -1;
    MARL_REQUIRE(params && target_params && mixer && mixer->mixer && mixer->target_mixer && mixer->mixer_grad && batch && workspace && grad && loss,
                 "gru_qmix_loss_grad: NULL pointer");
    MARL_REQUIRE(batch->obs_agent_stride == 0 && batch->obs_row_stride == 0, "gru_qmix_loss_grad: the dqn/train.py Batch layout only");
#define X(d, h, a)                                               \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a)  \
        return gru_qmix_loss_grad<GruShape<d, h, a>>(s, params, target_params, mixer, batch, gamma, double_q, workspace, workspace_bytes, grad, loss, \
                                                     (hipStream_t)stream);
    MARL_GRU_SHAPES(X)
#undef X
    return -1;
}

extern "C" int marlhip_gru_loss_grad_std(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* batch,
                                         float gamma, int32_t double_q, const marlhip_ret_stats* stats, void* workspace, int64_t workspace_bytes,
                                         float* grad, float* loss, void* stream) {
    if (gru_check(s) != 0) return -1;
    MARL_REQUIRE(params && target_params && batch && workspace && grad && loss && stats && stats->mean && stats->var && stats->count,
                 "gru_loss_grad_std: NULL pointer");
    MARL_REQUIRE(batch->obs_agent_stride == 0 && batch->obs_row_stride == 0, "gru_loss_grad_std: the dqn/train.py Batch layout only");
    RetStats rst;
    ret_stats_fill(rst, stats);
    // columns = 0: per-agent statistics, the independent learner; columns = batch: VDNetwork's per-batch-column statistics (marlhip_ret_stats)
    MARL_REQUIRE(stats->columns == 0 || stats->columns == batch->batch, "gru_loss_grad_std: statistics with %d columns for a batch of %d",
                 stats->columns, batch->batch);
    const int mode = stats->columns == 0 ? 0 : 1;
#define X(d, h, a)                                               \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a)  \
        return gru_loss_grad<GruShape<d, h, a>>(s, params, target_params, batch, gamma, double_q, mode, workspace, workspace_bytes, grad, loss, \
                                                (hipStream_t)stream, &rst);
    MARL_GRU_SHAPES(X)
#undef X
    return -1;
}
