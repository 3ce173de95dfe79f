// Q-networks without a fused kernel - hidden layers wider than 128 units, observation / action widths outside the compiled lists
// (FCNetwork builds any list of widths, marlbase/utils/models.py:14-48): the three layers as f32 MFMA GEMMs over all rows
// (wide_mlp.h), the TD stage of the recurrent learner between forward and backward (td_rows.h).  Slower than the fused kernels
// (activations travel through HBM, the forward is recomputed by the backward), any size, same arithmetic order per dot product
// whatever the batch.  IDQN and VDN; the collectors of such networks run through the modular entry points (marlhip_wide_forward ->
// marlhip_act_from_q / marlhip_sample_from_logits -> env step -> replay add).
#include "td_rows.h"
#include "wide_mlp.h"
#include "dqn_update_kernels.h"  // QmixCtx / QmixIo of the mixer stage (templates only: nothing is instantiated here)

using namespace marl;

static WideNet wide_net(const marlhip_net_shape* s, int n_out) { return WideNet{s->obs_dim, s->hidden, n_out, s->n_hidden > 0 ? s->n_hidden : 2}; }

static int wide_check(const marlhip_net_shape* s, int n_out) {
    MARL_REQUIRE(s != nullptr, "net shape is NULL");
    if (agent_map_validate(s, true) != 0) return -1;
    MARL_REQUIRE(s->n_agents >= 1 && s->n_agents <= 16 && s->obs_dim >= 1 && s->hidden >= 1 && s->hidden <= 1024 && n_out >= 1 && n_out <= 64,
                 "wide network: shape P=%d D=%d H=%d outputs=%d out of range", s->n_agents, s->obs_dim, s->hidden, n_out);
    return 0;
}

extern "C" int marlhip_wide_nparams(const marlhip_net_shape* s, int32_t n_out) {
    if (wide_check(s, n_out) != 0) return -1;
    return (int)wide_net(s, n_out).nparam();
}

extern "C" int64_t marlhip_wide_forward_workspace_bytes(const marlhip_net_shape* s, int32_t n_rows) {
    if (wide_check(s, 1) != 0) return -1;
    MARL_REQUIRE(n_rows > 0, "wide_forward_workspace_bytes: n_rows must be > 0");
    return wide_ws(wide_net(s, 1), s->n_agents, n_rows, false).total;
}

extern "C" int marlhip_wide_forward(const marlhip_net_shape* s, int32_t n_out, const float* params, const float* obs, int64_t agent_stride,
                                    int64_t row_stride, int32_t n_rows, float* out, void* workspace, int64_t workspace_bytes, void* stream) {
    if (wide_check(s, n_out) != 0) return -1;
    MARL_REQUIRE(params && obs && out && workspace && n_rows > 0 && row_stride > 0 && agent_stride >= 0, "wide_forward: bad argument");
    const WideNet net = wide_net(s, n_out);
    MARL_REQUIRE(workspace_bytes >= wide_ws(net, s->n_agents, n_rows, false).total, "wide_forward: workspace %lld < %lld bytes",
                 (long long)workspace_bytes, (long long)wide_ws(net, s->n_agents, n_rows, false).total);
    return wide_forward_rows(net, s->n_agents, agent_map(s), params, obs, agent_stride, row_stride, n_rows, out, workspace, (hipStream_t)stream);
}

namespace {
struct WideDqnWs {
    int64_t q, tq, dq, lrow, std, bwd, total;
};
int64_t wide_std_floats(int P, int64_t R) { return (3 * P + 4) * R + 2 * P * ((R + 255) / 256); }
WideDqnWs wide_dqn_ws(const WideNet& net, int P, int T, int B) {
    WideDqnWs w;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o = (o + bytes + 255) & ~(int64_t)255; return at; };
    const int64_t rows_all = (int64_t)(T + 1) * B;
    w.q = take(P * rows_all * net.A * 4);
    w.tq = take(P * rows_all * net.A * 4);
    w.dq = take(P * rows_all * net.A * 4);
    w.lrow = take((int64_t)T * B * 4);
    w.std = take(wide_std_floats(P, (int64_t)T * B) * 4);  // chosen / bootstrap values, dL/dchosen, r, (1 - done), filled, partial sums, returns
    w.bwd = o;  // the forward passes' activation buffers share the backward workspace (sized for all T + 1 steps)
    w.total = o + wide_ws(net, P, (int)rows_all, true).total;
    return w;
}
}  // namespace

extern "C" int64_t marlhip_wide_dqn_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch) {
    if (wide_check(s, s ? s->n_actions : 0) != 0) return -1;
    MARL_REQUIRE(max_len > 0 && batch > 0, "wide_dqn_workspace_bytes: empty batch");
    return wide_dqn_ws(wide_net(s, s->n_actions), s->n_agents, max_len, batch).total;
}

static int wide_dqn_body(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* bt, float gamma,
                         int double_q, int mode, const RetStats* rst, void* workspace, int64_t workspace_bytes, float* grad, float* loss, void* stream) {
    if (wide_check(s, s->n_actions) != 0) return -1;
    MARL_REQUIRE(bt->obss && bt->actions && bt->rewards && bt->dones && bt->filled && bt->max_len > 0 && bt->batch > 0, "wide_dqn_loss_grad: bad batch");
    MARL_REQUIRE(bt->obs_agent_stride == 0 && bt->obs_row_stride == 0 && bt->act_agent_stride == 0 && bt->act_row_stride == 0,
                 "wide_dqn_loss_grad: the dqn/train.py Batch layout only");
    const int P = s->n_agents, T = bt->max_len, B = bt->batch, A = s->n_actions, D = s->obs_dim;
    const WideNet net = wide_net(s, A);
    const WideDqnWs wl = wide_dqn_ws(net, P, T, B);
    MARL_REQUIRE(workspace_bytes >= wl.total, "wide_dqn_loss_grad: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)wl.total);
    hipStream_t st = (hipStream_t)stream;
    char* base = static_cast<char*>(workspace);
    auto f = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
    const AgentMap am = agent_map(s);
    const int rows_all = (T + 1) * B;
    const int64_t as = (int64_t)rows_all * D;
    timing_begin(TIMER_LOSSGRAD, st);
    int rc = wide_forward_rows(net, P, am, params, bt->obss, as, D, rows_all, f(wl.q), base + wl.bwd, st);
    if (rc != 0) return rc;
    rc = wide_forward_rows(net, P, am, target_params, bt->obss, as, D, rows_all, f(wl.tq), base + wl.bwd, st);
    if (rc != 0) return rc;
    (void)hipMemsetAsync(f(wl.dq), 0, (size_t)P * rows_all * A * sizeof(float), st);
    if (rst != nullptr) {
        // standardise_returns (dqn/model.py:146-158, 256-264): the stage the recurrent learner runs between its sequence passes (gru.hip) -
        // chosen / bootstrap values -> returns from the de-standardised bootstrap, statistics update, dL/dchosen -> dense rows
        const int64_t R = (int64_t)T * B;
        float* chosen = f(wl.std);
        float* tqsel = chosen + P * R;
        float* dqm = tqsel + P * R;
        float* r0 = dqm + P * R;
        float* dn = r0 + R;
        float* fl = dn + R;
        float* rbuf = fl + R;  // VDN: the standardised returns [R]
        float* partial = rbuf + R;
        const dim3 gridR((unsigned)((R + 255) / 256));
        hipLaunchKernelGGL(gru_qsel_kernel, gridR, dim3(256), 0, st, P, T, B, A, (const float*)f(wl.q), (const float*)f(wl.tq), *bt, double_q, chosen, tqsel,
                           r0, dn, fl);
        MARL_CHECK_LAUNCH("gru_qsel_kernel (wide)");
        if (mode == 1) {
            rc = launch_colstd(T, B, gamma, *rst, tqsel, P, (size_t)R, r0, dn, rbuf, st);
            if (rc != 0) return rc;
            MixBufs mix = {};
            mix.chosen = chosen; mix.tqsel = tqsel; mix.r0 = r0; mix.dn = dn; mix.fl = fl; mix.dq = dqm; mix.lrow = f(wl.lrow);
            hipLaunchKernelGGL(vdn_mix_kernel, dim3((unsigned)(gridR.x > 1024 ? 1024 : gridR.x)), dim3(256), 0, st, mix, P, T, B, gamma, (const float*)rbuf);
            for (int p = 1; p < P; ++p)  // one dL/dchosen for every agent
                (void)hipMemcpyAsync(dqm + (size_t)p * R, dqm, (size_t)R * sizeof(float), hipMemcpyDeviceToDevice, st);
        } else {
            rc = launch_std_mixer(P, (int)R, gamma, *rst, chosen, tqsel, bt->rewards, dn, fl, dqm, f(wl.lrow), partial, st);
            if (rc != 0) return rc;
        }
        hipLaunchKernelGGL(gru_expand_dq_kernel, gridR, dim3(256), 0, st, P, T, B, A, (const float*)dqm, *bt, f(wl.dq));
        MARL_CHECK_LAUNCH("gru_expand_dq_kernel (wide)");
    } else {
        hipLaunchKernelGGL(gru_td_kernel, dim3((T * B + 255) / 256), dim3(256), 0, st, P, T, B, A, (const float*)f(wl.q), (const float*)f(wl.tq), *bt, gamma,
                           double_q, mode == 1 ? 1 : 0, f(wl.dq), f(wl.lrow));
        MARL_CHECK_LAUNCH("gru_td_kernel (wide)");
    }
    // rows t < T of every agent's [T + 1][B] block: the first T * B rows
    rc = wide_backward_rows(net, P, am, params, bt->obss, as, D, T * B, bt->filled, f(wl.dq), (int64_t)rows_all * A, f(wl.lrow), base + wl.bwd, grad, loss, st);
    timing_end(TIMER_LOSSGRAD, st);
    return rc;
}

extern "C" int marlhip_wide_dqn_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* bt,
                                          float gamma, int32_t double_q, int32_t mode, void* workspace, int64_t workspace_bytes, float* grad,
                                          float* loss, void* stream) {
    MARL_REQUIRE(s && params && target_params && bt && workspace && grad && loss, "wide_dqn_loss_grad: NULL pointer");
    MARL_REQUIRE(mode == 0 || mode == 1, "wide_dqn_loss_grad: mode %d (0 = IDQN, 1 = VDN; QMIX: marlhip_wide_qmix_loss_grad)", mode);
    return wide_dqn_body(s, params, target_params, bt, gamma, double_q, mode, nullptr, workspace, workspace_bytes, grad, loss, stream);
}

extern "C" int marlhip_wide_dqn_loss_grad_std(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* bt,
                                              float gamma, int32_t double_q, const marlhip_ret_stats* stats, void* workspace,
                                              int64_t workspace_bytes, float* grad, float* loss, void* stream) {
    MARL_REQUIRE(s && params && target_params && bt && workspace && grad && loss && stats && stats->mean && stats->var && stats->count,
                 "wide_dqn_loss_grad_std: NULL pointer");
    // columns = 0: per-agent statistics, the independent learner; columns = batch: VDNetwork's per-batch-column statistics (marlhip_ret_stats)
    MARL_REQUIRE(stats->columns == 0 || stats->columns == bt->batch, "wide_dqn_loss_grad_std: statistics with %d columns for a batch of %d",
                 stats->columns, bt->batch);
    RetStats rst;
    ret_stats_fill(rst, stats);
    return wide_dqn_body(s, params, target_params, bt, gamma, double_q, stats->columns == 0 ? 0 : 1, &rst, workspace, workspace_bytes, grad, loss,
                         stream);
}

// ---- QMIX with such agent networks: GEMM forward of the online and target networks -> chosen / bootstrap values -> the mixer stage of
// qmix.h (through dqn_update.hip, as the recurrent learner uses it) -> dL/dchosen expanded to dense rows -> GEMM backward
namespace marl {
int qmix_mix_stage(const marlhip_net_shape* s, const QmixCtx* qx, const marlhip_batch* bt, const QmixIo* io, float gamma, int phase,
                   const float* loss, hipStream_t stream);
int64_t qmix_mixer_ws_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch);
int64_t qmix_mixer_ws_bytes_mx(const marlhip_net_shape* s, int embed_dim, int hypernet_layers, int hypernet_embed, int32_t max_len, int32_t batch);
void qmix_ctx_mixing(QmixCtx& qx, const marlhip_net_shape* s, const marlhip_qmix_mixer* mx);
}

static int64_t wide_qmix_extra(const marlhip_net_shape* s, int T, int B) { return ((int64_t)(3 * s->n_agents + 3) * T * B * 4 + 255) / 256 * 256; }

extern "C" int64_t marlhip_wide_qmix_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch) {
    const int64_t a = marlhip_wide_dqn_workspace_bytes(s, max_len, batch), m = a < 0 ? -1 : qmix_mixer_ws_bytes(s, max_len, batch);
    if (a < 0 || m < 0) return -1;
    return a + wide_qmix_extra(s, max_len, batch) + m;
}

extern "C" int64_t marlhip_wide_qmix_workspace_bytes_mx(const marlhip_net_shape* s, const marlhip_qmix_mixer* mx, int32_t max_len, int32_t batch) {
    MARL_REQUIRE(mx != nullptr, "wide_qmix_workspace_bytes: NULL mixer");
    const int64_t a = marlhip_wide_dqn_workspace_bytes(s, max_len, batch),
                  m = a < 0 ? -1 : qmix_mixer_ws_bytes_mx(s, mx->embed_dim, mx->hypernet_layers, mx->hypernet_embed, max_len, batch);
    if (a < 0 || m < 0) return -1;
    return a + wide_qmix_extra(s, max_len, batch) + m;
}

extern "C" int marlhip_wide_qmix_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_qmix_mixer* mx,
                                           const marlhip_batch* bt, float gamma, int32_t double_q, void* workspace, int64_t workspace_bytes,
                                           float* grad, float* loss, void* stream) {
    MARL_REQUIRE(s && params && target_params && mx && mx->mixer && mx->target_mixer && mx->mixer_grad && bt && workspace && grad && loss,
                 "wide_qmix_loss_grad: NULL pointer");
    if (wide_check(s, s->n_actions) != 0) return -1;
    MARL_REQUIRE(bt->obss && bt->actions && bt->rewards && bt->dones && bt->filled && bt->max_len > 0 && bt->batch > 0, "wide_qmix_loss_grad: bad batch");
    MARL_REQUIRE(bt->obs_agent_stride == 0 && bt->obs_row_stride == 0, "wide_qmix_loss_grad: the dqn/train.py Batch layout only");
    const int P = s->n_agents, T = bt->max_len, B = bt->batch, A = s->n_actions, D = s->obs_dim;
    const int64_t R = (int64_t)T * B;
    const WideNet net = wide_net(s, A);
    const WideDqnWs wl = wide_dqn_ws(net, P, T, B);
    const int64_t extra = wide_qmix_extra(s, T, B), mixws = qmix_mixer_ws_bytes_mx(s, mx->embed_dim, mx->hypernet_layers, mx->hypernet_embed, T, B);
    if (mixws < 0) return -1;
    MARL_REQUIRE(workspace_bytes >= wl.total + extra + mixws, "wide_qmix_loss_grad: workspace %lld < %lld bytes", (long long)workspace_bytes,
                 (long long)(wl.total + extra + mixws));
    hipStream_t st = (hipStream_t)stream;
    char* base = static_cast<char*>(workspace);
    auto f = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
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
        MARL_REQUIRE(stt->mean && stt->var && stt->count && stt->columns == B, "wide_qmix_loss_grad: return statistics need columns = batch (%d), got %d",
                     B, stt->columns);
        ret_stats_fill(rst, stt);
        qx.rst = &rst;
    }
    const AgentMap am = agent_map(s);
    const int rows_all = (T + 1) * B;
    const int64_t as = (int64_t)rows_all * D;
    const dim3 gridR((unsigned)((R + 255) / 256));
    timing_begin(TIMER_LOSSGRAD, st);
    int rc = wide_forward_rows(net, P, am, params, bt->obss, as, D, rows_all, f(wl.q), base + wl.bwd, st);
    if (rc != 0) return rc;
    rc = wide_forward_rows(net, P, am, target_params, bt->obss, as, D, rows_all, f(wl.tq), base + wl.bwd, st);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(gru_qsel_kernel, gridR, dim3(256), 0, st, P, T, B, A, (const float*)f(wl.q), (const float*)f(wl.tq), *bt, double_q, chosen, tqsel,
                       r0, dn, fl);
    MARL_CHECK_LAUNCH("gru_qsel_kernel (wide)");
    QmixIo io = {chosen, tqsel, r0, dn, fl, dqm, f(wl.lrow), nullptr};
    rc = qmix_mix_stage(s, &qx, bt, &io, gamma, 0, nullptr, st);
    if (rc != 0) return rc;
    (void)hipMemsetAsync(f(wl.dq), 0, (size_t)P * rows_all * A * sizeof(float), st);
    hipLaunchKernelGGL(gru_expand_dq_kernel, gridR, dim3(256), 0, st, P, T, B, A, (const float*)dqm, *bt, f(wl.dq));
    MARL_CHECK_LAUNCH("gru_expand_dq_kernel (wide)");
    rc = wide_backward_rows(net, P, am, params, bt->obss, as, D, T * B, bt->filled, f(wl.dq), (int64_t)rows_all * A, f(wl.lrow), base + wl.bwd, grad, loss, st);
    timing_end(TIMER_LOSSGRAD, st);
    if (rc != 0) return rc;
    return qmix_mix_stage(s, &qx, bt, &io, gamma, 1, loss, st);
}
