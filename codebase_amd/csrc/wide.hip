// Q-networks without a fused kernel - hidden layers wider than 128 units, observation / action widths outside the compiled lists
// (FCNetwork builds any list of widths, marlbase/utils/models.py:14-48): the three layers as f32 MFMA GEMMs over all rows
// (wide_mlp.h), the TD stage of the recurrent learner between forward and backward (td_rows.h).  Slower than the fused kernels
// (activations travel through HBM, the forward is recomputed by the backward), any size, same arithmetic order per dot product
// whatever the batch.  IDQN and VDN; the collectors of such networks run through the modular entry points (marlhip_wide_forward ->
// marlhip_act_from_q / marlhip_sample_from_logits -> env step -> replay add).
#include "td_rows.h"
#include "wide_mlp.h"

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
    int64_t q, tq, dq, lrow, bwd, total;
};
WideDqnWs wide_dqn_ws(const WideNet& net, int P, int T, int B) {
    WideDqnWs w;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o = (o + bytes + 255) & ~(int64_t)255; return at; };
    const int64_t rows_all = (int64_t)(T + 1) * B;
    w.q = take(P * rows_all * net.A * 4);
    w.tq = take(P * rows_all * net.A * 4);
    w.dq = take(P * rows_all * net.A * 4);
    w.lrow = take((int64_t)T * B * 4);
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

extern "C" int marlhip_wide_dqn_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* bt,
                                          float gamma, int32_t double_q, int32_t mode, void* workspace, int64_t workspace_bytes, float* grad,
                                          float* loss, void* stream) {
    MARL_REQUIRE(s && params && target_params && bt && workspace && grad && loss, "wide_dqn_loss_grad: NULL pointer");
    if (wide_check(s, s->n_actions) != 0) return -1;
    MARL_REQUIRE(mode == 0 || mode == 1, "wide_dqn_loss_grad: mode %d (0 = IDQN, 1 = VDN; the mixer of QMIX goes with the fused agent kernels)", mode);
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
    hipLaunchKernelGGL(gru_td_kernel, dim3((T * B + 255) / 256), dim3(256), 0, st, P, T, B, A, (const float*)f(wl.q), (const float*)f(wl.tq), *bt, gamma,
                       double_q, mode == 1 ? 1 : 0, f(wl.dq), f(wl.lrow));
    MARL_CHECK_LAUNCH("gru_td_kernel (wide)");
    // rows t < T of every agent's [T + 1][B] block: the first T * B rows
    rc = wide_backward_rows(net, P, am, params, bt->obss, as, D, T * B, bt->filled, f(wl.dq), (int64_t)rows_all * A, f(wl.lrow), base + wl.bwd, grad, loss, st);
    timing_end(TIMER_LOSSGRAD, st);
    return rc;
}
