// extern "C" entry points of the DQN-family learner step (IDQN / VDN / QMIX); kernels and launchers live in
// dqn_update_kernels.h (shared with a2c.hip).
#include "dqn_update_kernels.h"

using namespace marl;

// the kernel instantiations live in dqn_update_h64.hip / _h64_oid / _h128 / _h128_oid (dqn_update_part.h)
namespace marl {
int64_t h16_pack_floats(int D, int H);  // dqn_update_h16.hip
#define MARL_PART_DECL(name)                                                                                                      \
    int name(const marlhip_net_shape*, const float*, const float*, const marlhip_batch*, const ReplaySrc*, float, int32_t, int32_t, \
             void*, int64_t, float*, float*, hipStream_t, const QmixCtx*, const RetStats*, bool*);
MARL_PART_DECL(lossgrad_part_h64)
MARL_PART_DECL(lossgrad_part_h64_oid)
MARL_PART_DECL(lossgrad_part_h128)
MARL_PART_DECL(lossgrad_part_h128_oid)
MARL_PART_DECL(lossgrad_part_rware)
#undef MARL_PART_DECL
#define MARL_PART_MIX_DECL(name) \
    int name(const marlhip_net_shape*, const QmixCtx*, const marlhip_batch*, const QmixIo*, float, int, const float*, hipStream_t, bool*);
MARL_PART_MIX_DECL(lossgrad_part_h64_mix)
MARL_PART_MIX_DECL(lossgrad_part_h64_oid_mix)
MARL_PART_MIX_DECL(lossgrad_part_rware_mix)
#undef MARL_PART_MIX_DECL
#define MARL_PART_FUSED_DECL(name)                                                                                                 \
    int name(const marlhip_net_shape*, const float*, const float*, const marlhip_batch*, const ReplaySrc*, float, int32_t, int32_t, \
             void*, int64_t, float*, float*, hipStream_t, UpdFuse*, bool*);
MARL_PART_FUSED_DECL(lossgrad_part_h64_fused)
MARL_PART_FUSED_DECL(lossgrad_part_h64_oid_fused)
#undef MARL_PART_FUSED_DECL

// QMIX mixer stage for callers that ran the agent networks themselves (gru.hip): phase 0 = mix, phase 1 = mixer-gradient reduce
int qmix_mix_stage(const marlhip_net_shape* s, const QmixCtx* qx, const marlhip_batch* bt, const QmixIo* io, float gamma, int phase,
                   const float* loss, hipStream_t stream) {
    if (qx->generic)
        return phase == 0 ? qmix_gen_mix(*qx, qx->gen, bt, nullptr, *io, gamma, stream) : qmix_gen_reduce(*qx, qx->gen, bt->max_len, bt->batch, loss, stream);
    bool found = false;
    for (auto part : {&lossgrad_part_h64_mix, &lossgrad_part_h64_oid_mix, &lossgrad_part_rware_mix}) {
        const int rc = part(s, qx, bt, io, gamma, phase, loss, stream, &found);
        if (found) return rc;
    }
    set_error("no QMIX mixer kernel for %d agents x %d observations", s->n_agents, s->obs_dim);
    return -1;
}

// which mixer stage a (shape, mixing) pair runs on: 0 = the compiled kernels of qmix.h ({64, 2, 32} on the (agents, obs) pairs of
// MARL_QMIX_SHAPES), 1 = the generic stage of qmix_gen.h (everything else QMixer.__init__ accepts), -1 = not a QMixer.
// MARLHIP_QMIX_GENERIC=1 (diagnostics / tests) sends the compiled configurations through the generic stage too.
int qmix_kind(const marlhip_net_shape* s, int embed_dim, int hypernet_layers, int hypernet_embed, QmixGenDims* dims) {
    static const bool force = getenv("MARLHIP_QMIX_GENERIC") != nullptr;
    const QmixGenDims d = {s->n_agents, s->obs_dim, embed_dim, hypernet_embed, hypernet_layers};
    if (dims != nullptr) *dims = d;
    if (!force && embed_dim == 64 && hypernet_layers == 2 && hypernet_embed == 32) {
#define X(p, dd) if (s->n_agents == p && s->obs_dim == dd) return 0;
        MARL_QMIX_SHAPES(X)
#undef X
    }
    return qmix_gen_check(d) != 0 ? -1 : 1;
}

void qmix_ctx_mixing(QmixCtx& qx, const marlhip_net_shape* s, const marlhip_qmix_mixer* mx) {
    qx.generic = qmix_kind(s, mx->embed_dim, mx->hypernet_layers, mx->hypernet_embed, &qx.gen) == 1;
}

int64_t qmix_mixer_ws_bytes_mx(const marlhip_net_shape* s, int embed_dim, int hypernet_layers, int hypernet_embed, int32_t max_len, int32_t batch) {
    QmixGenDims d;
    const int kind = qmix_kind(s, embed_dim, hypernet_layers, hypernet_embed, &d);
    if (kind < 0) return -1;
    if (kind == 1) return qmix_gen_ws_bytes(d, max_len, batch);
#define X(p, dd) if (s->n_agents == p && s->obs_dim == dd) return qmix_ws_layout<QmixShape<p, dd>>(max_len, batch).total;
    MARL_QMIX_SHAPES(X)
#undef X
    return -1;
}

int64_t qmix_mixer_ws_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch) { return qmix_mixer_ws_bytes_mx(s, 64, 2, 32, max_len, batch); }
}  // namespace marl

extern "C" int marlhip_net_nparams(const marlhip_net_shape* s) {
    MARL_REQUIRE(s != nullptr, "net shape is NULL");
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return MlpShape<d, h, a>::NPARAM;
    MARL_NET_SHAPES(X)
#undef X
    set_error("no MFMA kernel for net shape D=%d H=%d A=%d (add it to MARL_NET_SHAPES)", s->obs_dim, s->hidden, s->n_actions);
    return -1;
}

extern "C" int64_t marlhip_dqn_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch) {
    const int np = marlhip_net_nparams(s);
    if (np < 0) return -1;
    bool tp = s->hidden > 64;
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) tp = use_tp<MlpShape<d, h, a>>();
    MARL_NET_SHAPES(X)
#undef X
    const UpdPlan pl = tp ? upd_plan_tp(s->n_agents, max_len, batch, s->obs_dim > MARL_TP_NB1S_D ? 1 : 2, MARL_TP_BWD_OCC) : upd_plan(s->n_agents, max_len, batch);
    // partial records + 16 B alignment slack + weight packs (<= 4 x nparams-padded floats per agent; see launch_lossgrad)
    int64_t pack = -1;
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) pack = 2 * MlpShape<d, h, a>::NFWD + MlpShape<d, h, a>::NBWD;
    MARL_NET_SHAPES(X)
#undef X
    {   // the opt-in split-fp16 learner (dqn_update_h16.hip) keeps its packs in the same region: reserve the larger of the two
        const int64_t h16 = h16_pack_floats(s->obs_dim, s->hidden);
        if (!tp && h16 > pack) pack = h16;
    }
    const int64_t base = ws_layout(s->n_agents, pl.nwg, np + 2, (int)pack, max_len, batch).total;
    if (!tp)  // the two-pass form's stored hidden layers (qsel pass -> bwd pass), whatever mode the caller goes on to use
        return ((base + 15) & ~(int64_t)15) + lds_h_floats(s->n_agents, max_len, batch, s->hidden) * (int64_t)sizeof(float) + 128;
    return ((base + 15) & ~(int64_t)15) + 2 * tp_h2_floats(s->n_agents, max_len, batch, s->hidden) * (int64_t)sizeof(float);  // pass F -> pass B activations (h2 | h1)
}

static int lossgrad_dispatch(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* bt,
                             const ReplaySrc* rsrc, float gamma, int32_t double_q, int32_t mode, void* workspace,
                             int64_t workspace_bytes, float* grad, float* loss, void* stream, const QmixCtx* qx = nullptr,
                             const RetStats* rst = nullptr) {
    MARL_REQUIRE(rst == nullptr || mode == 3 || mode == 1, "dqn_loss_grad: return statistics go with the IDQN (3) / VDN (1) steps");
    MARL_REQUIRE(mode == 0 || mode == 1 || (mode == 2 && qx != nullptr) || (mode == 3 && rst != nullptr),
                 "dqn_loss_grad: mode %d unknown (0 = IDQN, 1 = VDN)", mode);
    if (agent_map_validate(s) != 0) return -1;
    MARL_REQUIRE(bt->act_agent_stride == 0 && bt->act_row_stride == 0, "dqn_loss_grad: action / reward strides are an actor-critic option");
    bool found = false;
    for (auto part : {&lossgrad_part_h64, &lossgrad_part_h128, &lossgrad_part_h64_oid, &lossgrad_part_h128_oid, &lossgrad_part_rware}) {
        const int rc = part(s, params, target_params, bt, rsrc, gamma, double_q, mode, workspace, workspace_bytes, grad, loss,
                            (hipStream_t)stream, qx, rst, &found);
        if (found) return rc;
    }
    set_error("no update kernel for net shape D=%d H=%d A=%d", s->obs_dim, s->hidden, s->n_actions);
    return -1;
}

extern "C" int marlhip_dqn_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params,
                                     const marlhip_batch* batch, float gamma, int32_t double_q, int32_t mode, void* workspace,
                                     int64_t workspace_bytes, float* grad, float* loss, void* stream) {
    MARL_REQUIRE(s && params && target_params && batch && workspace && grad && loss, "dqn_loss_grad: NULL pointer");
    MARL_REQUIRE(batch->obss && batch->actions && batch->rewards && batch->dones && batch->filled, "dqn_loss_grad: NULL batch field");
    MARL_REQUIRE(batch->max_len > 0 && batch->batch > 0, "dqn_loss_grad: empty batch");
    return lossgrad_dispatch(s, params, target_params, batch, nullptr, gamma, double_q, mode, workspace, workspace_bytes, grad, loss,
                             stream);
}

extern "C" int marlhip_dqn_loss_grad_replay(const marlhip_net_shape* s, const float* params, const float* target_params,
                                            const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* idx,
                                            int32_t batch, int32_t length, uint64_t seed, uint32_t counter, int32_t* idx_out,
                                            float gamma, int32_t double_q, int32_t mode, void* workspace, int64_t workspace_bytes,
                                            float* grad, float* loss, void* stream) {
    MARL_REQUIRE(s && params && target_params && rs && rb && workspace && grad && loss, "dqn_loss_grad_replay: NULL pointer");
    MARL_REQUIRE(rb->obs && rb->act && rb->rew && rb->done && rb->filled, "dqn_loss_grad_replay: NULL replay buffer");
    MARL_REQUIRE(rs->n_agents == s->n_agents && rs->obs_dim == s->obs_dim, "dqn_loss_grad_replay: replay / net shape mismatch");
    MARL_REQUIRE(batch > 0 && rs->max_len > 0, "dqn_loss_grad_replay: empty batch");
    MARL_REQUIRE(idx != nullptr || (length > 0 && length <= rs->capacity), "dqn_loss_grad_replay: length %d out of range", length);
    marlhip_batch bt = {};
    bt.max_len = rs->max_len;
    bt.batch = batch;
    ReplaySrc src;
    src.rb = *rb; src.idx = idx; src.idx_out = idx_out; src.seed = seed; src.counter = counter; src.length = length; src.capacity = rs->capacity;
    return lossgrad_dispatch(s, params, target_params, &bt, &src, gamma, double_q, mode, workspace, workspace_bytes, grad, loss, stream);
}

// ---- IDQN with standardise_returns (QNetwork._compute_loss, marlbase/dqn/model.py:146-158) -------------------------------
static int std_stats(const marlhip_ret_stats* st, RetStats* out, int mode, int batch) {
    MARL_REQUIRE(st && st->mean && st->var && st->count, "dqn_loss_grad_std: NULL return statistics");
    MARL_REQUIRE(mode == 0 || mode == 1, "dqn_loss_grad_std: mode %d (0 = IDQN, per-agent statistics; 1 = VDN, per-batch-column statistics)", mode);
    MARL_REQUIRE(mode == 0 ? st->columns == 0 : st->columns == batch,
                 "dqn_loss_grad_std: statistics with %d columns for mode %d and a batch of %d (IDQN: columns = 0, mean / var [n_agents]; VDN: "
                 "columns = batch, mean / var [batch])", st->columns, mode, batch);
    ret_stats_fill(*out, st);
    return 0;
}

extern "C" int marlhip_dqn_loss_grad_std(const marlhip_net_shape* s, const float* params, const float* target_params,
                                         const marlhip_batch* batch, float gamma, int32_t double_q, int32_t mode,
                                         const marlhip_ret_stats* stats, void* workspace, int64_t workspace_bytes, float* grad, float* loss,
                                         void* stream) {
    MARL_REQUIRE(s && params && target_params && batch && workspace && grad && loss, "dqn_loss_grad_std: NULL pointer");
    MARL_REQUIRE(batch->obss && batch->actions && batch->rewards && batch->dones && batch->filled, "dqn_loss_grad_std: NULL batch field");
    MARL_REQUIRE(batch->max_len > 0 && batch->batch > 0, "dqn_loss_grad_std: empty batch");
    RetStats rst;
    if (std_stats(stats, &rst, mode, batch->batch) != 0) return -1;
    return lossgrad_dispatch(s, params, target_params, batch, nullptr, gamma, double_q, mode == 0 ? 3 : 1, workspace, workspace_bytes, grad, loss,
                             stream, nullptr, &rst);
}

extern "C" int marlhip_dqn_loss_grad_std_replay(const marlhip_net_shape* s, const float* params, const float* target_params,
                                                const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* idx,
                                                int32_t batch, int32_t length, uint64_t seed, uint32_t counter, int32_t* idx_out,
                                                float gamma, int32_t double_q, int32_t mode, const marlhip_ret_stats* stats, void* workspace,
                                                int64_t workspace_bytes, float* grad, float* loss, void* stream) {
    MARL_REQUIRE(s && params && target_params && rs && rb && workspace && grad && loss, "dqn_loss_grad_std_replay: NULL pointer");
    MARL_REQUIRE(rb->obs && rb->act && rb->rew && rb->done && rb->filled, "dqn_loss_grad_std_replay: NULL replay buffer");
    MARL_REQUIRE(rs->n_agents == s->n_agents && rs->obs_dim == s->obs_dim, "dqn_loss_grad_std_replay: replay / net shape mismatch");
    MARL_REQUIRE(batch > 0 && rs->max_len > 0, "dqn_loss_grad_std_replay: empty batch");
    MARL_REQUIRE(idx != nullptr || (length > 0 && length <= rs->capacity), "dqn_loss_grad_std_replay: length %d out of range", length);
    RetStats rst;
    if (std_stats(stats, &rst, mode, batch) != 0) return -1;
    marlhip_batch bt = {};
    bt.max_len = rs->max_len;
    bt.batch = batch;
    ReplaySrc src;
    src.rb = *rb; src.idx = idx; src.idx_out = idx_out; src.seed = seed; src.counter = counter; src.length = length; src.capacity = rs->capacity;
    return lossgrad_dispatch(s, params, target_params, &bt, &src, gamma, double_q, mode == 0 ? 3 : 1, workspace, workspace_bytes, grad, loss,
                             stream, nullptr, &rst);
}

// ---- QMIX (QMixNetwork, marlbase/dqn/model.py:334-443) ------------------------------------------------------
extern "C" int marlhip_qmix_nparams(const marlhip_net_shape* s, int32_t embed_dim, int32_t hypernet_layers, int32_t hypernet_embed) {
    MARL_REQUIRE(s != nullptr, "net shape is NULL");
    QmixGenDims d;
    const int kind = qmix_kind(s, embed_dim, hypernet_layers, hypernet_embed, &d);
    if (kind < 0) return -1;
    if (kind == 1) return (int)qmix_gen_nparams(d);  // (for {64, 2, 32} the same number as QmixShape::NPARAM: the same canonical order)
#define X(p, dd) if (s->n_agents == p && s->obs_dim == dd) return QmixShape<p, dd>::NPARAM;
    MARL_QMIX_SHAPES(X)
#undef X
    return -1;
}

static int64_t qmix_agent_ws(const marlhip_net_shape* s, int32_t max_len, int32_t batch) {
    const int64_t a = marlhip_dqn_workspace_bytes(s, max_len, batch);
    return a < 0 ? a : ((a + 255) & ~(int64_t)255);
}

extern "C" int64_t marlhip_qmix_workspace_bytes_mx(const marlhip_net_shape* s, const marlhip_qmix_mixer* mx, int32_t max_len, int32_t batch) {
    MARL_REQUIRE(s != nullptr && mx != nullptr, "qmix_workspace_bytes: NULL pointer");
    const int64_t a = qmix_agent_ws(s, max_len, batch), m = a < 0 ? -1 : qmix_mixer_ws_bytes_mx(s, mx->embed_dim, mx->hypernet_layers, mx->hypernet_embed, max_len, batch);
    return (a < 0 || m < 0) ? -1 : a + m;
}

extern "C" int64_t marlhip_qmix_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch) {
    MARL_REQUIRE(s != nullptr, "net shape is NULL");
    const int64_t a = qmix_agent_ws(s, max_len, batch), m = a < 0 ? -1 : qmix_mixer_ws_bytes(s, max_len, batch);
    return (a < 0 || m < 0) ? -1 : a + m;
}

static int qmix_call(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_qmix_mixer* mx,
                     const marlhip_batch* bt, const ReplaySrc* rsrc, float gamma, int32_t double_q, void* workspace,
                     int64_t workspace_bytes, float* grad, float* loss, void* stream) {
    MARL_REQUIRE(mx && mx->mixer && mx->target_mixer && mx->mixer_grad, "qmix_loss_grad: NULL mixer pointer");
    if (qmix_kind(s, mx->embed_dim, mx->hypernet_layers, mx->hypernet_embed, nullptr) < 0) return -1;
    const int64_t a = qmix_agent_ws(s, bt->max_len, bt->batch);
    MARL_REQUIRE(a >= 0 && workspace_bytes > a, "qmix_loss_grad: workspace %lld too small", (long long)workspace_bytes);
    QmixCtx qx;
    qx.mixer = mx->mixer; qx.tmixer = mx->target_mixer; qx.mgrad = mx->mixer_grad;
    qx.ws = static_cast<char*>(workspace) + a;
    qx.ws_bytes = workspace_bytes - a;
    qx.l1_fp16 = mx->l1_fp16 != 0;
    qmix_ctx_mixing(qx, s, mx);
    RetStats rst;
    if (mx->ret_stats != nullptr) {  // QMixNetwork with standardise_returns: per-batch-column statistics (model.py:415-422)
        const marlhip_ret_stats* st = mx->ret_stats;
        MARL_REQUIRE(st->mean && st->var && st->count && st->columns == bt->batch,
                     "qmix_loss_grad: return statistics need mean / var [batch] and columns = batch (%d), got columns = %d", bt->batch, st->columns);
        ret_stats_fill(rst, st);
        qx.rst = &rst;
    }
    return lossgrad_dispatch(s, params, target_params, bt, rsrc, gamma, double_q, 2, workspace, a, grad, loss, stream, &qx);
}

extern "C" int marlhip_qmix_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params,
                                      const marlhip_qmix_mixer* mixer, const marlhip_batch* batch, float gamma, int32_t double_q,
                                      void* workspace, int64_t workspace_bytes, float* grad, float* loss, void* stream) {
    MARL_REQUIRE(s && params && target_params && batch && workspace && grad && loss, "qmix_loss_grad: NULL pointer");
    MARL_REQUIRE(batch->obss && batch->actions && batch->rewards && batch->dones && batch->filled, "qmix_loss_grad: NULL batch field");
    MARL_REQUIRE(batch->max_len > 0 && batch->batch > 0, "qmix_loss_grad: empty batch");
    return qmix_call(s, params, target_params, mixer, batch, nullptr, gamma, double_q, workspace, workspace_bytes, grad, loss, stream);
}

extern "C" int marlhip_qmix_loss_grad_replay(const marlhip_net_shape* s, const float* params, const float* target_params,
                                             const marlhip_qmix_mixer* mixer, const marlhip_replay_shape* rs,
                                             const marlhip_replay_buffers* rb, const int32_t* idx, int32_t batch, int32_t length,
                                             uint64_t seed, uint32_t counter, int32_t* idx_out, float gamma, int32_t double_q,
                                             void* workspace, int64_t workspace_bytes, float* grad, float* loss, void* stream) {
    MARL_REQUIRE(s && params && target_params && rs && rb && workspace && grad && loss, "qmix_loss_grad_replay: NULL pointer");
    MARL_REQUIRE(rb->obs && rb->act && rb->rew && rb->done && rb->filled, "qmix_loss_grad_replay: NULL replay buffer");
    MARL_REQUIRE(rs->n_agents == s->n_agents && rs->obs_dim == s->obs_dim, "qmix_loss_grad_replay: replay / net shape mismatch");
    MARL_REQUIRE(batch > 0 && rs->max_len > 0, "qmix_loss_grad_replay: empty batch");
    MARL_REQUIRE(idx != nullptr || (length > 0 && length <= rs->capacity), "qmix_loss_grad_replay: length %d out of range", length);
    marlhip_batch bt = {};
    bt.max_len = rs->max_len;
    bt.batch = batch;
    ReplaySrc src;
    src.rb = *rb; src.idx = idx; src.idx_out = idx_out; src.seed = seed; src.counter = counter; src.length = length; src.capacity = rs->capacity;
    return qmix_call(s, params, target_params, mixer, &bt, &src, gamma, double_q, workspace, workspace_bytes, grad, loss, stream);
}

static AdamArgs adam_args(int64_t step, double lr, double beta1, double beta2, double eps, float max_norm, float grad_scale,
                          int32_t hard_update, float tau) {
    AdamArgs a;
    // python-float (fp64) scalars exactly as torch.optim.adam._single_tensor_adam forms them
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    a.lr_step = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.w1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.eps = (float)eps; a.max_norm = max_norm; a.grad_scale = grad_scale; a.tau = tau; a.hard_update = hard_update;
    return a;
}

// marlhip_idqn_update_n for the LDS-resident-pack learner shapes (hidden 64, IDQN / VDN): per update 3 launches instead of 4 -
// loss/grad (packs kept current by the previous update's Adam launch), reduce + clip-norm partials, clip + Adam + target + packs.
// *handled = false: not such a learner, the caller runs the generic loop.
namespace marl {
int idqn_update_n_fused(const marlhip_idqn_learner* L, int32_t n_updates, int32_t length, uint64_t seed, uint32_t counter0,
                        int64_t* adam_step, int64_t* updates, int64_t* last_target_update, void* stream, bool* handled,
                        marlhip_exchange_fn exchange, void* exchange_ctx, int32_t world) {
    *handled = false;
    if (getenv("MARLHIP_NO_FUSED_EPILOGUE") != nullptr || (L->mode != 0 && L->mode != 1) || L->net.hidden > 64) return 0;
    if (agent_map_validate(&L->net) != 0) return -1;
    marlhip_batch bt = {};
    bt.obss = L->obss; bt.actions = L->actions; bt.rewards = L->rewards; bt.dones = L->dones; bt.filled = L->filled;
    bt.max_len = L->rs.max_len; bt.batch = L->batch;
    const double tui = L->target_update_interval_or_tau;
    const int np = marlhip_net_nparams(&L->net);
    if (np < 0) return -1;
    UpdFuse fuse = {};
    fuse.packs_valid = 0;
    fuse.params_rw = L->params; fuse.target_rw = L->target; fuse.exp_avg = L->exp_avg; fuse.exp_avg_sq = L->exp_avg_sq;
    fuse.gnorm = L->gnorm;
    fuse.exchange = exchange;
    fuse.exchange_ctx = exchange_ctx;
    const float grad_scale = exchange != nullptr ? 1.0f / (float)world : 1.0f;
    // filled-aware task plans for the single-pass learner (update_plan.h); MARLHIP_NO_PLAN=1: every tile walks all T steps (diagnostics, A/B rows)
    fuse.plan_enabled = (L->mode == 0 && !L->materialise_batch && getenv("MARLHIP_NO_PLAN") == nullptr) ? 1 : 0;
    for (int u = 0; u < n_updates; ++u) {
        fuse.updates_left = n_updates - u;
        const bool hard = tui > 1.0 && (double)(*updates + 1 - *last_target_update) >= tui;
        const float tau = tui < 1.0 ? (float)tui : 0.f;
        fuse.adam = adam_args(*adam_step + 1, L->lr, L->beta1, L->beta2, L->eps, L->max_norm, grad_scale, hard ? 1 : 0, tau);
        ReplaySrc src;
        src.rb = L->rb; src.idx = nullptr; src.idx_out = L->idx; src.seed = seed; src.counter = counter0 + (uint32_t)u; src.length = length; src.capacity = L->rs.capacity;
        if (L->materialise_batch) {
            const int rc = marlhip_replay_sample(&L->rs, &L->rb, nullptr, L->batch, length, seed, counter0 + (uint32_t)u, L->idx, L->obss,
                                                 L->actions, L->rewards, L->dones, L->filled, stream);
            if (rc < 0) return rc;
        }
        bool found = false;
        int rc = 0;
        fuse.sumsq = L->scratch;  // clip-norm partials: ceil(n / 64) floats (marlhip_idqn_learner.scratch)
        for (auto part : {&lossgrad_part_h64_fused, &lossgrad_part_h64_oid_fused}) {
            rc = part(&L->net, L->params, L->target, &bt, L->materialise_batch ? nullptr : &src, L->gamma, L->double_q, L->mode,
                      L->workspace, L->workspace_bytes, L->grad, L->loss, (hipStream_t)stream, &fuse, &found);
            if (found) break;
        }
        if (!found) {
            if (u == 0) return 0;  // not a fused-kernel shape: nothing was enqueued
            set_error("idqn_update_n: fused epilogue lost its shape");
            return -1;
        }
        if (rc < 0) return rc;
        *updates += 1;
        *adam_step += 1;
        if (hard) *last_target_update = *updates;
    }
    *handled = true;
    return 0;
}
}  // namespace marl

// The filled-aware task plans marlhip_idqn_update_n builds for its single-pass learner, for inspection (tests, tools): plan_out receives
// n_updates x dims[1] int32 - per update [4 header: slots, chunk length, longest episode, stored transitions][batch episode indices,
// longest first (stable)][slots x waves tasks: tile << 16 | t0 << 8 | t1, 0 = none] - exactly what the learner kernel reads.
extern "C" int marlhip_update_plan(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, int32_t n_agents, int32_t batch, int32_t length,
                                   uint64_t seed, uint32_t counter0, int32_t n_updates, int32_t* plan_out, int64_t plan_out_ints, int32_t* idx_out,
                                   int32_t* dims_out, void* stream) {
    MARL_REQUIRE(rs && rb && dims_out, "update_plan: NULL pointer");
    MARL_REQUIRE(n_agents >= 1 && batch >= 1 && length >= 1 && length <= rs->capacity && n_updates >= 0, "update_plan: bad counts");
    const UpdPlan pl = upd_plan(n_agents, rs->max_len, batch);
    const PlanDims d = plan_dims(n_agents, rs->max_len, batch, pl.nwg, UPD_WAVES, pl.n_chunks);
    dims_out[0] = d.planned; dims_out[1] = d.stride; dims_out[2] = PLAN_HDR; dims_out[3] = d.waves; dims_out[4] = d.cap_slots;
    dims_out[5] = d.nc_static; dims_out[6] = d.ngroups; dims_out[7] = d.T;
    if (plan_out == nullptr || n_updates == 0) return 0;
    MARL_REQUIRE(d.planned, "update_plan: this (batch, max_len) is outside the planner's limits (the learner runs its static plan)");
    MARL_REQUIRE(plan_out_ints >= (int64_t)n_updates * d.stride, "update_plan: plan_out holds %lld int32, %lld needed", (long long)plan_out_ints,
                 (long long)n_updates * d.stride);
    static LdsAttr attr;
    if (attr.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&update_plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr.done();
    }
    hipLaunchKernelGGL(update_plan_kernel, dim3(n_updates), dim3(PLAN_THREADS), plan_lds_bytes(d), (hipStream_t)stream, *rb, seed, counter0, length, d,
                       plan_out, idx_out, n_updates - 1);
    MARL_CHECK_LAUNCH("update_plan_kernel");
    return 0;
}

static int clip_step(const AdamArgs& a, int64_t n, float* params, const float* grad, float* state1, float* state2, float* target_params,
                     float grad_scale, float* scratch, float* gnorm_out, hipStream_t st) {
    const int nblocks = (int)((n + 255) / 256);
    if (n <= 4096) {  // one workgroup does norm + clip + step + target: one launch, best while the block is small (measured round 3: at the
        // 11 k - 22 k parameters of the 64-64 learners the single workgroup takes 13.8 us against 4.6 + 4.3 us for the two launches below)
        hipLaunchKernelGGL(adam_fused_kernel, dim3(1), dim3(1024), 0, st, n, params, grad, state1, state2, target_params, a, gnorm_out);
        MARL_CHECK_LAUNCH("adam_fused_kernel");
    } else {
        // the norm is only formed when somebody uses it (QMIX's mixer block and unclipped actor-critic steps do not: one launch less)
        const bool need_norm = a.max_norm > 0.f || gnorm_out != nullptr;
        if (need_norm) {
            hipLaunchKernelGGL(sumsq_kernel, dim3(nblocks), dim3(256), 0, st, grad, n, grad_scale, scratch);
            MARL_CHECK_LAUNCH("sumsq_kernel");
        }
        hipLaunchKernelGGL(adam_kernel, dim3(nblocks), dim3(256), 0, st, n, need_norm ? nblocks : 0, params, grad, state1, state2, target_params, a,
                           (const float*)scratch, gnorm_out);
        MARL_CHECK_LAUNCH("adam_kernel");
    }
    return 0;
}

extern "C" int marlhip_dqn_clip_adam(int64_t n, float* params, const float* grad, float* exp_avg, float* exp_avg_sq,
                                     float* target_params, int64_t step, double lr, double beta1, double beta2, double eps,
                                     float max_norm, float grad_scale, int32_t hard_update, float tau, float* scratch,
                                     float* gnorm_out, void* stream) {
    MARL_REQUIRE(n > 0 && params && grad && exp_avg && exp_avg_sq && scratch, "dqn_clip_adam: NULL pointer");
    MARL_REQUIRE(step >= 1, "dqn_clip_adam: step must be >= 1");
    const AdamArgs a = adam_args(step, lr, beta1, beta2, eps, max_norm, grad_scale, hard_update, tau);
    return clip_step(a, n, params, grad, exp_avg, exp_avg_sq, target_params, grad_scale, scratch, gnorm_out, (hipStream_t)stream);
}

extern "C" int marlhip_dqn_clip_step(int32_t optimizer, int64_t n, float* params, const float* grad, float* state1, float* state2,
                                     float* target_params, int64_t step, double lr, float max_norm, float grad_scale, int32_t hard_update,
                                     float tau, float* scratch, float* gnorm_out, void* stream) {
    MARL_REQUIRE(n > 0 && params && grad && state1 && state2 && scratch, "dqn_clip_step: NULL pointer");
    MARL_REQUIRE(step >= 1, "dqn_clip_step: step must be >= 1");
    MARL_REQUIRE(optimizer >= 0 && optimizer <= 3, "dqn_clip_step: optimizer %d (0 Adam, 1 SGD, 2 RMSprop, 3 AdamW)", optimizer);
    // torch's defaults: Adam / AdamW betas (0.9, 0.999), eps 1e-8, AdamW weight_decay 1e-2; RMSprop alpha 0.99, eps 1e-8
    AdamArgs a = adam_args(step, lr, 0.9, 0.999, 1e-8, max_norm, grad_scale, hard_update, tau);
    a.opt = optimizer;
    a.neg_lr = (float)(-lr);
    if (optimizer == 2) {
        a.alpha = (float)0.99;
        a.w2 = (float)(1.0 - 0.99);
    }
    if (optimizer == 3) a.decay = (float)(1.0 - lr * 1e-2);
    return clip_step(a, n, params, grad, state1, state2, target_params, grad_scale, scratch, gnorm_out, (hipStream_t)stream);
}
