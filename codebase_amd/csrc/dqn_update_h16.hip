// extern "C" entry points of the OPT-IN split-fp16 learner (dqn_update_h16.h): marlhip_dqn_loss_grad_split16 (one loss / gradient
// from a Batch) and marlhip_idqn_update_n_split16 (n updates from the replay, the bench path's shape).  IDQN, hidden 64,
// observation width <= 32 (the compiled shapes: LBF 12 .. 27 and their ObserveID forms 14 .. 31).
#include "dqn_update_h16.h"

using namespace marl;

#define MARL_H16_SHAPES(X) X(12, 64, 6) X(15, 64, 6) X(18, 64, 6) X(21, 64, 6) X(24, 64, 6) X(27, 64, 6) X(14, 64, 6) X(17, 64, 6) X(25, 64, 6) X(31, 64, 6)

namespace marl {
int64_t h16_pack_floats(int D, int H) {  // floats per agent of the split-fp16 packs, 0 when the shape has no such kernel
#define X(d, h, a) if (D == d && H == h) return H16Pack<MlpShape<d, h, a>>::TOTAL;
    MARL_H16_SHAPES(X)
#undef X
    return 0;
}
}  // namespace marl

static AdamArgs h16_adam_args(int64_t step, double lr, double beta1, double beta2, double eps, float max_norm, int hard, float tau) {
    AdamArgs a;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.lr_step = (float)(lr / bc1); a.bc2_sqrt = (float)sqrt(bc2); a.w1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.w2 = (float)(1.0 - beta2);
    a.eps = (float)eps; a.max_norm = max_norm; a.grad_scale = 1.f; a.tau = tau; a.hard_update = hard;
    return a;
}

// one loss / gradient; fuse != nullptr: followed by reduce + clip-norm partials and clip + Adam + target (marlhip_idqn_update_n_split16)
template <class S>
static int h16_step(const marlhip_net_shape* s, const float* params, const float* tparams, const marlhip_batch* bt, const ReplaySrc* rsrc,
                    float gamma, int double_q, void* ws, int64_t ws_bytes, float* grad, float* loss, hipStream_t st, UpdFuse* fuse) {
    UpdPlan pl;
    int rec = 0;
    int rc;
    if (rsrc != nullptr) rc = launch_lossgrad_h16<S, true>(s, params, tparams, bt, *rsrc, gamma, double_q, ws, ws_bytes, st, &pl, &rec);
    else {
        ReplaySrc none = {};
        rc = launch_lossgrad_h16<S, false>(s, params, tparams, bt, none, gamma, double_q, ws, ws_bytes, st, &pl, &rec);
    }
    if (rc != 0) return rc;
    const AgentMap am = agent_map(s);
    const int P = s->n_agents, n = am.nblk * S::NPARAM, nsq = (n + 63) / 64;
    if (fuse == nullptr) {
        hipLaunchKernelGGL(dqn_reduce_kernel, dim3(nsq), dim3(256), 0, st, (const float*)ws, P, pl.nwg, S::NPARAM, am, grad, loss);
        MARL_CHECK_LAUNCH("dqn_reduce_kernel");
        return 0;
    }
    hipLaunchKernelGGL(dqn_reduce_sq_kernel, dim3(nsq), dim3(256), 0, st, (const float*)ws, P, pl.nwg, S::NPARAM, am, grad, loss, fuse->sumsq);
    MARL_CHECK_LAUNCH("dqn_reduce_sq_kernel");
    hipLaunchKernelGGL(adam_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (int64_t)n, nsq, fuse->params_rw, (const float*)grad, fuse->exp_avg,
                       fuse->exp_avg_sq, fuse->target_rw, fuse->adam, (const float*)fuse->sumsq, fuse->gnorm);
    MARL_CHECK_LAUNCH("adam_kernel");
    return 0;
}

static int h16_dispatch(const marlhip_net_shape* s, const float* params, const float* tparams, const marlhip_batch* bt, const ReplaySrc* rsrc,
                        float gamma, int double_q, void* ws, int64_t ws_bytes, float* grad, float* loss, void* stream, UpdFuse* fuse) {
    MARL_REQUIRE(s->n_hidden == 0 || s->n_hidden == 2, "split16 learner: two hidden layers");
    MARL_REQUIRE(bt->action_mask == nullptr, "split16 learner: no action masks (opt-in experiment; use the exact-f32 entry points)");
    if (agent_map_validate(s) != 0) return -1;
#define X(d, h, a)                                                                                                             \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a)                                                                \
        return h16_step<MlpShape<d, h, a>>(s, params, tparams, bt, rsrc, gamma, double_q, ws, ws_bytes, grad, loss, (hipStream_t)stream, fuse);
    MARL_H16_SHAPES(X)
#undef X
    set_error("split16 learner: no kernel for net shape D=%d H=%d A=%d (hidden 64, observation width <= 32)", s->obs_dim, s->hidden, s->n_actions);
    return -1;
}

extern "C" int marlhip_dqn_loss_grad_split16(const marlhip_net_shape* s, const float* params, const float* target_params,
                                             const marlhip_batch* batch, float gamma, int32_t double_q, void* workspace,
                                             int64_t workspace_bytes, float* grad, float* loss, void* stream) {
    MARL_REQUIRE(s && params && target_params && batch && workspace && grad && loss, "dqn_loss_grad_split16: NULL pointer");
    MARL_REQUIRE(batch->obss && batch->actions && batch->rewards && batch->dones && batch->filled, "dqn_loss_grad_split16: NULL batch field");
    MARL_REQUIRE(batch->max_len > 0 && batch->batch > 0, "dqn_loss_grad_split16: empty batch");
    MARL_REQUIRE(batch->act_agent_stride == 0 && batch->act_row_stride == 0, "dqn_loss_grad_split16: the dqn/train.py Batch layout only");
    return h16_dispatch(s, params, target_params, batch, nullptr, gamma, double_q, workspace, workspace_bytes, grad, loss, stream, nullptr);
}

extern "C" int marlhip_idqn_update_n_split16(const marlhip_idqn_learner* L, int32_t n_updates, int32_t length, uint64_t seed, uint32_t counter0,
                                             int64_t* adam_step, int64_t* updates, int64_t* last_target_update, void* stream) {
    MARL_REQUIRE(L && adam_step && updates && last_target_update, "idqn_update_n_split16: NULL pointer");
    MARL_REQUIRE(n_updates >= 0 && L->batch > 0 && L->mode == 0 && !L->materialise_batch, "idqn_update_n_split16: IDQN (mode 0), in-kernel replay gather");
    MARL_REQUIRE(length > 0 && length <= L->rs.capacity, "idqn_update_n_split16: length %d out of range", length);
    marlhip_batch bt = {};
    bt.max_len = L->rs.max_len; bt.batch = L->batch;
    const double tui = L->target_update_interval_or_tau;
    UpdFuse fuse = {};
    fuse.params_rw = L->params; fuse.target_rw = L->target; fuse.exp_avg = L->exp_avg; fuse.exp_avg_sq = L->exp_avg_sq; fuse.gnorm = L->gnorm;
    fuse.sumsq = L->scratch;
    for (int u = 0; u < n_updates; ++u) {
        const bool hard = tui > 1.0 && (double)(*updates + 1 - *last_target_update) >= tui;
        fuse.adam = h16_adam_args(*adam_step + 1, L->lr, L->beta1, L->beta2, L->eps, L->max_norm, hard ? 1 : 0, tui < 1.0 ? (float)tui : 0.f);
        ReplaySrc src;
        src.rb = L->rb; src.idx = nullptr; src.idx_out = L->idx; src.seed = seed; src.counter = counter0 + (uint32_t)u; src.length = length;
        src.capacity = L->rs.capacity;
        const int rc = h16_dispatch(&L->net, L->params, L->target, &bt, &src, L->gamma, L->double_q, L->workspace, L->workspace_bytes, L->grad,
                                    L->loss, stream, &fuse);
        if (rc < 0) return rc;
        *updates += 1;
        *adam_step += 1;
        if (hard) *last_target_update = *updates;
    }
    return 0;
}
