// Host-side driver of the learner half of a round: n x (replay sample -> loss/grad -> clip+Adam+target)
// enqueued from ONE C call.  Same kernels and arithmetic as the individual entry points; this only
// removes the per-launch cost of crossing the Python/ctypes boundary three times per update (the
// update is ~150 us of GPU work, so ~20 us of host time per call would otherwise starve the queue).
// The update / target-update bookkeeping is QNetwork.update + update_target (marlbase/dqn/model.py:165-185).
#include "common.h"

namespace marl {
int idqn_update_n_fused(const marlhip_idqn_learner* L, int32_t n_updates, int32_t length, uint64_t seed, uint32_t counter0,
                        int64_t* adam_step, int64_t* updates, int64_t* last_target_update, void* stream, bool* handled,
                        marlhip_exchange_fn exchange, void* exchange_ctx, int32_t world);
}

static int update_n(const marlhip_idqn_learner* L, int32_t n_updates, int32_t length, uint64_t seed, uint32_t counter0,
                    int64_t* adam_step, int64_t* updates, int64_t* last_target_update, marlhip_exchange_fn exchange, void* exchange_ctx,
                    int32_t world, void* stream);

extern "C" int marlhip_idqn_update_n(const marlhip_idqn_learner* L, int32_t n_updates, int32_t length, uint64_t seed,
                                     uint32_t counter0, int64_t* adam_step, int64_t* updates, int64_t* last_target_update,
                                     void* stream) {
    return update_n(L, n_updates, length, seed, counter0, adam_step, updates, last_target_update, nullptr, nullptr, 1, stream);
}

// The data-parallel form (nothing like it in the reference, which is one process): per update ONE exchange of the flat
// gradient between the reduce launch and the clip + Adam launch.  `exchange(ctx, grad, count, stream)` must leave the SUM over all
// ranks in `grad`, ordered after the work already enqueued on `stream` and before what is enqueued next (an ncclAllReduce on that
// stream; torch.distributed.all_reduce issued under the same current stream); clip + Adam then run with grad_scale = 1 / world on
// the identical reduced gradient on every rank, the clip norm being that of the reduced gradient (dqn/model.py:170 on the global
// batch).  The n-updates loop stays in this call: the exchange is the only host hop of an update.
extern "C" int marlhip_idqn_update_n_dist(const marlhip_idqn_learner* L, int32_t n_updates, int32_t length, uint64_t seed,
                                          uint32_t counter0, int64_t* adam_step, int64_t* updates, int64_t* last_target_update,
                                          marlhip_exchange_fn exchange, void* exchange_ctx, int32_t world, void* stream) {
    MARL_REQUIRE(exchange != nullptr && world >= 1, "idqn_update_n_dist: exchange callback is NULL or world < 1");
    return update_n(L, n_updates, length, seed, counter0, adam_step, updates, last_target_update, exchange, exchange_ctx, world, stream);
}

static int update_n(const marlhip_idqn_learner* L, int32_t n_updates, int32_t length, uint64_t seed, uint32_t counter0,
                    int64_t* adam_step, int64_t* updates, int64_t* last_target_update, marlhip_exchange_fn exchange, void* exchange_ctx,
                    int32_t world, void* stream) {
    MARL_REQUIRE(L && adam_step && updates && last_target_update, "idqn_update_n: NULL pointer");
    MARL_REQUIRE(n_updates >= 0 && L->batch > 0, "idqn_update_n: bad counts");
    const int np = marlhip_net_nparams(&L->net);
    if (np < 0) return -1;
    marlhip_batch bt = {};
    bt.obss = L->obss; bt.actions = L->actions; bt.rewards = L->rewards; bt.dones = L->dones; bt.filled = L->filled;
    bt.max_len = L->rs.max_len; bt.batch = L->batch;
    const double tui = L->target_update_interval_or_tau;
    {   // hidden-64 IDQN / VDN learners: 3 launches per update (the Adam launch writes the next update's weight packs)
        bool handled = false;
        const int rc = marl::idqn_update_n_fused(L, n_updates, length, seed, counter0, adam_step, updates, last_target_update, stream, &handled,
                                                 exchange, exchange_ctx, world);
        if (rc < 0 || handled) return rc;
    }
    for (int u = 0; u < n_updates; ++u) {
        int rc;
        if (L->materialise_batch) {
            rc = marlhip_replay_sample(&L->rs, &L->rb, nullptr, L->batch, length, seed, counter0 + (uint32_t)u, L->idx, L->obss,
                                       L->actions, L->rewards, L->dones, L->filled, stream);
            if (rc < 0) return rc;
            rc = marlhip_dqn_loss_grad(&L->net, L->params, L->target, &bt, L->gamma, L->double_q, L->mode, L->workspace,
                                       L->workspace_bytes, L->grad, L->loss, stream);
        } else {  // gather inside the loss/grad kernel: no sample launch, no Batch round trip
            rc = marlhip_dqn_loss_grad_replay(&L->net, L->params, L->target, &L->rs, &L->rb, nullptr, L->batch, length, seed,
                                              counter0 + (uint32_t)u, L->idx, L->gamma, L->double_q, L->mode, L->workspace,
                                              L->workspace_bytes, L->grad, L->loss, stream);
        }
        if (rc < 0) return rc;
        const int64_t n_all = (int64_t)(L->net.n_networks > 0 ? L->net.n_networks : L->net.n_agents) * np;
        if (exchange != nullptr) {
            marl::timing_begin(marl::TIMER_EXCHANGE, (hipStream_t)stream);
            rc = exchange(exchange_ctx, L->grad, n_all, stream);
            marl::timing_end(marl::TIMER_EXCHANGE, (hipStream_t)stream);
            MARL_REQUIRE(rc == 0, "idqn_update_n_dist: the gradient exchange callback failed (%d)", rc);
        }
        *updates += 1;
        *adam_step += 1;
        const bool hard = tui > 1.0 && (double)(*updates - *last_target_update) >= tui;
        const float tau = tui < 1.0 ? (float)tui : 0.f;
        rc = marlhip_dqn_clip_adam(n_all, L->params, L->grad, L->exp_avg, L->exp_avg_sq, L->target,
                                   *adam_step, L->lr, L->beta1, L->beta2, L->eps, L->max_norm, exchange != nullptr ? 1.0f / (float)world : 1.0f,
                                   hard ? 1 : 0, tau, L->scratch, L->gnorm, stream);
        if (rc < 0) return rc;
        if (hard) *last_target_update = *updates;
    }
    return 0;
}

// QMIX: the same loop around marlhip_qmix_loss_grad_replay and the two optimiser steps QmixUpdater.apply issues (codebase_amd/hip.py) -
// the per-update host loop of VectorisedIDQN moved behind one call, with the same exchange hook as marlhip_idqn_update_n_dist.
extern "C" int marlhip_qmix_update_n(const marlhip_qmix_learner* Q, int32_t n_updates, int32_t length, uint64_t seed, uint32_t counter0,
                                     int64_t* adam_step, int64_t* updates, int64_t* last_target_update, marlhip_exchange_fn exchange,
                                     void* exchange_ctx, int32_t world, void* stream) {
    MARL_REQUIRE(Q && adam_step && updates && last_target_update, "qmix_update_n: NULL pointer");
    const marlhip_idqn_learner* L = &Q->base;
    MARL_REQUIRE(n_updates >= 0 && L->batch > 0 && world >= 1, "qmix_update_n: bad counts");
    MARL_REQUIRE(Q->mixer_rw == Q->mixer.mixer && Q->target_mixer_rw == Q->mixer.target_mixer, "qmix_update_n: mixer_rw / target_mixer_rw must alias mixer.mixer / mixer.target_mixer");
    const int np = marlhip_net_nparams(&L->net);
    if (np < 0) return -1;
    const int nm = marlhip_qmix_nparams(&L->net, Q->mixer.embed_dim, Q->mixer.hypernet_layers, Q->mixer.hypernet_embed);
    if (nm < 0) return -1;
    const int64_t n_all = (int64_t)(L->net.n_networks > 0 ? L->net.n_networks : L->net.n_agents) * np;
    MARL_REQUIRE(exchange == nullptr || Q->mixer.mixer_grad == L->grad + n_all, "qmix_update_n: the exchange needs [critic | mixer] gradients in one allocation");
    const double tui = L->target_update_interval_or_tau;
    const float scale = exchange != nullptr ? 1.0f / (float)world : 1.0f;
    for (int u = 0; u < n_updates; ++u) {
        int rc = marlhip_qmix_loss_grad_replay(&L->net, L->params, L->target, &Q->mixer, &L->rs, &L->rb, nullptr, L->batch, length, seed,
                                               counter0 + (uint32_t)u, L->idx, L->gamma, L->double_q, L->workspace, L->workspace_bytes, L->grad,
                                               L->loss, stream);
        if (rc < 0) return rc;
        if (exchange != nullptr) {
            marl::timing_begin(marl::TIMER_EXCHANGE, (hipStream_t)stream);
            rc = exchange(exchange_ctx, L->grad, n_all + nm, stream);
            marl::timing_end(marl::TIMER_EXCHANGE, (hipStream_t)stream);
            MARL_REQUIRE(rc == 0, "qmix_update_n: the gradient exchange callback failed (%d)", rc);
        }
        *updates += 1;
        *adam_step += 1;
        const bool hard = tui > 1.0 && (double)(*updates - *last_target_update) >= tui;
        const float tau = tui < 1.0 ? (float)tui : 0.f;
        if (Q->optimizer == 0) {
            rc = marlhip_dqn_clip_adam(n_all, L->params, L->grad, L->exp_avg, L->exp_avg_sq, L->target, *adam_step, L->lr, L->beta1, L->beta2, L->eps,
                                       L->max_norm, scale, hard ? 1 : 0, tau, L->scratch, L->gnorm, stream);
            if (rc < 0) return rc;
            rc = marlhip_dqn_clip_adam(nm, Q->mixer_rw, Q->mixer.mixer_grad, Q->mixer_exp_avg, Q->mixer_exp_avg_sq, Q->target_mixer_rw, *adam_step, L->lr,
                                       L->beta1, L->beta2, L->eps, 0.f, scale, hard ? 1 : 0, tau, Q->mixer_scratch, nullptr, stream);
        } else {
            rc = marlhip_dqn_clip_step(Q->optimizer, n_all, L->params, L->grad, L->exp_avg, L->exp_avg_sq, L->target, *adam_step, L->lr, L->max_norm, scale,
                                       hard ? 1 : 0, tau, L->scratch, L->gnorm, stream);
            if (rc < 0) return rc;
            rc = marlhip_dqn_clip_step(Q->optimizer, nm, Q->mixer_rw, Q->mixer.mixer_grad, Q->mixer_exp_avg, Q->mixer_exp_avg_sq, Q->target_mixer_rw, *adam_step,
                                       L->lr, 0.f, scale, hard ? 1 : 0, tau, Q->mixer_scratch, nullptr, stream);
        }
        if (rc < 0) return rc;
        if (hard) *last_target_update = *updates;
    }
    return 0;
}
