// extern "C" entry points of the actor-critic learner step when actor.use_rnn != critic.use_rnn.  The reference builds the two families from
// their own flags (marlbase/ac/model.py:45-97: MultiAgent*Network(..., actor.use_rnn, ...) / (..., critic.use_rnn, ...)), so a recurrent actor
// next to feed-forward critics (or the reverse) is a configuration A2CNetwork / PPONetwork accept.  a2c_core.h's step is generic in the two
// shapes - forward rows / sequence forwards per family, one elementwise kernel, backward rows / back-propagation through time per family - so
// this file only instantiates ac_step_t<GruShape, MlpShape> and ac_step_t<MlpShape, GruShape> for the LBF observation widths and gives them
// entry points; a2c.hip / gru_ac.hip keep the two pure combinations.  Independent critics only (no centralised critic here).
#include "a2c_core.h"

using namespace marl;

// (obs dim, hidden, actions): the LBF widths and the warehouse at hidden 64 and 128 (gru_ac.hip's list without the observe_id widths)
#define MARL_MIXED_AC_SHAPES(X) \
    X(12, 64, 6) X(15, 64, 6) X(18, 64, 6) X(21, 64, 6) X(24, 64, 6) X(27, 64, 6) X(39, 64, 6) X(12, 128, 6) X(15, 128, 6) X(18, 128, 6) X(21, 128, 6) \
    X(24, 128, 6) X(27, 128, 6) X(39, 128, 6) X(71, 64, 5) X(71, 128, 5) /* rware */

// Depths (C-ABI 219): the recurrent family may be a stack (csrc/gru_stack.h) - recurrent ACTORS take theirs from s->n_hidden = len(actor.layers)
// (2..5), recurrent CRITICS from marlhip_ac_config.critic_n_hidden = len(critic.layers) (0 = 2 = one GRU layer); the feed-forward family
// has two hidden layers.
static int mixed_check(const marlhip_net_shape* s, int actor_rnn, int critic_n_hidden) {
    MARL_REQUIRE(s != nullptr, "net shape is NULL");
    if (agent_map_validate(s, actor_rnn != 0) != 0) return -1;
    if (actor_rnn) {
        MARL_REQUIRE(gru_depth(s) >= 1 && gru_depth(s) <= GRU_MAX_LAYERS, "mixed actor-critic: %d stacked GRU layers for the actors (1..%d)", gru_depth(s), GRU_MAX_LAYERS);
        MARL_REQUIRE(critic_n_hidden == 0 || critic_n_hidden == 2, "mixed actor-critic: feed-forward critics have two hidden layers (critic_n_hidden %d)", critic_n_hidden);
    } else {
        MARL_REQUIRE(critic_n_hidden == 0 || (critic_n_hidden >= 2 && critic_n_hidden <= GRU_MAX_LAYERS + 1),
                     "mixed actor-critic: critic_n_hidden %d (0, 2..%d: 1..%d stacked GRU layers)", critic_n_hidden, GRU_MAX_LAYERS + 1, GRU_MAX_LAYERS);
    }
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return 0;
    MARL_MIXED_AC_SHAPES(X)
#undef X
    set_error("no mixed recurrent / feed-forward actor-critic kernels for obs_dim %d, hidden %d, %d actions (MARL_MIXED_AC_SHAPES)", s->obs_dim, s->hidden,
              s->n_actions);
    return -1;
}

// actor_rnn != 0: recurrent actors (GruShape blocks: marlhip_gru_nparams) + feed-forward critics (marlhip_ac_critic_nparams);
// actor_rnn == 0: feed-forward actors (marlhip_net_nparams) + recurrent critics (marlhip_gru_ac_critic_nparams)
extern "C" int64_t marlhip_mixed_ac_workspace_bytes_lc(const marlhip_net_shape* s, int32_t actor_rnn, int32_t critic_n_hidden, int32_t max_len, int32_t batch) {
    if (mixed_check(s, actor_rnn, critic_n_hidden) != 0) return -1;
    const int La = actor_rnn ? gru_depth(s) : 1, Lc = actor_rnn ? 1 : (critic_n_hidden > 0 ? critic_n_hidden - 1 : 1);
#define X(d, h, a)                                                                                                                    \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a)                                                                       \
        return actor_rnn ? ac_ws_layout<GruShape<d, h, a>, MlpShape<d, h, 1>>(s->n_agents, max_len, batch, La, Lc).total              \
                         : ac_ws_layout<MlpShape<d, h, a>, GruShape<d, h, 1>>(s->n_agents, max_len, batch, La, Lc).total;
    MARL_MIXED_AC_SHAPES(X)
#undef X
    return -1;
}

extern "C" int64_t marlhip_mixed_ac_workspace_bytes(const marlhip_net_shape* s, int32_t actor_rnn, int32_t max_len, int32_t batch) {
    return marlhip_mixed_ac_workspace_bytes_lc(s, actor_rnn, 0, max_len, batch);
}

static int mixed_call(const marlhip_net_shape* s, int actor_rnn, const float* actor, const float* critic, const float* target, const marlhip_batch* bt,
                      const marlhip_ac_config* c, int mode, void* ws, int64_t ws_bytes, float* actor_grad, float* critic_grad, float* metrics,
                      void* stream) {
    MARL_REQUIRE(c != nullptr, "mixed_ac_loss_grad: NULL config");
    if (mixed_check(s, actor_rnn, c->critic_n_hidden) != 0) return -1;
    MARL_REQUIRE(actor && critic && bt && ws, "mixed_ac_loss_grad: NULL pointer");
    MARL_REQUIRE(mode == 1 || (actor_grad && critic_grad && metrics), "mixed_ac_loss_grad: NULL output");
    MARL_REQUIRE(mode == 2 || target != nullptr, "mixed_ac_loss_grad: NULL target critic");
    MARL_REQUIRE(bt->obss && bt->actions && bt->rewards && bt->dones && bt->filled, "mixed_ac_loss_grad: NULL batch field");
    MARL_REQUIRE(bt->max_len > 0 && bt->batch > 0, "mixed_ac_loss_grad: empty batch");
    MARL_REQUIRE(c->n_steps >= 1 && c->n_steps <= 16, "mixed_ac_loss_grad: n_steps %d (1..16)", c->n_steps);
    MARL_REQUIRE(!c->centralised_critic, "mixed_ac_loss_grad: independent critics only");
    MARL_REQUIRE(!c->actor_forward_kept && !c->defer_critic_backward, "mixed_ac_loss_grad: no kept forward pass / deferred critics with a recurrent family");
    MARL_REQUIRE((c->ret_mean == nullptr) == (c->ret_var == nullptr) && (c->ret_mean == nullptr) == (c->ret_count == nullptr),
                 "mixed_ac_loss_grad: return statistics must be given together (mean, var, count) or not at all");
#define X(d, h, a)                                                                                                                                   \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) {                                                                                    \
        if (actor_rnn)                                                                                                                               \
            return ac_step_t<GruShape<d, h, a>, MlpShape<d, h, 1>>(s->n_agents, agent_map(s), actor, critic, target, bt, c, mode, ws, ws_bytes, actor_grad, \
                                                                   critic_grad, metrics, (hipStream_t)stream);                                       \
        return ac_step_t<MlpShape<d, h, a>, GruShape<d, h, 1>>(s->n_agents, agent_map(s), actor, critic, target, bt, c, mode, ws, ws_bytes, actor_grad,    \
                                                               critic_grad, metrics, (hipStream_t)stream);                                           \
    }
    MARL_MIXED_AC_SHAPES(X)
#undef X
    return -1;
}

extern "C" int marlhip_mixed_a2c_loss_grad(const marlhip_net_shape* s, int32_t actor_rnn, const float* actor, const float* critic, const float* target_critic,
                                           const marlhip_batch* batch, const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes,
                                           float* actor_grad, float* critic_grad, float* metrics, void* stream) {
    return mixed_call(s, actor_rnn, actor, critic, target_critic, batch, cfg, 0, workspace, workspace_bytes, actor_grad, critic_grad, metrics, stream);
}

extern "C" int marlhip_mixed_ppo_prepare(const marlhip_net_shape* s, int32_t actor_rnn, const float* actor, const float* critic, const float* target_critic,
                                         const marlhip_batch* batch, const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes, void* stream) {
    return mixed_call(s, actor_rnn, actor, critic, target_critic, batch, cfg, 1, workspace, workspace_bytes, nullptr, nullptr, nullptr, stream);
}

extern "C" int marlhip_mixed_ppo_loss_grad(const marlhip_net_shape* s, int32_t actor_rnn, const float* actor, const float* critic, const marlhip_batch* batch,
                                           const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes, float* actor_grad, float* critic_grad,
                                           float* metrics, void* stream) {
    return mixed_call(s, actor_rnn, actor, critic, nullptr, batch, cfg, 2, workspace, workspace_bytes, actor_grad, critic_grad, metrics, stream);
}
