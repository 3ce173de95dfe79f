// Actor-critic learner step (IA2C / IPPO) for gfx950.  Replaces A2CNetwork.update / PPONetwork.update
// (marlbase/ac/model.py:189-246, 264-352) up to and including loss.backward(); clip + Adam + target update
// are marlhip_dqn_clip_adam on the joint [actor | critic] block.
//
// The step is built from two MFMA primitives shared with the DQN family (dqn_update_kernels.h, mlp.h) and
// one elementwise kernel between them:
//   forward rows   mlp_rows_fwd_kernel: out[p][row][:] = MLP_p(obs row) for the target critic (all T+1 rows),
//                  the critic and the actor (rows t < T).  Weights of one agent sit in LDS as A-operand packs.
//   ac_elem_kernel one thread per (t, b): n-step returns (marlbase/utils/utils.py:38-63), advantage, Categorical
//                  log-prob / entropy of the fp32 softmax, the policy-gradient (A2C) or clipped-surrogate (PPO)
//                  loss row, and dL/dlogits, dL/dvalue for every agent.
//   backward rows  dqn_lossgrad_kernel MODE 4 (hidden 64) / tp_bwd_kernel FULL (hidden 128): forward again +
//                  backward with the external output gradient, deterministic partial-record reduction, 1/sum(filled).
// Rows come straight from the ac/train.py Batch (agents innermost) through marlhip_batch's strides.
#include "a2c_core.h"

using namespace marl;

// (obs dim, hidden, actions) with compiled actor and critic (1 output) kernels: the LBF and warehouse shapes of common.h
#define MARL_AC_SHAPES(X) \
    X(12, 64, 6) X(15, 64, 6) X(18, 64, 6) X(21, 64, 6) X(24, 64, 6) X(27, 64, 6) X(39, 64, 6) X(12, 128, 6) X(15, 128, 6) X(18, 128, 6) X(21, 128, 6) X(24, 128, 6) X(27, 128, 6) X(39, 128, 6) \
    X(14, 64, 6) X(17, 64, 6) X(25, 64, 6) X(31, 64, 6) X(47, 64, 6) X(14, 128, 6) X(17, 128, 6) X(25, 128, 6) X(31, 128, 6) X(47, 128, 6) /* env.observe_id */ \
    X(71, 64, 5) X(71, 128, 5) /* rware */
// (agents, obs dim, hidden) with a compiled CENTRALISED critic (P * D inputs): hidden 128 up to 4 agents (weights and dW1
// accumulators of a P*D-wide first layer still fit the register file), hidden 64 for 2 agents (LDS-resident packs)
#define MARL_MAC_SHAPES(X) X(2, 12, 128) X(2, 15, 128) X(3, 18, 128) X(3, 24, 128) X(4, 21, 128) X(4, 27, 128) X(2, 12, 64) X(2, 15, 64)

// a centralised critic with a fused kernel (MARL_MAC_SHAPES); every other (agents, obs dim) takes the wide path (wide_mlp.h: run-time
// input width, activations in HBM) next to the fused actor kernels of MARL_AC_SHAPES
static bool mac_compiled(const marlhip_net_shape* s) {
    if (s->n_hidden != 0 && s->n_hidden != 2) return false;
#define X(p, d, h) if (s->n_agents == p && s->obs_dim == d && s->hidden == h && s->n_actions == 6) return true;
    MARL_MAC_SHAPES(X)
#undef X
    return false;
}

// actors (and independent critics) with fused kernels; every other two-layer shape runs BOTH networks on the GEMM path of
// wide_mlp.h (hidden > 128, observation widths / action counts outside MARL_AC_SHAPES): slower, any size
static bool ac_compiled(const marlhip_net_shape* s) {
    if (s->n_hidden != 0 && s->n_hidden != 2) return false;
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return true;
    MARL_AC_SHAPES(X)
#undef X
    return false;
}

static int ac_check(const marlhip_net_shape* s, int centralised = 0) {
    (void)centralised;
    MARL_REQUIRE(s != nullptr, "net shape is NULL");
    if (agent_map_validate(s, true) != 0) return -1;
    MARL_REQUIRE(s->n_agents >= 1, "ac: no agents");
    MARL_REQUIRE(s->obs_dim >= 1 && s->hidden >= 1 && s->hidden <= 1024 && s->n_actions >= 1 && s->n_actions <= 64,
                 "ac: net shape D=%d H=%d A=%d out of range", s->obs_dim, s->hidden, s->n_actions);
    return 0;
}

// the run-time shapes of the all-GEMM step: actor D -> H -> H -> A, critic (P * D or D) -> H -> H -> 1
// critic_L > 0: the critics' own number of hidden layers (marlhip_ac_config.critic_n_hidden: ac/model.py:45-97 builds actor and critic from
// their own `layers` lists), else the shape's
static void wide_set(const marlhip_net_shape* s, int centralised, int critic_L = 0) {
    const int L = s->n_hidden > 0 ? s->n_hidden : 2;
    WideRt<0>::set(s->obs_dim, s->hidden, s->n_actions, L);
    WideRt<1>::set(centralised ? s->n_agents * s->obs_dim : s->obs_dim, s->hidden, 1, critic_L > 0 ? critic_L : L);
}

extern "C" int marlhip_ac_critic_nparams(const marlhip_net_shape* s, int32_t centralised) {
    if (ac_check(s, centralised) != 0) return -1;
    if (centralised && mac_compiled(s)) {
#define X(p, d, h) if (s->n_agents == p && s->obs_dim == d && s->hidden == h) return MlpShape<p * d, h, 1>::NPARAM;
        MARL_MAC_SHAPES(X)
#undef X
    }
    if (!ac_compiled(s)) return (int)WideNet{centralised ? s->n_agents * s->obs_dim : s->obs_dim, s->hidden, 1, s->n_hidden > 0 ? s->n_hidden : 2}.nparam();
    if (centralised) return (int)WideNet{s->n_agents * s->obs_dim, s->hidden, 1}.nparam();
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return MlpShape<d, h, 1>::NPARAM;
    MARL_AC_SHAPES(X)
#undef X
    return -1;
}

extern "C" int64_t marlhip_ac_workspace_bytes_lc(const marlhip_net_shape* s, int32_t centralised, int32_t critic_n_hidden, int32_t max_len,
                                                  int32_t batch) {
    if (critic_n_hidden <= 0 || critic_n_hidden == (s != nullptr && s->n_hidden > 0 ? s->n_hidden : 2))
        return marlhip_ac_workspace_bytes(s, centralised, max_len, batch);
    if (ac_check(s, centralised) != 0) return -1;
    MARL_REQUIRE(critic_n_hidden <= 16, "ac_workspace_bytes: %d hidden layers for the critics (1..16)", critic_n_hidden);
    MARL_REQUIRE(!ac_compiled(s), "ac_workspace_bytes: critics of another depth than the actors run on the GEMM path - give the shape a width without "
                                  "fused kernels (hidden > 128)");
    wide_set(s, centralised, critic_n_hidden);
    return ac_ws_layout<WideRt<0>, WideRt<1>>(s->n_agents, max_len, batch).total;
}

extern "C" int64_t marlhip_ac_workspace_bytes(const marlhip_net_shape* s, int32_t centralised, int32_t max_len, int32_t batch) {
    if (ac_check(s, centralised) != 0) return -1;
    if (centralised && mac_compiled(s)) {
#define X(p, d, h)                                                   \
    if (s->n_agents == p && s->obs_dim == d && s->hidden == h)       \
        return ac_ws_layout<MlpShape<d, h, 6>, MlpShape<p * d, h, 1>>(s->n_agents, max_len, batch).total;
        MARL_MAC_SHAPES(X)
#undef X
    }
    if (!ac_compiled(s)) {
        wide_set(s, centralised);
        return ac_ws_layout<WideRt<0>, WideRt<1>>(s->n_agents, max_len, batch).total;
    }
    if (centralised) {
#define X(d, h, a)                                                       \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) {        \
        WideCritic<h>::D = s->n_agents * d;                              \
        return ac_ws_layout<MlpShape<d, h, a>, WideCritic<h>>(s->n_agents, max_len, batch).total; \
    }
        MARL_AC_SHAPES(X)
#undef X
    }
#define X(d, h, a) \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return ac_ws_layout<MlpShape<d, h, a>, MlpShape<d, h, 1>>(s->n_agents, max_len, batch).total;
    MARL_AC_SHAPES(X)
#undef X
    return -1;
}

static int ac_call(const marlhip_net_shape* s, const float* actor, const float* critic, const float* target, const marlhip_batch* bt,
                   const marlhip_ac_config* c, int mode, void* ws, int64_t ws_bytes, float* actor_grad, float* critic_grad,
                   float* metrics, void* stream) {
    MARL_REQUIRE(actor && critic && bt && c && ws, "ac_loss_grad: NULL pointer");
    if (ac_check(s, c->centralised_critic) != 0) return -1;
    MARL_REQUIRE(mode == 1 || (actor_grad && critic_grad && metrics), "ac_loss_grad: NULL output");
    MARL_REQUIRE(mode == 2 || target != nullptr, "ac_loss_grad: NULL target critic");
    MARL_REQUIRE(bt->obss && bt->actions && bt->rewards && bt->dones && bt->filled, "ac_loss_grad: NULL batch field");
    MARL_REQUIRE(bt->max_len > 0 && bt->batch > 0, "ac_loss_grad: empty batch");
    MARL_REQUIRE(c->n_steps >= 1 && c->n_steps <= 16, "ac_loss_grad: n_steps %d outside [1, 16]", c->n_steps);
    MARL_REQUIRE((c->ret_mean == nullptr) == (c->ret_var == nullptr) && (c->ret_mean == nullptr) == (c->ret_count == nullptr),
                 "ac_loss_grad: return statistics must be given together (mean, var, count) or not at all");
#define MARL_AC_ARGS s->n_agents, agent_map(s), actor, critic, target, bt, c, mode, ws, ws_bytes, actor_grad, critic_grad, metrics, (hipStream_t)stream
    if (!ac_compiled(s)) {  // both networks on the GEMM path
        MARL_REQUIRE(!c->centralised_critic || (bt->obs_row_stride == (int64_t)s->n_agents * s->obs_dim && bt->obs_agent_stride == s->obs_dim),
                     "ac_loss_grad: a centralised critic needs the ac/train.py Batch layout (agents concatenated in a row)");
        MARL_REQUIRE(c->critic_n_hidden >= 0 && c->critic_n_hidden <= 16, "ac_loss_grad: critic_n_hidden %d outside [0, 16]", c->critic_n_hidden);
        wide_set(s, c->centralised_critic, c->critic_n_hidden);
        return ac_step_t<WideRt<0>, WideRt<1>>(MARL_AC_ARGS);
    }
    MARL_REQUIRE(c->critic_n_hidden == 0 || c->critic_n_hidden == 2, "ac_loss_grad: critics with %d hidden layers next to fused two-layer actors - the "
                 "GEMM path takes both networks (give the shape a width without fused kernels)", c->critic_n_hidden);
    if (c->centralised_critic) {
        MARL_REQUIRE(bt->obs_row_stride == (int64_t)s->n_agents * s->obs_dim && bt->obs_agent_stride == s->obs_dim,
                     "ac_loss_grad: a centralised critic needs the ac/train.py Batch layout (agents concatenated in a row)");
        if (mac_compiled(s)) {
#define X(p, d, h) if (s->n_agents == p && s->obs_dim == d && s->hidden == h) return ac_step<d, h, 6, p * d>(MARL_AC_ARGS);
            MARL_MAC_SHAPES(X)
#undef X
        }
#define X(d, h, a)                                                       \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) {        \
        WideCritic<h>::D = s->n_agents * d;                              \
        return ac_step_t<MlpShape<d, h, a>, WideCritic<h>>(MARL_AC_ARGS); \
    }
        MARL_AC_SHAPES(X)
#undef X
    }
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return ac_step<d, h, a, d>(MARL_AC_ARGS);
    MARL_AC_SHAPES(X)
#undef X
#undef MARL_AC_ARGS
    return -1;
}

extern "C" int marlhip_ac_forward_rows(const marlhip_net_shape* s, int32_t value_net, const float* params, const float* obs,
                                       int64_t agent_stride, int64_t row_stride, int32_t n_rows, float* out, void* workspace,
                                       int64_t workspace_bytes, void* stream) {
    if (ac_check(s, value_net == 2) != 0) return -1;
    ScratchScope scratch(workspace, workspace_bytes);
    MARL_REQUIRE(params && obs && out && n_rows > 0 && row_stride > 0 && (agent_stride > 0 || value_net == 2), "ac_forward_rows: bad argument");
    marlhip_batch bt = {};
    bt.obss = obs; bt.max_len = 1; bt.batch = 1;
    bt.obs_agent_stride = value_net == 2 ? -1 : agent_stride;
    bt.obs_row_stride = row_stride;
    if (value_net == 2 && mac_compiled(s)) {
#define X(p, d, h)                                               \
    if (s->n_agents == p && s->obs_dim == d && s->hidden == h)   \
        return launch_forward_rows<MlpShape<p * d, h, 1>>(s->n_agents, agent_map(s), params, &bt, n_rows, out, (hipStream_t)stream);
        MARL_MAC_SHAPES(X)
#undef X
    }
    if (value_net == 2 && ac_compiled(s)) {  // wide centralised critics: the fused kernels of wide_critic.h (the values the learner step sees, bit for bit)
#define X(d, h, a)                                                                                                           \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) {                                                             \
        WideCritic<h>::D = s->n_agents * d;                                                                                   \
        return launch_forward_rows<WideCritic<h>>(s->n_agents, agent_map(s), params, &bt, n_rows, out, (hipStream_t)stream);  \
    }
        MARL_AC_SHAPES(X)
#undef X
    }
    if (value_net == 2 || !ac_compiled(s)) {  // the GEMM path: any input / hidden width
        WideRt<1>::set(value_net == 2 ? s->n_agents * s->obs_dim : s->obs_dim, s->hidden, value_net ? 1 : s->n_actions, s->n_hidden > 0 ? s->n_hidden : 2);
        return launch_forward_rows<WideRt<1>>(s->n_agents, agent_map(s), params, &bt, n_rows, out, (hipStream_t)stream);
    }
#define X(d, h, a)                                                                                                         \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a)                                                             \
        return value_net ? launch_forward_rows<MlpShape<d, h, 1>>(s->n_agents, agent_map(s), params, &bt, n_rows, out, (hipStream_t)stream) \
                         : launch_forward_rows<MlpShape<d, h, a>>(s->n_agents, agent_map(s), params, &bt, n_rows, out, (hipStream_t)stream);
    MARL_AC_SHAPES(X)
#undef X
    return -1;
}

// ---- the rollout that keeps the actors' forward pass for the step (AcKeep, common.h; ac_keep_layout, a2c_core.h) -------------------
namespace {
template <class T>
struct ShapeTag {
    using type = T;
};

// f(actor shape, critic shape) for the shapes with FUSED actors - ac_call's dispatch for them
template <class F>
int ac_fused_actor_dispatch(const marlhip_net_shape* s, int centralised, F&& f) {
    MARL_REQUIRE(ac_compiled(s), "ac_collect_keep: actors D=%d H=%d A=%d run on the GEMM path, which has no fused collector", s->obs_dim, s->hidden,
                 s->n_actions);
    if (centralised) {
        if (mac_compiled(s)) {
#define X(p, d, h) if (s->n_agents == p && s->obs_dim == d && s->hidden == h) return f(ShapeTag<MlpShape<d, h, 6>>{}, ShapeTag<MlpShape<p * d, h, 1>>{});
            MARL_MAC_SHAPES(X)
#undef X
        }
#define X(d, h, a)                                                                   \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) {                    \
        WideCritic<h>::D = s->n_agents * d;                                          \
        return f(ShapeTag<MlpShape<d, h, a>>{}, ShapeTag<WideCritic<h>>{});          \
    }
        MARL_AC_SHAPES(X)
#undef X
    }
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return f(ShapeTag<MlpShape<d, h, a>>{}, ShapeTag<MlpShape<d, h, 1>>{});
    MARL_AC_SHAPES(X)
#undef X
    set_error("ac_collect_keep: no fused actor shape D=%d H=%d A=%d", s->obs_dim, s->hidden, s->n_actions);
    return -1;
}

int ac_keep_for(const marlhip_net_shape* s, int centralised, int T, int B, void* ws, int64_t ws_bytes, AcKeep* k, hipStream_t st) {
    MARL_REQUIRE(ws != nullptr, "ac_collect_keep: NULL learner workspace");
    if (ac_check(s, centralised) != 0) return -1;
    return ac_fused_actor_dispatch(s, centralised, [&](auto sa, auto sc) {
        return ac_keep_layout<typename decltype(sa)::type, typename decltype(sc)::type>(s->n_agents, T, B, ws, ws_bytes, k, st);
    });
}
}  // namespace

extern "C" int marlhip_ac_collect_keep(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const float* actor_params, uint32_t round,
                                       int32_t max_len, int32_t use_proper_termination, float* batch_obs, int64_t* batch_act, float* batch_rew,
                                       uint8_t* batch_done, float* batch_filled, float* fin_return, int32_t* fin_length, int32_t* t_max,
                                       void* workspace, int64_t workspace_bytes, int32_t centralised, void* learner_workspace,
                                       int64_t learner_workspace_bytes, void* stream) {
    MARL_REQUIRE(cfg != nullptr, "ac_collect_keep: NULL config");
    AcKeep k = {};
    if (ac_keep_for(s, centralised, max_len, cfg->n_envs, learner_workspace, learner_workspace_bytes, &k, (hipStream_t)stream) != 0) return -1;
    AcKeepScope scope(k);
    return marlhip_ac_collect(cfg, s, actor_params, round, max_len, use_proper_termination, batch_obs, batch_act, batch_rew, batch_done, batch_filled,
                              fin_return, fin_length, t_max, workspace, workspace_bytes, stream);
}

extern "C" int marlhip_rware_ac_collect_keep(const marlhip_rware_config* cfg, const marlhip_net_shape* s, const float* actor_params, uint32_t round,
                                             int32_t max_len, int32_t use_proper_termination, float* batch_obs, int64_t* batch_act, float* batch_rew,
                                             uint8_t* batch_done, float* batch_filled, float* fin_return, int32_t* fin_length, int32_t* t_max,
                                             void* workspace, int64_t workspace_bytes, int32_t centralised, void* learner_workspace,
                                             int64_t learner_workspace_bytes, void* stream) {
    MARL_REQUIRE(cfg != nullptr, "rware_ac_collect_keep: NULL config");
    AcKeep k = {};
    if (ac_keep_for(s, centralised, max_len, cfg->n_envs, learner_workspace, learner_workspace_bytes, &k, (hipStream_t)stream) != 0) return -1;
    AcKeepScope scope(k);
    return marlhip_rware_ac_collect(cfg, s, actor_params, round, max_len, use_proper_termination, batch_obs, batch_act, batch_rew, batch_done,
                                    batch_filled, fin_return, fin_length, t_max, workspace, workspace_bytes, stream);
}

extern "C" int marlhip_a2c_loss_grad(const marlhip_net_shape* s, const float* actor, const float* critic, const float* target_critic,
                                     const marlhip_batch* batch, const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes,
                                     float* actor_grad, float* critic_grad, float* metrics, void* stream) {
    return ac_call(s, actor, critic, target_critic, batch, cfg, 0, workspace, workspace_bytes, actor_grad, critic_grad, metrics, stream);
}

extern "C" int marlhip_ppo_prepare(const marlhip_net_shape* s, const float* actor, const float* critic, const float* target_critic,
                                   const marlhip_batch* batch, const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
    return ac_call(s, actor, critic, target_critic, batch, cfg, 1, workspace, workspace_bytes, nullptr, nullptr, nullptr, stream);
}

extern "C" int marlhip_ppo_loss_grad(const marlhip_net_shape* s, const float* actor, const float* critic, const marlhip_batch* batch,
                                     const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes, float* actor_grad,
                                     float* critic_grad, float* metrics, void* stream) {
    return ac_call(s, actor, critic, nullptr, batch, cfg, 2, workspace, workspace_bytes, actor_grad, critic_grad, metrics, stream);
}
