// extern "C" entry points of the actor-critic learner step with recurrent networks (`use_rnn: True` in ia2c.yaml / ippo.yaml):
// a2c_core.h's step with GruShape actors and critics - target-critic / critic / actor SEQUENCE forwards from zero hidden states
// (ac/model.py:190-193,205-214), the same elementwise kernel, back-propagation through time for the two backward passes.
#include "a2c_core.h"

using namespace marl;

// (obs dim, hidden, actions): the LBF widths and the warehouse, hidden 64 and 128; critics are the same template with one output
#define MARL_GRU_AC_SHAPES(X)                                                                                \
    X(12, 64, 6) X(15, 64, 6) X(18, 64, 6) X(21, 64, 6) X(24, 64, 6) X(27, 64, 6) X(39, 64, 6) X(71, 64, 5) \
    X(12, 128, 6) X(15, 128, 6) X(18, 128, 6) X(21, 128, 6) X(24, 128, 6) X(27, 128, 6) X(39, 128, 6) X(71, 128, 5) \
    X(14, 64, 6) X(17, 64, 6) X(25, 64, 6) X(31, 64, 6) X(47, 64, 6) X(14, 128, 6) X(17, 128, 6) X(25, 128, 6) X(31, 128, 6) X(47, 128, 6) /* env.observe_id */

// (agents, obs dim, hidden) with a compiled recurrent CENTRALISED critic (P * D inputs, critic.centralised: maa2c / mappo with use_rnn)
#define MARL_GRU_MAC_SHAPES(X) X(2, 12, 64) X(2, 15, 64) X(3, 18, 64) X(3, 24, 64) X(4, 21, 64) X(4, 27, 64) X(2, 12, 128) X(2, 15, 128) X(3, 18, 128) X(3, 24, 128) X(4, 21, 128) X(4, 27, 128)

static int gru_ac_check(const marlhip_net_shape* s, int centralised = 0) {
    MARL_REQUIRE(s != nullptr, "net shape is NULL");
    if (agent_map_validate(s, true) != 0) return -1;
    MARL_REQUIRE(gru_depth(s) >= 1 && gru_depth(s) <= GRU_MAX_LAYERS, "recurrent networks: n_hidden %d = %d stacked GRU layers (1..%d)", s->n_hidden,
                 gru_depth(s), GRU_MAX_LAYERS);
    if (centralised) {
#define X(p, d, h) if (s->n_agents == p && s->obs_dim == d && s->hidden == h && s->n_actions == 6) return 0;
        MARL_GRU_MAC_SHAPES(X)
#undef X
        set_error("no recurrent centralised-critic kernels for %d agents x obs_dim %d, hidden %d (MARL_GRU_MAC_SHAPES)", s->n_agents, s->obs_dim,
                  s->hidden);
        return -1;
    }
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return 0;
    MARL_GRU_AC_SHAPES(X)
#undef X
    set_error("no recurrent actor-critic kernels for obs_dim %d, hidden %d, %d actions", s->obs_dim, s->hidden, s->n_actions);
    return -1;
}

extern "C" int marlhip_gru_ac_critic_nparams(const marlhip_net_shape* s, int32_t centralised) {
    if (gru_ac_check(s, centralised) != 0) return -1;
    if (centralised) {
#define X(p, d, h) if (s->n_agents == p && s->obs_dim == d && s->hidden == h) return GruShape<p * d, h, 1>::nparam(gru_depth(s));
        MARL_GRU_MAC_SHAPES(X)
#undef X
    }
#define X(d, h, a) if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return GruShape<d, h, 1>::nparam(gru_depth(s));
    MARL_GRU_AC_SHAPES(X)
#undef X
    return -1;
}

static int gru_critic_depth(const marlhip_net_shape* s, int critic_n_hidden) { return critic_n_hidden > 0 ? critic_n_hidden - 1 : gru_depth(s); }

extern "C" int64_t marlhip_gru_ac_workspace_bytes_lc(const marlhip_net_shape* s, int32_t centralised, int32_t critic_n_hidden, int32_t max_len, int32_t batch) {
    if (gru_ac_check(s, centralised) != 0) return -1;
    MARL_REQUIRE(critic_n_hidden == 0 || (critic_n_hidden >= 2 && critic_n_hidden <= GRU_MAX_LAYERS + 1), "gru_ac_workspace_bytes: critic_n_hidden %d (0, 2..%d)",
                 critic_n_hidden, GRU_MAX_LAYERS + 1);
    const int Lc = gru_critic_depth(s, critic_n_hidden);
    if (centralised) {
#define X(p, d, h) \
    if (s->n_agents == p && s->obs_dim == d && s->hidden == h) return ac_ws_layout<GruShape<d, h, 6>, GruShape<p * d, h, 1>>(s->n_agents, max_len, batch, gru_depth(s), Lc).total;
        MARL_GRU_MAC_SHAPES(X)
#undef X
    }
#define X(d, h, a) \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) return ac_ws_layout<GruShape<d, h, a>, GruShape<d, h, 1>>(s->n_agents, max_len, batch, gru_depth(s), Lc).total;
    MARL_GRU_AC_SHAPES(X)
#undef X
    return -1;
}

extern "C" int64_t marlhip_gru_ac_workspace_bytes(const marlhip_net_shape* s, int32_t centralised, int32_t max_len, int32_t batch) {
    return marlhip_gru_ac_workspace_bytes_lc(s, centralised, 0, max_len, batch);
}

static int gru_ac_call(const marlhip_net_shape* s, const float* actor, const float* critic, const float* target, const marlhip_batch* bt,
                       const marlhip_ac_config* c, int mode, void* ws, int64_t ws_bytes, float* actor_grad, float* critic_grad, float* metrics,
                       void* stream) {
    MARL_REQUIRE(c != nullptr, "gru_ac_loss_grad: NULL config");
    if (gru_ac_check(s, c->centralised_critic) != 0) return -1;
    MARL_REQUIRE(actor && critic && bt && ws, "gru_ac_loss_grad: NULL pointer");
    MARL_REQUIRE(mode == 1 || (actor_grad && critic_grad && metrics), "gru_ac_loss_grad: NULL output");
    MARL_REQUIRE(mode == 2 || target != nullptr, "gru_ac_loss_grad: NULL target critic");
    MARL_REQUIRE(bt->obss && bt->actions && bt->rewards && bt->dones && bt->filled, "gru_ac_loss_grad: NULL batch field");
    MARL_REQUIRE(c->n_steps >= 1 && c->n_steps <= 16, "gru_ac_loss_grad: n_steps %d (1..16)", c->n_steps);
    MARL_REQUIRE(c->critic_n_hidden == 0 || (c->critic_n_hidden >= 2 && c->critic_n_hidden <= GRU_MAX_LAYERS + 1),
                 "gru_ac_loss_grad: critic_n_hidden %d (0 = as the actors; 2..%d = 1..%d stacked GRU layers)", c->critic_n_hidden, GRU_MAX_LAYERS + 1, GRU_MAX_LAYERS);
    if (c->centralised_critic) {
        MARL_REQUIRE(bt->obs_agent_stride == s->obs_dim && bt->obs_row_stride == (int64_t)s->n_agents * s->obs_dim,
                     "gru_ac_loss_grad: a centralised critic needs the ac/train.py Batch layout (agents concatenated in a row)");
#define X(p, d, h)                                                                                                                       \
    if (s->n_agents == p && s->obs_dim == d && s->hidden == h)                                                                           \
        return ac_step_t<GruShape<d, h, 6>, GruShape<p * d, h, 1>>(s->n_agents, agent_map(s), actor, critic, target, bt, c, mode, ws, ws_bytes, actor_grad, \
                                                                   critic_grad, metrics, (hipStream_t)stream);
        MARL_GRU_MAC_SHAPES(X)
#undef X
    }
#define X(d, h, a)                                                                                                                   \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a)                                                                      \
        return ac_step_t<GruShape<d, h, a>, GruShape<d, h, 1>>(s->n_agents, agent_map(s), actor, critic, target, bt, c, mode, ws, ws_bytes, actor_grad, \
                                                               critic_grad, metrics, (hipStream_t)stream);
    MARL_GRU_AC_SHAPES(X)
#undef X
    return -1;
}

extern "C" int marlhip_gru_a2c_loss_grad(const marlhip_net_shape* s, const float* actor, const float* critic, const float* target_critic,
                                         const marlhip_batch* batch, const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes,
                                         float* actor_grad, float* critic_grad, float* metrics, void* stream) {
    return gru_ac_call(s, actor, critic, target_critic, batch, cfg, 0, workspace, workspace_bytes, actor_grad, critic_grad, metrics, stream);
}

extern "C" int marlhip_gru_ppo_prepare(const marlhip_net_shape* s, const float* actor, const float* critic, const float* target_critic,
                                       const marlhip_batch* batch, const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes, void* stream) {
    return gru_ac_call(s, actor, critic, target_critic, batch, cfg, 1, workspace, workspace_bytes, nullptr, nullptr, nullptr, stream);
}

extern "C" int marlhip_gru_ppo_loss_grad(const marlhip_net_shape* s, const float* actor, const float* critic, const marlhip_batch* batch,
                                         const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes, float* actor_grad, float* critic_grad,
                                         float* metrics, void* stream) {
    return gru_ac_call(s, actor, critic, nullptr, batch, cfg, 2, workspace, workspace_bytes, actor_grad, critic_grad, metrics, stream);
}

// sequence forward of the actors (value_net = 0: logits [P][steps][B][A]) or critics (value_net = 1: values [P][steps][B][1]) with the hidden
// state carried by the caller (A2CNetwork.act / get_value, ac/model.py:147-163); obs rows at obs + p * agent_stride + (t * B + b) * row_stride
extern "C" int marlhip_gru_ac_forward(const marlhip_net_shape* s, int32_t value_net, const float* params, const float* obs, int64_t agent_stride,
                                      int64_t row_stride, int32_t steps, int32_t batch, const float* h_in, float* h_out, float* out, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
    if (gru_ac_check(s, value_net == 2) != 0) return -1;
    ScratchScope scratch(workspace, workspace_bytes);
    MARL_REQUIRE(params && obs && out && steps > 0 && batch > 0 && row_stride > 0 && agent_stride >= 0, "gru_ac_forward: bad argument");
    const hipStream_t st = (hipStream_t)stream;
    const int P = s->n_agents;
    auto run = [&](auto shape) -> int {
        using S = decltype(shape);
        const int L = gru_depth(s);  // h_in / h_out: [L][P][batch][H]
        const size_t pack_bytes = ((size_t)L * P * S::NFWD * sizeof(float) + 255) & ~(size_t)255;
        const size_t chain_bytes = L > 1 ? (size_t)(L - 1) * gru_layer_rec<S>(P, steps, batch) * sizeof(float) : 0;
        float* packs = collect_pack_scratch(pack_bytes + chain_bytes, st);  // (marlhip_gru_forward_workspace_bytes)
        if (packs == nullptr) return -1;  // error text set by collect_pack_scratch
        gru_set_attrs<S>();
        gru_pack_fwd_layers<S>(P, L, params, agent_map(s), packs, st);
        gru_fwd_layers<S>(P, L, packs, obs, (size_t)agent_stride, (size_t)row_stride, steps, batch, h_in, h_out, out,
                          chain_bytes ? reinterpret_cast<float*>(reinterpret_cast<char*>(packs) + pack_bytes) : (float*)nullptr, false, st);
        MARL_CHECK_LAUNCH("gru_seq_fwd_kernel (ac forward)");
        return 0;
    };
    if (value_net == 2) {  // centralised critic: every agent's critic reads the same P * D row (agent_stride 0)
#define X(p, d, h) if (s->n_agents == p && s->obs_dim == d && s->hidden == h) return run(GruShape<p * d, h, 1>{});
        MARL_GRU_MAC_SHAPES(X)
#undef X
    }
#define X(d, h, a)                                                 \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) {  \
        if (value_net) return run(GruShape<d, h, 1>{});            \
        return run(GruShape<d, h, a>{});                           \
    }
    MARL_GRU_AC_SHAPES(X)
#undef X
    return -1;
}

// Categorical(logits).sample() for N envs (A2CNetwork.act, ac/model.py:147-153) the way the fused rollout collector draws it: inverse
// CDF of the fp32 softmax (sequential sums) with one Philox uniform per (env, step, agent) - word 1 + p of the action-noise block
static __global__ __launch_bounds__(256) void sample_logits_kernel(int P, int N, int A, const float* __restrict__ logits, uint64_t seed,
                                                            const uint32_t* __restrict__ episode, int t, int64_t* __restrict__ actions) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    for (int p = 0; p < P; ++p) {
        const float* l = logits + ((size_t)p * N + n) * A;
        float m = l[0];
        for (int a = 1; a < A; ++a) m = fmaxf(m, l[a]);
        float sum = 0.f;
        for (int a = 0; a < A; ++a) sum += expf(l[a] - m);
        const float thr = u01_f32(act_noise_word(seed, (uint32_t)n, episode[n], (uint32_t)t, 1 + p)) * sum;
        int act = A - 1;
        for (int a = A - 1; a >= 0; --a) {  // first a with cumsum(e)[a] > thr (cumsums formed front to back)
            float ca = 0.f;
            for (int b = 0; b <= a; ++b) ca += expf(l[b] - m);
            if (ca > thr) act = a;
        }
        actions[(size_t)p * N + n] = act;
    }
}

extern "C" int marlhip_sample_from_logits(int32_t n_agents, int32_t n_envs, int32_t n_actions, const float* logits, uint64_t seed,
                                          const uint32_t* episode, int32_t t, int64_t* actions, void* stream) {
    MARL_REQUIRE(logits && episode && actions && n_agents > 0 && n_envs > 0 && n_actions > 0, "sample_from_logits: bad argument");
    hipLaunchKernelGGL(sample_logits_kernel, dim3((n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_agents, n_envs, n_actions, logits, seed,
                       episode, t, actions);
    MARL_CHECK_LAUNCH("sample_logits_kernel");
    return 0;
}
