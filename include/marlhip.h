/* marlhip.h - C-ABI of libmarlhip.so, the MI355X (gfx950) implementation of marlbase's
 * independent-learner hot path.
 *
 * The reference (marl-book/codebase) is pure Python: it has no FFI/plugin registry.  Its
 * extension surface is Hydra `_target_` callables plus duck typing.  Each entry
 * point below names the reference code it stands in for; the Python adapters in codebase_amd/
 * re-create the reference objects (make_env, ReplayBuffer, QNetwork, dqn.train.main) on top of
 * these calls, and INTEGRATION.md shows the ctypes stubs a marlbase maintainer would add.
 *
 * Conventions
 *  - every call returns 0 on success, <0 on error (text: marlhip_last_error()).
 *  - the CALLER owns all memory: arguments are raw DEVICE pointers (PyTorch-ROCm tensors) plus
 *    element counts; the library never allocates or frees device memory.  Entry points that need
 *    scratch take a `workspace` (sizes: the marlhip_*_workspace_bytes functions).
 *  - every call takes a hipStream_t (as void*), only enqueues, never synchronises.  The one call that
 *    overlaps work on a second stream (the recurrent actor-critic step) takes that stream from the
 *    caller (marlhip_ac_config.side_stream) and joins it back before it returns.
 *  - state kept between calls: the text behind marlhip_last_error() (per host thread), whether a
 *    kernel's dynamic-LDS attribute has been raised on a device (idempotent), and - only while
 *    marlhip_timing_enable(1) - the HIP events of the measurement aid at the end of this file.
 *    Nothing else: calls on different host threads / streams share no buffers.
 *  - layouts are agent-major, env-minor: obs[P][N][D] f32, actions[P][N] i32, rewards[P][N] f32,
 *    done[N] u8.  N = envs, P = agents, D = obs dim, A = actions, T = time_limit, B = batch.
 *  - integer results (env state, done flags, observations, greedy actions, sampled indices) are
 *    bit-exact functions of the inputs; fp32 losses/gradients agree with torch fp32 to ~1e-6.
 */
#ifndef MARLHIP_H
#define MARLHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MARLHIP_VERSION 219

int marlhip_version(void);
const char* marlhip_last_error(void);
/* 1 if a HIP device is usable from this process (hipGetDeviceCount > 0), else 0 */
int marlhip_device_available(void);

/* ------------------------------------------------------------------------------------------
 * Level-Based Foraging, batched.   Replaces lbforaging's ForagingEnv (third-party; reached via
 * gym.make at marlbase/utils/envs.py:90-92) under TimeLimit + RecordEpisodeStatistics
 * (utils/envs.py:96-97, utils/wrappers.py:13-45) and the optional CooperativeReward wrapper
 * (utils/wrappers.py:106-108).  Field values mirror upstream's registration kwargs.
 * ---------------------------------------------------------------------------------------- */
typedef struct marlhip_lbf_config {
    int32_t n_envs;
    int32_t n_agents;          /* players */
    int32_t n_food;            /* max_num_food */
    int32_t rows, cols;        /* field_size */
    int32_t sight;             /* == size for full observability, 2 for "-2s" */
    int32_t max_episode_steps; /* upstream registration: 50 -> `done` */
    int32_t time_limit;        /* env.time_limit (TimeLimit wrapper) -> `truncated`; 0 = none */
    int32_t force_coop;
    int32_t min_player_level, max_player_level;
    int32_t min_food_level, max_food_level; /* max_food_level <= 0: None (sum of 3 lowest player levels) */
    int32_t normalize_reward;
    int32_t cooperative;       /* CooperativeReward wrapper */
    double penalty;
    uint64_t seed;             /* Philox key */
    float* reward_stats;       /* env.standardise_rewards (StandardiseReward, utils/wrappers.py:111-142): device array
                                  [n_envs][3*n_agents + 1] fp32, zero-initialised, one streaming mean / variance record per
                                  env (sumw | wmean | t per agent, step count as int32 bits) that persists across episodes;
                                  NULL = wrapper off.  Applied before CooperativeReward, after RecordEpisodeStatistics. */
    int32_t observe_id;        /* env.observe_id (ObserveID, utils/wrappers.py:73-103): every observation is prefixed with the
                                  one-hot agent index, obs_dim = n_agents + 3 * (n_food + n_agents) */
} marlhip_lbf_config;

/* device buffers of a batched env (allocated by the caller) */
typedef struct marlhip_lbf_buffers {
    uint8_t* state;     /* [N][marlhip_lbf_state_stride] packed records: F x (row,col,level) u8 in
                           row-major field order (level 0 = absent), P x (row,col,level) u8,
                           current_step u16 LE, food_spawned u16 LE, zero pad to 4 B */
    uint32_t* episode;  /* [N] episodes started so far (index into the Philox reset stream) */
    float* ep_return;   /* [P][N] running RAW per-agent return (RecordEpisodeStatistics, fp32 adds) */
    int32_t* ep_length; /* [N] steps taken in the running episode */
} marlhip_lbf_buffers;

int marlhip_lbf_state_stride(const marlhip_lbf_config* cfg); /* bytes per env record, <0 if unsupported */
int marlhip_lbf_obs_dim(const marlhip_lbf_config* cfg);      /* 3 * (n_food + n_agents) [+ n_agents with observe_id] */

/* env.reset(): re-spawn the envs with mask[n] != 0 (mask NULL = all), episode[n] += 1 for those,
 * zero their running statistics, write their observations (obs may be NULL).
 * Call sites replaced: marlbase/dqn/train.py:180,203,246,267; utils/envs.py:111. */
int marlhip_lbf_reset(const marlhip_lbf_config* cfg, const marlhip_lbf_buffers* buf, const uint8_t* mask,
                      float* obs /* [P][N][D] */, void* stream);

/* observations of the current state (no transition) */
int marlhip_lbf_observe(const marlhip_lbf_config* cfg, const marlhip_lbf_buffers* buf, float* obs, void* stream);

/* env.step(actions): one joint transition of every env with active[n] != 0 (NULL = all).
 * rewards are what the learner sees (fp64 -> fp32 once; summed first if cfg->cooperative).
 * On done|truncated the finished episode's statistics (info["episode_returns"],
 * info["episode_length"], wrappers.py:35-41) are latched into fin_return/fin_length.
 * auto_reset != 0 gives gymnasium(<1.0) AsyncVectorEnv semantics (marlbase/ac/train.py:79):
 * a finished env is reset inside the call, `obs` holds the reset observation and final_obs
 * (may be NULL) the terminal one.
 * Call sites replaced: marlbase/dqn/train.py:191,217,257; marlbase/ac/train.py:79. */
int marlhip_lbf_step(const marlhip_lbf_config* cfg, const marlhip_lbf_buffers* buf, const uint8_t* active,
                     const int32_t* actions /* [P][N] */, float* obs /* [P][N][D] */, float* rewards /* [P][N] */,
                     uint8_t* done /* [N] */, uint8_t* truncated /* [N] */, float* fin_return /* [P][N] */,
                     int32_t* fin_length /* [N] */, int32_t auto_reset, float* final_obs /* [P][N][D] or NULL */,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-robot warehouse (rware), batched.   Replaces rware's Warehouse (third-party; reached via gym.make at
 * marlbase/utils/envs.py:27-37,90-92) with flattened observations, msg_bits 0, sensor_range 1, under the same wrapper
 * stack as above.  Field values mirror upstream's `rware-{tiny,small,medium,large}-{n}ag[-easy|-hard]-v2`
 * registration: tiny = 1 shelf row x 3 shelf columns, column_height 8, request_queue_size n (2n easy, n/2 hard),
 * max_steps 500, individual rewards.  5 actions (noop, forward, left, right, toggle-load); observation = 71 floats
 * (x, y, carrying, direction one-hot, on-highway, then 3x3 cells x (agent, direction one-hot, shelf, requested)).
 * Movement conflicts follow upstream's graph rule (cycles move, swaps do not, else the longest chain into a free
 * cell moves); equally long chains meeting at one cell: the lowest agent index wins (upstream: CPython set order).
 * Replacement requests draw from Philox stream 3 over the un-queued shelves in id order.
 * Buffers are a marlhip_lbf_buffers whose state records are marlhip_rware_state_stride bytes:
 * grid u8[rows*cols] (shelf id on the cell, 0 none) | P x (x, y, dir, carried shelf, has_delivered) u8 |
 * 2P queue ids u8 | steps u16 LE | inactive steps u16 LE | pad to 4 B.
 * ---------------------------------------------------------------------------------------- */
typedef struct marlhip_rware_config {
    int32_t n_envs;
    int32_t n_agents;
    int32_t shelf_rows, shelf_columns, column_height;
    int32_t request_queue_size;
    int32_t max_steps;            /* upstream registration: 500 -> `done`; 0 = none */
    int32_t max_inactivity_steps; /* 0 = none */
    int32_t time_limit;           /* env.time_limit (TimeLimit wrapper) -> `truncated`; 0 = none */
    int32_t reward_type;          /* 0 global, 1 individual, 2 two-stage */
    int32_t cooperative;          /* CooperativeReward wrapper */
    int32_t observe_id;           /* ObserveID wrapper: one-hot agent index in front of every observation */
    uint64_t seed;                /* Philox key */
    float* reward_stats;          /* StandardiseReward records [n_envs][3*n_agents + 1] or NULL (see marlhip_lbf_config) */
} marlhip_rware_config;

int marlhip_rware_state_stride(const marlhip_rware_config* cfg); /* bytes per env record, <0 if unsupported */
int marlhip_rware_obs_dim(const marlhip_rware_config* cfg);      /* 71 [+ n_agents with observe_id] */
int marlhip_rware_grid(const marlhip_rware_config* cfg, int32_t* rows, int32_t* cols, int32_t* n_shelves);
/* same contracts as marlhip_lbf_reset / _observe / _step; actions in [0, 5) */
int marlhip_rware_reset(const marlhip_rware_config* cfg, const marlhip_lbf_buffers* buf, const uint8_t* mask,
                        float* obs /* [P][N][D] */, void* stream);
int marlhip_rware_observe(const marlhip_rware_config* cfg, const marlhip_lbf_buffers* buf, float* obs, void* stream);
int marlhip_rware_step(const marlhip_rware_config* cfg, const marlhip_lbf_buffers* buf, const uint8_t* active,
                       const int32_t* actions /* [P][N] */, float* obs /* [P][N][D] */, float* rewards /* [P][N] */,
                       uint8_t* done /* [N] */, uint8_t* truncated /* [N] */, float* fin_return /* [P][N] */,
                       int32_t* fin_length /* [N] */, int32_t auto_reset, float* final_obs /* [P][N][D] or NULL */,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-agent Q-networks.  All agents share one shape (true for every LBF task); parameters of
 * agent i are one contiguous fp32 block in torch parameters() order of
 * critic.independent.{i}.network.{0,2,4}.{weight,bias} (marlbase/utils/models.py:34-42,146-154):
 *     W1[H][D] b1[H] W2[H][H] b2[H] W3[A][H] b3[A]            (nn.Linear row-major)
 * so state_dict tensors are plain slices of the block (checkpoints: dqn/train.py:340-343).
 * ---------------------------------------------------------------------------------------- */
typedef struct marlhip_net_shape {
    int32_t n_agents; /* P */
    int32_t obs_dim;  /* D */
    int32_t hidden;   /* H: two hidden layers of this width (algorithm.model.layers=[H,H]) */
    int32_t n_actions;/* A */
    /* parameter sharing (MultiAgentSharedNetwork, marlbase/utils/models.py:176-300): n_networks = 0 means independent
     * networks (one block per agent); otherwise agent i evaluates block net_of[i] (0 <= net_of[i] < n_networks), every
     * `params` / `target` / `grad` argument is [n_networks][nparams] and a network's gradient is the sum over its agents. */
    int32_t n_networks;
    int32_t net_of[16];
    int32_t n_hidden; /* hidden layers: 0 = 2.  Every fused kernel implements two; 1..16 layers of width `hidden` run on the GEMM path
                       * (marlhip_wide_*, and the actor-critic entry points, which route such shapes there) */
} marlhip_net_shape;

int marlhip_net_nparams(const marlhip_net_shape* s); /* per network block; <0 if the shape has no kernel */

/* bytes of `workspace` the forward-only entry points below need for this network shape (the collectors, marlhip_ac_forward_rows,
 * marlhip_gru_forward, marlhip_gru_ac_forward: MFMA weight packs of every agent, built per call from the canonical parameters;
 * covers the one-output critic and the centralised-critic variants of the shape).  The learner entry points size their own
 * workspaces (marlhip_dqn_workspace_bytes, marlhip_ac_workspace_bytes, ...). */
int64_t marlhip_forward_workspace_bytes(const marlhip_net_shape* s);

/* QNetwork.act (marlbase/dqn/model.py:94-116), batched over N envs: Q = critic_i(obs_i);
 * ONE uniform per env decides random-vs-greedy for the whole joint action (model.py:105);
 * greedy = first index of the max (torch.argmax).  Noise: if u != NULL use u[n] and
 * rand_actions[P][N]; else Philox(seed; env n, episode[n], t = ep_length[n]) as the fused
 * collector does.  q_out ([P][N][A]) may be NULL. */
int marlhip_dqn_act(const marlhip_net_shape* s, const float* params /* [P][nparams] */, const float* obs /* [P][N][D] */,
                    int32_t n_envs, float epsilon, const float* u /* [N] or NULL */,
                    const int32_t* rand_actions /* [P][N] or NULL */, uint64_t seed, const uint32_t* episode /* [N] */,
                    const int32_t* ep_length /* [N] */, int32_t* actions /* [P][N] */, float* q_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Episode replay.  Replaces marlbase/dqn/train.py:19-124 (ReplayBuffer).  Storage is
 * EPISODE-major so a sampled episode is contiguous (the reference's time-major numpy arrays
 * make every sampled episode 26 strided 60-byte reads per agent):
 *     obs   f32 [cap][P][T+1][D]     act u8 [cap][P][T]     rew f32 [cap][P][T]
 *     done  u8  [cap][T+1]           filled u8 [cap][T]
 * ---------------------------------------------------------------------------------------- */
typedef struct marlhip_replay_shape {
    int32_t capacity; /* buffer_size, in episodes */
    int32_t n_agents, obs_dim, max_len; /* P, D, T */
} marlhip_replay_shape;

typedef struct marlhip_replay_buffers {
    float* obs;
    uint8_t* act;
    float* rew;
    uint8_t* done;
    uint8_t* filled;
} marlhip_replay_buffers;

/* ReplayBuffer.init_episode (train.py:65-71) for N envs: obs -> row t=0 of slot[n]. */
int marlhip_replay_init_episode(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb,
                                const int32_t* slot /* [N] */, const uint8_t* active /* [N] or NULL */,
                                const float* obs /* [P][N][D] */, int32_t n_envs, void* stream);

/* ReplayBuffer.add (train.py:73-89) for N envs: obs -> row t[n]+1, action/reward -> t[n],
 * done -> t[n]+1, filled[t[n]] = 1.  Like the reference it never clears older rows of a
 * re-used slot (stale tails survive a ring wrap). */
int marlhip_replay_add(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* slot,
                       const int32_t* t /* [N] */, const uint8_t* active, const float* obs, const int32_t* actions,
                       const float* rewards, const uint8_t* done, int32_t n_envs, void* stream);
/* marlhip_replay_add for one step of a modular collection round, with the bookkeeping around it: rows of the envs with alive[n] != 0 are
 * stored at step t (= their episode step: all envs of a round start together), the stored done flag is done | truncated unless
 * use_proper_termination (dqn/train.py:219-225), then alive[n] &= ~(done | truncated). */
int marlhip_replay_add_step(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* slot /* [N] */, int32_t t,
                            int32_t use_proper_termination, uint8_t* alive /* [N] in/out */, const float* obs /* [P][N][D] */,
                            const int32_t* actions /* [P][N] */, const float* rewards /* [P][N] */, const uint8_t* done, const uint8_t* truncated,
                            int32_t n_envs, void* stream);

/* ReplayBuffer.sample (train.py:94-124): gather B whole episodes into the reference's Batch
 * layout: obss f32 [P][T+1][B][D], actions i64 [P][T][B], rewards f32 [P][T][B],
 * dones f32 [T+1][B], filled f32 [T][B].  idx != NULL: use idx[b] (parity with injected
 * np.random.randint draws); idx == NULL: idx[b] = Philox(seed; stream 2, counter) mod-free
 * bounded draw in [0, length), also written to idx_out if non-NULL. */
int marlhip_replay_sample(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* idx,
                          int32_t batch, int32_t length, uint64_t seed, uint32_t counter, int32_t* idx_out,
                          float* obss, int64_t* actions, float* rewards, float* dones, float* filled, void* stream);

/* ------------------------------------------------------------------------------------------
 * Learner step.  Replaces QNetwork._compute_loss + update (marlbase/dqn/model.py:118-174):
 * critic/target forwards over whole episodes, Double-Q target, masked MSE summed over agents,
 * backward, global-norm clip, Adam, target update.
 * ---------------------------------------------------------------------------------------- */
typedef struct marlhip_batch {
    const float* obss;      /* [P][T+1][B][D] */
    const int64_t* actions; /* [P][T][B] */
    const float* rewards;   /* [P][T][B] */
    const float* dones;     /* [T+1][B] */
    const float* filled;    /* [T][B] */
    int32_t max_len, batch; /* T, B */
    /* optional strides in ELEMENTS, 0 = the dqn/train.py layout above.  The ac/train.py Batch (ac/train.py:36-49) keeps the
     * agents innermost: obss [T+1][B][P*D] -> obs_agent_stride = D, obs_row_stride = P*D; actions / rewards [T][B][P] ->
     * act_agent_stride = 1, act_row_stride = P.  (obs strides: every learner entry point; act strides: marlhip_ac_* only)
     * obs_agent_stride < 0: every agent reads the SAME rows (centralised critics: the whole P*D row is the input). */
    int64_t obs_agent_stride, obs_row_stride, act_agent_stride, act_row_stride;
    /* batch.action_mask (dqn/train.py:118-124; dqn/model.py:133-143): f32 [P][T+1][B][A], 1 = allowed, or NULL.  In the DQN
     * family the bootstrap of transition t reads the target (and, for Double-Q, the online) values of observation t+1 with
     * the disallowed actions at -1e8, exactly as the reference overwrites them.  For the actor-critic learners
     * (batch.action_masks, ac/train.py:53-63; ac/model.py:135-145) the layout is the Batch's [T+1][B][P][A] and the
     * logits become logits * mask + (1 - mask) * -1e8 before the Categorical.  No env on this path emits masks (LBF and
     * rware do not); they enter through the scalar-env python loops and the Batch arguments. */
    const float* action_mask;
} marlhip_batch;

/* bytes of scratch marlhip_dqn_loss_grad needs for this (shape, T, B): partial gradient records, MFMA weight packs, the mixer planes and
 * the hidden layers one pass leaves for the next (hidden 128: the critic's second layer, 4 H bytes per transition row; hidden 64 two-pass
 * forms - VDN, QMIX, standardised returns, 71-wide rows: both layers, 512 bytes per row) */
int64_t marlhip_dqn_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch);

/* loss (scalar, model.py:160-163) and its gradient w.r.t. the critic parameters, both written to
 * device memory: grad[P][nparams] (same layout as params), loss[0] = value, loss[1] = sum(filled).
 * mode: 0 = IDQN (per-agent targets, model.py:118-163), 1 = VDN (sum over agents, agent-0 reward,
 * model.py:224-269; runs as agent-forward -> sum-mixer kernel -> agent-backward).
 * double_q as cfg.double_q (model.py:138-145). */
int marlhip_dqn_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params,
                          const marlhip_batch* batch, float gamma, int32_t double_q, int32_t mode, void* workspace,
                          int64_t workspace_bytes, float* grad, float* loss, void* stream);

/* Same loss and gradient, but the B episodes are gathered INSIDE the kernel straight from the
 * episode-major replay (ReplayBuffer.sample + _compute_loss fused, train.py:94-124 + model.py:118-163):
 * no Batch is materialised, each sampled episode is read once.  idx != NULL: episodes idx[b];
 * idx == NULL: the same Philox draw marlhip_replay_sample makes for (seed, counter, length).
 * idx_out (may be NULL) records the indices used.  Bitwise identical to sample -> loss_grad. */
int marlhip_dqn_loss_grad_replay(const marlhip_net_shape* s, const float* params, const float* target_params,
                                 const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* idx,
                                 int32_t batch, int32_t length, uint64_t seed, uint32_t counter, int32_t* idx_out,
                                 float gamma, int32_t double_q, int32_t mode, void* workspace, int64_t workspace_bytes,
                                 float* grad, float* loss, void* stream);

/* clip_grad_norm_(critic.parameters(), max_norm) over ALL agents' parameters (model.py:169-170;
 * max_norm <= 0: no clipping), torch.optim.Adam single-tensor step (model.py:171; step = 1-based
 * update count, bias corrections computed in fp64 on the host side of this call), then the
 * target update (model.py:176-196): hard_update != 0 copies params -> target, else tau > 0
 * Polyak-averages, else target untouched.  grad_scale multiplies the gradient first (1/world
 * for an all-reduced SUM).  gnorm_out[0] (may be NULL) receives the pre-clip total norm. */
int marlhip_dqn_clip_adam(int64_t n, float* params, const float* grad, float* exp_avg, float* exp_avg_sq,
                          float* target_params, int64_t step, double lr, double beta1, double beta2, double eps,
                          float max_norm, float grad_scale, int32_t hard_update, float tau,
                          float* scratch /* >= ceil(n/256) floats */, float* gnorm_out, void* stream);

/* the same step for the other optimisers `getattr(optim, cfg.optimizer)(params, lr=cfg.lr)` can name (dqn/model.py:66-71, ac/model.py:
 * 103-105), with torch's default hyper-parameters: optimizer 0 = Adam (as marlhip_dqn_clip_adam with betas (0.9, 0.999), eps 1e-8),
 * 1 = SGD, 2 = RMSprop (alpha 0.99, eps 1e-8; state2 = square_avg), 3 = AdamW (weight_decay 1e-2).  state1 / state2: the optimiser's
 * two [n] state slots (unused ones still have to be valid memory). */
int marlhip_dqn_clip_step(int32_t optimizer, int64_t n, float* params, const float* grad, float* state1, float* state2,
                          float* target_params, int64_t step, double lr, float max_norm, float grad_scale, int32_t hard_update, float tau,
                          float* scratch /* >= ceil(n/256) floats */, float* gnorm_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused collector.  Replaces _collect_trajectory (marlbase/dqn/train.py:202-237) for N envs in
 * ONE launch per round: reset -> T x (act -> step -> replay add), env state in registers, the
 * critics in LDS, transitions streamed to the replay slots slot_base+n (mod capacity).
 * Envs whose episode ends early idle for the rest of the round.  Outputs per env: episode
 * length and RAW per-agent returns (the reference's info["episode_returns"]).
 * write_replay == 0 gives _evaluate (train.py:177-199).  clear_stale != 0 zeroes filled[t]
 * beyond the episode end (the reference does not, a7).
 * ---------------------------------------------------------------------------------------- */
int marlhip_idqn_collect(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const float* params, float epsilon,
                         uint32_t round, const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb,
                         int32_t slot_base, int32_t write_replay, int32_t clear_stale, int32_t use_proper_termination,
                         float* fin_return /* [P][N] */, int32_t* fin_length /* [N] */, void* workspace,
                         int64_t workspace_bytes /* >= marlhip_forward_workspace_bytes(s) */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused actor-critic rollout collector.  Replaces _collect_trajectories (marlbase/ac/train.py:24-119)
 * for N vector envs in one launch: envs.reset() -> while running.any(): A2CNetwork.act (actor MLP ->
 * Categorical sample, ac/model.py:147-153) -> envs.step (auto-reset) -> masked writes of the running envs.
 * actor_params: [P][nparams] blocks of actor.independent.{i}.network.* (same layout as the Q-networks).
 * Outputs in the reference's time-major Batch layout (ac/train.py:36-49), fully written (zeros where the
 * reference leaves its fresh zeros): batch_obs f32 [T+1][N][P*D] (agents concatenated), batch_act i64
 * [T][N][P], batch_rew f32 [T][N][P], batch_done u8 [T+1][N], batch_filled f32 [T][N]; per-env episode
 * statistics as info["final_info"][i] carries them; t_max[0] = the reference's returned `t`.
 * Sampling = inverse CDF of the fp32 softmax with one Philox uniform per (env, step, agent); reset stream
 * index 2*round, auto-reset observation from 2*round+1.
 * ---------------------------------------------------------------------------------------- */
int marlhip_ac_collect(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const float* actor_params,
                       uint32_t round, int32_t max_len, int32_t use_proper_termination, float* batch_obs,
                       int64_t* batch_act, float* batch_rew, uint8_t* batch_done, float* batch_filled,
                       float* fin_return /* [P][N] */, int32_t* fin_length /* [N] */, int32_t* t_max /* [1] */,
                       void* workspace, int64_t workspace_bytes /* >= marlhip_forward_workspace_bytes(s) */, void* stream);

/* The second pass of a rollout.  The reference keeps stepping the envs whose episode has ended (auto-reset vector env; the policy acts
 * on them too) until the LAST env's first episode ends and appends the `final_info` of every further episode finishing meanwhile to the
 * rollout's infos (ac/train.py:71,101-110); with env.standardise_rewards those steps also move the envs' running reward statistics.
 * marlhip_ac_collect ignores an env once its first episode is over; this call runs the listed envs again - from the auto-reset state
 * (reset stream 2 * round + 1, then + 2, ...), from step t_start[i] (= the length of env_ids[i]'s first episode) to t_stop (= the
 * first pass's t_max), with the first pass's action noise - writes no batch and leaves, per listed env i, cnt[i] <= cap records:
 * ret[i][k][P] episode returns, meta[i][k] = (episode length, finishing step).  Same collector kernels, same workspace. */
int marlhip_ac_collect_later_episodes(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const float* actor_params, uint32_t round,
                                      int32_t max_len, const int32_t* env_ids /* [n_envs] */, const int32_t* t_start /* [n_envs] */,
                                      int32_t n_envs, int32_t t_stop, int32_t cap, float* ret /* [n_envs][cap][P] */,
                                      int32_t* meta /* [n_envs][cap][2] */, int32_t* cnt /* [n_envs] */, void* workspace,
                                      int64_t workspace_bytes /* >= marlhip_forward_workspace_bytes(s) */, void* stream);

/* One step's bookkeeping of a MODULAR rollout (recurrent actors, actors on the GEMM path: forward, sampling and env step are their own
 * calls): ac/train.py:90-110 for all envs - masked writes of the still-running envs into row t / t + 1 of the time-major batch, the first
 * episode's statistics from the env's fin_return / fin_length, a record (finishing step, env, length | returns) for every LATER episode
 * that finishes (later_count may be NULL: no records; the count may exceed later_cap, the records do not), then running &= ~finished.
 * `obs` / `rewards` / `done` / `truncated` / `env_fin_*` are the env-step call's outputs, `actions` the sampled actions. */
int marlhip_ac_store_step(int32_t n_envs, int32_t n_agents, int32_t obs_dim, int32_t t, int32_t use_proper_termination, uint8_t* running /* [N] in/out */,
                          const float* obs /* [P][N][D] */, const int64_t* actions /* [P][N] */, const float* rewards /* [P][N] */,
                          const uint8_t* done, const uint8_t* truncated, const float* env_fin_return /* [P][N] */, const int32_t* env_fin_length,
                          float* batch_obs_t1 /* [N][P*D] */, int64_t* batch_act_t /* [N][P] */, float* batch_rew_t /* [N][P] */,
                          uint8_t* batch_done_t1 /* [N] */, float* batch_filled_t /* [N] */, float* fin_return /* [P][N] */, int32_t* fin_length,
                          int32_t* later_count, int32_t later_cap, float* later_returns /* [cap][P] */, int32_t* later_meta /* [cap][3] */,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Recurrent Q-networks (`use_rnn: True`; RNNNetwork, marlbase/utils/models.py:51-116): Linear(D, H) -> ReLU -> one-layer
 * nn.GRU(H, H) -> Linear(H, A), compiled for hidden 64.  One agent's block in parameters() order:
 *   first_layer.weight [H][D] | .bias [H] | rnn.weight_ih_l0 [3H][H] | rnn.weight_hh_l0 [3H][H] | rnn.bias_ih_l0 [3H] |
 *   rnn.bias_hh_l0 [3H] | final_layer.weight [A][H] | .bias [A]                       (gate order r, z, n)
 * marlhip_gru_forward: q[p][t][b][:] for t = 0..steps-1 of the sequences obs [P][steps][B][D], from the hidden state h_in
 * [P][B][H] (NULL = zeros: `hiddens=None`, dqn/model.py:127,133), final hidden state to h_out (NULL = not wanted).
 * steps = 1 is QNetwork.act's forward (dqn/model.py:99) with the caller carrying the hidden state; steps = T+1 is the
 * learner's.  record (NULL or marlhip_gru_record_floats floats) receives the per-step activations the backward pass reads.
 * ---------------------------------------------------------------------------------------- */
int marlhip_gru_nparams(const marlhip_net_shape* s); /* per agent block; <0 if the shape has no recurrent kernel */
/* C-ABI 218 - stacked recurrent layers (RNNNetwork: nn.GRU(num_layers = len(layers) - 1), marlbase/utils/models.py:74-90): every
 * marlhip_gru_* entry point reads the depth from marlhip_net_shape.n_hidden = len(layers) (0 = 2 = one GRU layer; 2..5 -> 1..4 layers of
 * width `hidden`).  The block is RNNNetwork's parameters() order: first_layer, then (weight_ih, weight_hh, bias_ih, bias_hh) of layer 0, 1,
 * ..., then final_layer.  Hidden states across calls (h_in / h_out) are [L][P][B][H] - nn.GRU's (num_layers, batch, hidden) per agent;
 * records are L times marlhip_gru_record_floats' one-layer size (the function returns the stack's).  The forward-only entry points
 * (marlhip_gru_forward, marlhip_gru_ac_forward) chain the layers through scratch behind the weight packs: their `workspace` holds
 * marlhip_gru_forward_workspace_bytes(s, steps, batch) bytes (for one layer: marlhip_forward_workspace_bytes(s) suffices, as before).
 * marlhip_gru_a2c_loss_grad / marlhip_gru_ppo_*: the critics' depth from marlhip_ac_config.critic_n_hidden when it differs (C-ABI 219);
 * marlhip_mixed_*: the recurrent family's depth as documented there. */
int64_t marlhip_gru_forward_workspace_bytes(const marlhip_net_shape* s, int32_t steps, int32_t batch);
int64_t marlhip_gru_record_floats(const marlhip_net_shape* s, int32_t steps, int32_t batch);
int marlhip_gru_forward(const marlhip_net_shape* s, const float* params /* [P][nparams] */, const float* obs, int32_t steps,
                        int32_t batch, const float* h_in, float* h_out, float* q_out /* [P][steps][B][A] */, float* record,
                        void* workspace, int64_t workspace_bytes /* >= marlhip_forward_workspace_bytes(s) */, void* stream);

/* QNetwork._compute_loss / VDNetwork._compute_loss + loss.backward() with recurrent networks (mode 0 / 1): sequence forward of
 * the critic (activations recorded) and the target from zero hidden states, TD rows (Double-Q or max, action masks honoured),
 * back-propagation through time, weight-gradient sums, deterministic record reduce.  grad [P][marlhip_gru_nparams], loss[2] =
 * (loss, sum(filled)) as marlhip_dqn_loss_grad; afterwards marlhip_dqn_clip_adam as for the feed-forward networks. */
int64_t marlhip_gru_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch);
int marlhip_gru_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* batch,
                          float gamma, int32_t double_q, int32_t mode, void* workspace, int64_t workspace_bytes, float* grad,
                          float* loss /* [2] */, void* stream);

/* the recurrent independent learner with standardise_returns (dqn/model.py:146-158; statistics as marlhip_dqn_loss_grad_std) */
int marlhip_gru_loss_grad_std(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* batch,
                              float gamma, int32_t double_q, const struct marlhip_ret_stats* stats, void* workspace,
                              int64_t workspace_bytes, float* grad, float* loss /* [2] */, void* stream);

/* QMixNetwork._compute_loss + backward with recurrent agent networks: the sequence kernels above for the agents, the mixer stage
 * of marlhip_qmix_loss_grad (same kernels, same mixer block layout) between them.  grad: agents' blocks; mixer->mixer_grad: the
 * mixer's; loss[2] as everywhere. */
int64_t marlhip_gru_qmix_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch);
/* C-ABI 212: for the mixing configuration in mixer->{embed_dim, hypernet_layers, hypernet_embed} (marlhip_qmix_workspace_bytes_mx) */
int64_t marlhip_gru_qmix_workspace_bytes_mx(const marlhip_net_shape* s, const struct marlhip_qmix_mixer* mixer, int32_t max_len, int32_t batch);
int marlhip_gru_qmix_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params,
                               const struct marlhip_qmix_mixer* mixer, const marlhip_batch* batch, float gamma, int32_t double_q,
                               void* workspace, int64_t workspace_bytes, float* grad, float* loss /* [2] */, void* stream);
/* marlhip_qmix_loss_grad with agent networks on the GEMM path (marlhip_wide_*: layers wider than 128 or not two deep); the mixer stage is
 * the same one, mixer->ret_stats (columns = batch) included */
int64_t marlhip_wide_qmix_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch);
int64_t marlhip_wide_qmix_workspace_bytes_mx(const marlhip_net_shape* s, const struct marlhip_qmix_mixer* mixer, int32_t max_len, int32_t batch);
int marlhip_wide_qmix_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params, const struct marlhip_qmix_mixer* mixer,
                                const marlhip_batch* batch, float gamma, int32_t double_q, void* workspace, int64_t workspace_bytes,
                                float* grad, float* loss, void* stream);


/* Actor-critic learner step with recurrent networks (`use_rnn: True` for actor and critic; ia2c.yaml / ippo.yaml): same contracts as
 * marlhip_a2c_loss_grad / marlhip_ppo_prepare / marlhip_ppo_loss_grad, the blocks in the recurrent layout (marlhip_gru_nparams for the
 * actors, marlhip_gru_ac_critic_nparams for the critics; cfg->centralised_critic as there).  marlhip_gru_ac_forward: sequence forward of
 * the actors (value_net = 0), critics (1) or centralised critics (2, agent_stride 0) with the hidden state carried by the caller (A2CNetwork.act / get_value). */
int marlhip_gru_ac_critic_nparams(const marlhip_net_shape* s, int32_t centralised);
int64_t marlhip_gru_ac_workspace_bytes(const marlhip_net_shape* s, int32_t centralised, int32_t max_len, int32_t batch);
/* C-ABI 219: recurrent critics of another depth than the recurrent actors (actor.layers / critic.layers are separate lists: marlbase/ac/model.py:45-97).
 * s->n_hidden = len(actor.layers); marlhip_ac_config.critic_n_hidden = len(critic.layers) (0: as the actors; 2..5), the critic blocks in
 * marlhip_gru_ac_critic_nparams of a shape whose n_hidden is the critics', marlhip_gru_ac_forward(value_net != 0) with that shape too. */
int64_t marlhip_gru_ac_workspace_bytes_lc(const marlhip_net_shape* s, int32_t centralised, int32_t critic_n_hidden, int32_t max_len, int32_t batch);
int marlhip_gru_a2c_loss_grad(const marlhip_net_shape* s, const float* actor, const float* critic, const float* target_critic,
                              const marlhip_batch* batch, const struct marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes,
                              float* actor_grad, float* critic_grad, float* metrics /* [5] */, void* stream);
int marlhip_gru_ppo_prepare(const marlhip_net_shape* s, const float* actor, const float* critic, const float* target_critic,
                            const marlhip_batch* batch, const struct marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes,
                            void* stream);
int marlhip_gru_ppo_loss_grad(const marlhip_net_shape* s, const float* actor, const float* critic, const marlhip_batch* batch,
                              const struct marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes, float* actor_grad,
                              float* critic_grad, float* metrics /* [5] */, void* stream);
int marlhip_gru_ac_forward(const marlhip_net_shape* s, int32_t value_net, const float* params, const float* obs, int64_t agent_stride,
                           int64_t row_stride, int32_t steps, int32_t batch, const float* h_in, float* h_out, float* out, void* workspace,
                           int64_t workspace_bytes /* >= marlhip_forward_workspace_bytes(s) */, void* stream);

/* actor.use_rnn != critic.use_rnn (C-ABI 217).  The reference builds the two families from their own flags (marlbase/ac/model.py:45-97), so a
 * recurrent actor next to feed-forward critics - or the reverse - is a configuration A2CNetwork / PPONetwork accept.  Same contracts as
 * marlhip_a2c_loss_grad / marlhip_ppo_prepare / marlhip_ppo_loss_grad with each block in ITS family's layout: actor_rnn != 0 - recurrent actors
 * (marlhip_gru_nparams) + feed-forward critics (marlhip_ac_critic_nparams(s, 0)); actor_rnn == 0 - feed-forward actors (marlhip_net_nparams) +
 * recurrent critics (marlhip_gru_ac_critic_nparams(s, 0)).  Independent critics; LBF observation widths (csrc/mixed_ac.hip: MARL_MIXED_AC_SHAPES);
 * cfg->actor_forward_kept / defer_critic_backward must be 0.  Forward passes for acting / values: the family's own marlhip_gru_ac_forward or
 * marlhip_ac_forward_rows. */
int64_t marlhip_mixed_ac_workspace_bytes(const marlhip_net_shape* s, int32_t actor_rnn, int32_t max_len, int32_t batch);
/* C-ABI 219: the recurrent family as a stack of GRU layers - recurrent actors: s->n_hidden = len(actor.layers) (2..5; the critics' two layers
 * stay implied, cfg->critic_n_hidden 0 or 2); recurrent critics: cfg->critic_n_hidden = len(critic.layers) (s->n_hidden = the actors' 2) and
 * the workspace from this query */
int64_t marlhip_mixed_ac_workspace_bytes_lc(const marlhip_net_shape* s, int32_t actor_rnn, int32_t critic_n_hidden, int32_t max_len, int32_t batch);
int marlhip_mixed_a2c_loss_grad(const marlhip_net_shape* s, int32_t actor_rnn, const float* actor, const float* critic, const float* target_critic,
                                const marlhip_batch* batch, const struct marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes,
                                float* actor_grad, float* critic_grad, float* metrics /* [5] */, void* stream);
int marlhip_mixed_ppo_prepare(const marlhip_net_shape* s, int32_t actor_rnn, const float* actor, const float* critic, const float* target_critic,
                              const marlhip_batch* batch, const struct marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes,
                              void* stream);
int marlhip_mixed_ppo_loss_grad(const marlhip_net_shape* s, int32_t actor_rnn, const float* actor, const float* critic, const marlhip_batch* batch,
                                const struct marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes, float* actor_grad,
                                float* critic_grad, float* metrics /* [5] */, void* stream);

/* Categorical(logits=logits[p][n]).sample() for every (agent, env) (ac/model.py:147-153), drawn as the fused rollout collector draws
 * it: inverse CDF of the fp32 softmax with the Philox uniform of (env n, episode[n], step t, word 1 + p).  actions: i64 [P][N]. */
int marlhip_sample_from_logits(int32_t n_agents, int32_t n_envs, int32_t n_actions, const float* logits /* [P][N][A] */, uint64_t seed,
                               const uint32_t* episode /* [N] */, int32_t t, int64_t* actions, void* stream);

/* ------------------------------------------------------------------------------------------
 * Networks without a fused kernel: two hidden layers of ANY width (FCNetwork takes any list, marlbase/utils/models.py:14-48;
 * hidden > 128 - e.g. layers [256, 256] - or an observation / action width outside the compiled lists).  The three layers run as
 * f32 MFMA GEMMs over all rows with the activations in HBM (csrc/wide_mlp.h): slower than the fused kernels, any size.  Parameters
 * in FCNetwork's parameters() order per block; net shape as everywhere (hidden = the width of all n_hidden layers, 1..16 of them;
 * unequal widths are zero-padded by the caller, which is exact).  marlhip_wide_forward serves QNetwork.act / get_value and the modular collectors
 * (-> marlhip_act_from_q / marlhip_sample_from_logits); marlhip_wide_dqn_loss_grad is marlhip_dqn_loss_grad for such networks
 * (mode 0 IDQN, 1 VDN; same batch, outputs and 1 / sum(filled) normalisation; marlhip_dqn_clip_adam applies it).  The actor-critic
 * entry points (marlhip_a2c_loss_grad, marlhip_ppo_*, marlhip_ac_forward_rows, marlhip_ac_critic_nparams, marlhip_ac_workspace_bytes)
 * take such shapes directly: a shape without fused kernels runs actors and critics on this path. */
int marlhip_wide_nparams(const marlhip_net_shape* s, int32_t n_out /* outputs: n_actions, or 1 for a critic */);
int64_t marlhip_wide_forward_workspace_bytes(const marlhip_net_shape* s, int32_t n_rows);
int marlhip_wide_forward(const marlhip_net_shape* s, int32_t n_out, const float* params /* [blocks][marlhip_wide_nparams] */,
                         const float* obs /* agent p, row r at obs + p * agent_stride + r * row_stride */, int64_t agent_stride,
                         int64_t row_stride, int32_t n_rows, float* out /* [P][n_rows][n_out] */, void* workspace,
                         int64_t workspace_bytes /* >= marlhip_wide_forward_workspace_bytes(s, n_rows) */, void* stream);
int64_t marlhip_wide_dqn_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch);
int marlhip_wide_dqn_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* batch,
                               float gamma, int32_t double_q, int32_t mode, void* workspace, int64_t workspace_bytes, float* grad,
                               float* loss /* [2]: loss, sum(filled) */, void* stream);
/* the same with standardise_returns (marlhip_dqn_loss_grad_std's contract: stats->columns = 0 -> per-agent statistics, the independent
 * learner; columns = batch -> VDNetwork's per-batch-column statistics); the statistics are updated in place */
int marlhip_wide_dqn_loss_grad_std(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* batch,
                                   float gamma, int32_t double_q, const struct marlhip_ret_stats* stats, void* workspace,
                                   int64_t workspace_bytes, float* grad, float* loss /* [2] */, void* stream);

/* the action choice of QNetwork.act (dqn/model.py:105-115) from given values q [P][N][A] (the recurrent path computes them with
 * marlhip_gru_forward): explore iff epsilon > u with ONE Philox uniform per env (env n, episode[n], t = ep_length[n], word 0),
 * random action of agent p = word 1+p, greedy = first maximum - the same words marlhip_dqn_act and the fused collector use. */
int marlhip_act_from_q(int32_t n_agents, int32_t n_envs, int32_t n_actions, const float* q, float epsilon, uint64_t seed,
                       const uint32_t* episode /* [N] */, const int32_t* ep_length /* [N] */, int32_t* actions /* [P][N] */,
                       void* stream);

/* the two fused collectors on the warehouse env (same contracts; net shape D = 71, A = 5; compiled for the tiny layouts,
 * 2 and 4 agents: the shelf layer of a workgroup's 64 envs lives in LDS behind the weight packs) */
int marlhip_rware_idqn_collect(const marlhip_rware_config* cfg, const marlhip_net_shape* s, const float* params, float epsilon,
                               uint32_t round, const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb,
                               int32_t slot_base, int32_t write_replay, int32_t clear_stale, int32_t use_proper_termination,
                               float* fin_return /* [P][N] */, int32_t* fin_length /* [N] */, void* workspace, int64_t workspace_bytes,
                               void* stream);
int marlhip_rware_ac_collect(const marlhip_rware_config* cfg, const marlhip_net_shape* s, const float* actor_params,
                             uint32_t round, int32_t max_len, int32_t use_proper_termination, float* batch_obs,
                             int64_t* batch_act, float* batch_rew, uint8_t* batch_done, float* batch_filled,
                             float* fin_return /* [P][N] */, int32_t* fin_length /* [N] */, int32_t* t_max /* [1] */,
                             void* workspace, int64_t workspace_bytes, void* stream);
int marlhip_rware_ac_collect_later_episodes(const marlhip_rware_config* cfg, const marlhip_net_shape* s, const float* actor_params, uint32_t round,
                                            int32_t max_len, const int32_t* env_ids, const int32_t* t_start, int32_t n_envs, int32_t t_stop,
                                            int32_t cap, float* ret, int32_t* meta, int32_t* cnt, void* workspace, int64_t workspace_bytes,
                                            void* stream); /* marlhip_ac_collect_later_episodes on the warehouse */

/* ------------------------------------------------------------------------------------------
 * Actor-critic learner step (IA2C / IPPO).  Replaces A2CNetwork.update / PPONetwork.update
 * (marlbase/ac/model.py:189-246, 264-352) up to loss.backward(): target-critic values of all T+1 observations,
 * n-step returns (marlbase/utils/utils.py:38-63), critic values and Categorical log-prob / entropy on obs[:-1],
 * filled-masked actor + value losses, gradient w.r.t. actor and critic.  Independent networks, no RNN / masks /
 * return standardisation.  The batch is the ac/train.py Batch (set the marlhip_batch strides; dones as fp32).
 * actor: [P][marlhip_net_nparams(s)] blocks of actor.independent.{i}.network.*; critic / target_critic:
 * [P][marlhip_ac_critic_nparams(s)] blocks (same MLP with ONE output).  Afterwards: clip_grad_norm_(self.parameters())
 * + Adam = ONE marlhip_dqn_clip_adam over the contiguous [actor | critic] block (target = NULL), then the
 * step-keyed target copy (model.py:233-239) on the critic block.
 * metrics[5] = loss, actor_loss, value_loss, entropy (model.py:241-246), sum(filled).
 * ---------------------------------------------------------------------------------------- */
typedef struct marlhip_ac_config {
    int32_t n_steps;      /* cfg.n_steps (1..16) */
    float entropy_coef, value_loss_coef;
    float ppo_clip;       /* PPO only */
    double gamma;         /* python float: gamma ** k is formed in fp64 and rounded once, like the reference */
    /* cfg.standardise_returns (model.py:195-204): device statistics [P], [P], [1] as in marlhip_ret_stats, or all NULL.
     * A2C: updated inside marlhip_a2c_loss_grad; PPO: inside marlhip_ppo_prepare (the epochs reuse them). */
    float* ret_mean;
    float* ret_var;
    double* ret_count;
    int32_t centralised_critic; /* critic.centralised (MAA2C / MAPPO, model.py:62-66,155-157): every agent's critic and target
                                   critic takes the concatenation of ALL agents' observations (P*D inputs); compiled for
                                   hidden 128 up to 4 agents and hidden 64 for 2 agents */
    void* side_stream;          /* hipStream_t of the caller on the same device, or NULL.  Recurrent actors + critics only: the
                                   critics' sequence passes are enqueued on it next to the actors' on the call's stream (fork /
                                   join through events inside the call; everything is joined back before the call returns).
                                   NULL: one stream, no overlap, same results. */
    /* data-parallel training with standardise_returns (C-ABI 208): the batch moments summed over the ranks before the running
     * statistics move - the marlhip_ret_stats fields of the same names (declared further down); all NULL: single process */
    int (*ret_exchange)(void* ctx, double* buf, int64_t count, void* stream);
    void* ret_exchange_ctx;
    double* ret_moments;        /* [2P + 1] device scratch */
    /* C-ABI 212: critic.parameter_sharing different from actor.parameter_sharing (ac/model.py:45-97 builds the two families from their own
     * settings).  critic_n_networks > 0: the critics (and target critics) use THIS agent -> network map - critic / target_critic / critic_grad
     * are then [critic_n_networks][n] blocks - while the actors keep marlhip_net_shape's; 0: one map for both (the default). */
    int32_t critic_n_networks;
    int32_t critic_net_of[16];
    /* C-ABI 213: the rollout was collected by marlhip_ac_collect_keep / marlhip_rware_ac_collect_keep into THIS call's workspace with THESE
     * actor parameters - the actors' logits and hidden layers of every batch row are already there and the call does not compute them
     * again.  For marlhip_a2c_loss_grad; for marlhip_ppo_prepare (the old log-probs) and the FIRST marlhip_ppo_loss_grad of a rollout
     * (the later epochs run on moved parameters).  0 (default): the call runs the actors' forward pass itself. */
    int32_t actor_forward_kept;
    /* C-ABI 214, marlhip_a2c_loss_grad with feed-forward networks: the critics' backward pass is enqueued on `side_stream` (required, not the
     * call's stream) behind the actors' and is NOT joined before the call returns.  Without a joint gradient clip (ia2c.yaml / maa2c.yaml:
     * grad_clip False) the optimiser step is elementwise, so the caller can step the ACTOR block on the call's stream and start the next
     * rollout while the critics' gradient, their step and their target update run on side_stream; the next call on this workspace (and any
     * reader of the critic blocks) must be ordered behind that work by the caller.  0 (default): both passes on the call's stream. */
    int32_t defer_critic_backward;
    /* C-ABI 215: the critics' number of hidden layers when it differs from the actors' (marlhip_net_shape.n_hidden) - ac/model.py:45-97
     * builds the two families from their own `layers` lists.  GEMM-path shapes only (both networks then run there, zero-padded to the
     * shape's width); critic / target_critic blocks are WideNet blocks of that depth (marlhip_ac_critic_nparams on a shape copy whose
     * n_hidden is the critics'), the workspace comes from marlhip_ac_workspace_bytes_lc.  0 (default): as the actors. */
    int32_t critic_n_hidden;
} marlhip_ac_config;

int marlhip_ac_critic_nparams(const marlhip_net_shape* s, int32_t centralised); /* per critic block */
int64_t marlhip_ac_workspace_bytes(const marlhip_net_shape* s, int32_t centralised, int32_t max_len, int32_t batch);
int64_t marlhip_ac_workspace_bytes_lc(const marlhip_net_shape* s, int32_t centralised, int32_t critic_n_hidden /* marlhip_ac_config's */,
                                      int32_t max_len, int32_t batch);
/* A2CNetwork.get_value / the actor forward of A2CNetwork.act (model.py:147-163) on arbitrary rows:
 * out[p][row][:] = MLP_p(obs + p * agent_stride + row * row_stride); value_net 1: the one-output critic shape; value_net 2:
 * the centralised critic (P*D inputs, agent_stride 0: rows are the concatenated observations). */
int marlhip_ac_forward_rows(const marlhip_net_shape* s, int32_t value_net, const float* params, const float* obs,
                            int64_t agent_stride, int64_t row_stride, int32_t n_rows, float* out, void* workspace,
                            int64_t workspace_bytes /* >= marlhip_forward_workspace_bytes(s) */, void* stream);
int marlhip_a2c_loss_grad(const marlhip_net_shape* s, const float* actor, const float* critic, const float* target_critic,
                          const marlhip_batch* batch, const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes,
                          float* actor_grad, float* critic_grad, float* metrics, void* stream);
/* marlhip_ac_collect / marlhip_rware_ac_collect that also KEEP the actors' forward pass for the learner step (C-ABI 213).  A2C updates once
 * per rollout on the parameters the rollout was sampled with (ac/train.py:203-212, ac/model.py:189-246), so the logits and hidden layers
 * A2CNetwork.update recomputes for every batch row (model.py:206-213) are the values the collector held when it sampled that row's action:
 * the same packs and operand order, the same bits.  The collector writes them into `learner_workspace` (the marlhip_a2c_loss_grad
 * workspace for max_len x n_envs rows, >= marlhip_ac_workspace_bytes(s, centralised, max_len, n_envs)) where the step reads them; the
 * following marlhip_a2c_loss_grad (or marlhip_ppo_prepare + first marlhip_ppo_loss_grad) on that workspace sets
 * marlhip_ac_config.actor_forward_kept.  Fused feed-forward actors and n_envs % 16
 * == 0 only (an error otherwise); rows of envs whose episode is over hold zeros (their gradients are masked by `filled`). */
/* A stream restricted to a share of the device's compute units (hipExtStreamCreateWithCUMask; C-ABI 214), for work that is meant to run NEXT
 * TO another kernel rather than in front of it: a kernel whose grid fills the chip holds every compute unit until it retires, so the
 * critics' backward pass of marlhip_ac_config.defer_critic_backward on an ordinary stream delays the following rollout by its whole length;
 * on a stream that owns `percent` of the units it leaves the rest to the rollout (same grid, same summation order, same bits).
 * pattern 0: the lowest-numbered bits of the runtime's compute-unit mask, 1: every other bit.  WHICH physical units a mask bit names is
 * device-specific: on multi-XCD parts (MI355X: 8 XCDs x 32 units) the runtime's bit order interleaves shader engines and XCDs, so "the
 * lowest half" may be whole XCDs or an uneven share per shader engine - that moves the measured overlap gain, never a result.  Callers
 * that report timings under such a stream should record (percent, pattern, unit count) next to them (bench.py: `side_stream`).
 * marlhip_stream_destroy releases the stream. */
int marlhip_stream_create_cu_share(int32_t percent /* 1..100 */, int32_t pattern, void** stream_out);
int marlhip_stream_destroy(void* stream);
int marlhip_ac_collect_keep(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const float* actor_params, uint32_t round,
                            int32_t max_len, int32_t use_proper_termination, float* batch_obs, int64_t* batch_act, float* batch_rew,
                            uint8_t* batch_done, float* batch_filled, float* fin_return, int32_t* fin_length, int32_t* t_max,
                            void* workspace, int64_t workspace_bytes, int32_t centralised, void* learner_workspace,
                            int64_t learner_workspace_bytes, void* stream);
int marlhip_rware_ac_collect_keep(const marlhip_rware_config* cfg, const marlhip_net_shape* s, const float* actor_params, uint32_t round,
                                  int32_t max_len, int32_t use_proper_termination, float* batch_obs, int64_t* batch_act, float* batch_rew,
                                  uint8_t* batch_done, float* batch_filled, float* fin_return, int32_t* fin_length, int32_t* t_max,
                                  void* workspace, int64_t workspace_bytes, int32_t centralised, void* learner_workspace,
                                  int64_t learner_workspace_bytes, void* stream);
/* PPONetwork.update: marlhip_ppo_prepare once per batch (returns + old log-probs, kept in the workspace: model.py:266-293),
 * then per epoch marlhip_ppo_loss_grad + marlhip_dqn_clip_adam (model.py:296-335). */
int marlhip_ppo_prepare(const marlhip_net_shape* s, const float* actor, const float* critic, const float* target_critic,
                        const marlhip_batch* batch, const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes,
                        void* stream);
int marlhip_ppo_loss_grad(const marlhip_net_shape* s, const float* actor, const float* critic, const marlhip_batch* batch,
                          const marlhip_ac_config* cfg, void* workspace, int64_t workspace_bytes, float* actor_grad,
                          float* critic_grad, float* metrics, void* stream);

/* ------------------------------------------------------------------------------------------
 * n learner updates from one call: n x (marlhip_replay_sample with device-drawn indices ->
 * marlhip_dqn_loss_grad -> marlhip_dqn_clip_adam), with QNetwork.update's bookkeeping
 * (marlbase/dqn/model.py:165-185): *updates += 1 per update, hard target copy when
 * updates - last_target_update >= target_update_interval_or_tau (> 1), Polyak when < 1.
 * Single-GPU convenience (no hook for a gradient all-reduce): identical arithmetic, fewer host calls.  For the hidden-64
 * IDQN / VDN learners an update is 3 launches here instead of 4: the MFMA weight packs are built once per call and kept current
 * by the Adam launch of every update (which also takes the clip norm from partial sums the gradient reduce leaves behind).
 * ---------------------------------------------------------------------------------------- */
typedef struct marlhip_idqn_learner {
    marlhip_net_shape net;
    marlhip_replay_shape rs;
    marlhip_replay_buffers rb;
    float *params, *target, *exp_avg, *exp_avg_sq, *grad, *loss;
    float* scratch; /* >= ceil(n / 64) floats, n = all parameters of all networks (clip-norm partial sums) */
    float* gnorm;
    void* workspace;
    int64_t workspace_bytes;
    float* obss;      /* sample outputs, shaped for `batch` episodes (marlhip_replay_sample) */
    int64_t* actions;
    float* rewards;
    float* dones;
    float* filled;
    int32_t* idx;
    int32_t batch;
    int32_t double_q, mode;
    int32_t materialise_batch; /* 0: gather in the loss/grad kernel (marlhip_dqn_loss_grad_replay); 1: sample -> Batch -> loss_grad */
    float gamma, max_norm;
    double lr, beta1, beta2, eps;
    double target_update_interval_or_tau;
} marlhip_idqn_learner;

int marlhip_idqn_update_n(const marlhip_idqn_learner* L, int32_t n_updates, int32_t length, uint64_t seed,
                          uint32_t counter0, int64_t* adam_step, int64_t* updates, int64_t* last_target_update,
                          void* stream);

/* Filled-aware learner updates (C-ABI 216).  The loss is a filled-weighted sum (dqn/model.py:160-163): the rows behind an episode's last
 * transition contribute exactly zero to it and to every gradient entry.  For the single-pass hidden-64 IDQN learner marlhip_idqn_update_n
 * therefore plans the walk before its update loop (one launch for all n updates; csrc/update_plan.h): the batch's episode draws - the
 * same Philox stream as marlhip_replay_sample - are ordered by stored length, longest first (a stable sort: a batch of full-length
 * episodes keeps its draw order and the update its bits), a 16-episode tile walks only the steps its longest episode has, in chunks
 * sized so that the launch's waves get the same number of steps.  Any permutation of the batch is the same update up to summation
 * order (deterministic).  marlhip_idqn_learner.idx still receives the LAST update's draws in draw order.  Limits: 256 <= batch <= 8192,
 * max_len <= 255 (otherwise, and with MARLHIP_NO_PLAN=1 in the environment, every tile walks all max_len steps as before).
 * marlhip_update_plan exposes the plans for inspection: dims_out[8] = {planned, int32 per update, header ints, waves per agent, table
 * rows, chunks per tile of the static plan, tiles, max_len}; plan_out (may be NULL: dims only) receives n_updates plans of dims_out[1]
 * int32 each - [slots, chunk length, longest episode, stored transitions][batch episode indices, longest first][slots x waves tasks:
 * tile << 16 | t0 << 8 | t1; 0 = none] - and idx_out (may be NULL) the last update's draws in draw order. */
int marlhip_update_plan(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, int32_t n_agents, int32_t batch, int32_t length,
                        uint64_t seed, uint32_t counter0, int32_t n_updates, int32_t* plan_out, int64_t plan_out_ints, int32_t* idx_out,
                        int32_t* dims_out, void* stream);

/* Data-parallel form of marlhip_idqn_update_n (C-ABI 208; SURVEY.md 8e - the reference is one process and has no counterpart): one
 * process per GPU, envs and replay sharded, weights replicated, and per update ONE exchange of the flat gradient.  After the
 * gradient reduce of every update the library calls `exchange(ctx, grad, count, stream)` - the only host hop of the update - which
 * must leave the SUM over all ranks in `grad` (count floats, device memory), ordered on `stream` after the kernels already enqueued
 * there and before the next ones: an ncclAllReduce(grad, grad, count, ncclFloat, ncclSum, comm, stream) of RCCL, or
 * torch.distributed.all_reduce issued while `stream` is torch's current stream.  Return 0, anything else aborts the call.  Clip + Adam
 * then run on every rank with grad_scale = 1 / world on the identical reduced gradient, the clip norm taken from the REDUCED gradient
 * (clip_grad_norm_ over the global batch, marlbase/dqn/model.py:170), so the replicas stay bitwise in step without a weight
 * broadcast.  Hidden-64 IDQN / VDN: 3 launches per update (loss/grad, reduce, [exchange], clip + Adam + target + next packs). */
typedef int (*marlhip_exchange_fn)(void* ctx, float* grad, int64_t count, void* stream);
int marlhip_idqn_update_n_dist(const marlhip_idqn_learner* L, int32_t n_updates, int32_t length, uint64_t seed,
                               uint32_t counter0, int64_t* adam_step, int64_t* updates, int64_t* last_target_update,
                               marlhip_exchange_fn exchange, void* exchange_ctx, int32_t world, void* stream);

/* In-library gradient exchange (C-ABI 210; csrc/p2p.hip): a one-shot all-reduce (SUM) over peer-mapped buffers, one kernel per
 * call, no host hop and no collective-library launch - what marlhip_idqn_update_n_dist's `exchange` argument is meant to be given on
 * a node whose GPUs reach each other directly (xGMI).  marlhip_p2p_allreduce HAS the marlhip_exchange_fn signature (ctx = the state).
 * Every rank publishes its gradient in its own (uncached, IPC-exported) buffer and sums all ranks' buffers in RANK ORDER: bitwise
 * identical sums on every rank.  Protocol: every rank calls _create (allocates ITS buffer: the one piece of device memory this
 * library owns, freed by _destroy), the hipIpcMemHandles (marlhip_p2p_handle_bytes() bytes each) are exchanged by the host side
 * (torch.distributed in codebase_amd/parallel.py), every rank calls _connect with all of them, then any number of _allreduce calls -
 * the same sequence of counts on every rank.  A peer that does not arrive within MARLHIP_P2P_TIMEOUT_MS (default 300000: diagnostic
 * only, a collective library would block; a malformed or non-positive value keeps the default; read at every exchange, so the host side
 * can run its set-up self-test under a short bound of its own) leaves the
 * local gradient untouched and raises the state's error word, which marlhip_p2p_status reads back (it synchronises: not for the
 * hot loop); once raised, later exchanges publish but no longer wait (one timeout per dead peer, not one per update).  _destroy frees the
 * buffer: the caller synchronises its stream first (no exchange may be in flight), and the peers must have finished reading - destroy
 * after a barrier of the job, as a communicator would be.  Nothing like it exists in the reference (one process). */
int marlhip_p2p_handle_bytes(void);
int marlhip_p2p_create(int32_t rank, int32_t world, int64_t max_floats, void** state_out, void* handle_out);
int marlhip_p2p_connect(void* state, const void* handles);
/* (the launch holds at most 32 workgroups - MARLHIP_P2P_MAX_WGS overrides -, each publishing all of its 1024-float chunks before it waits
 * for the first: the number of compute units that can hold a waiting workgroup is bounded whatever the gradient's size) */
int marlhip_p2p_allreduce(void* state, float* grad, int64_t count, void* stream);
/* C-ABI 211: the same exchange in the launch geometry marlhip_idqn_update_n_dist's fused reduce uses (one workgroup per 64 values, a
 * flag per 64 floats) - for the set-up's self-test at the real gradient size (codebase_amd/parallel.py); shares the epoch counter
 * with marlhip_p2p_allreduce, so every rank issues the same sequence of calls of either kind. */
int marlhip_p2p_allreduce_wave64(void* state, float* grad, int64_t count, void* stream);
int marlhip_p2p_status(void* state);
int marlhip_p2p_destroy(void* state);

/* OPT-IN, a deviation from the exact-f32 default (C-ABI 209): the same loss / gradient (marlhip_dqn_loss_grad, mode 0) and the same
 * n-updates loop (marlhip_idqn_update_n) with every f32 product formed from fp16 halves on the double-rate matrix pipe
 * (v_mfma_f32_16x16x32_f16, fp32 accumulate): x = xh + 2^-11 xl, three MFMAs per product group, relative error ~2^-21 per product
 * (csrc/dqn_update_h16.h).  Same arguments, workspace (marlhip_dqn_workspace_bytes) and record / reduce / clip / Adam arithmetic;
 * IDQN only, two hidden layers of 64, observation width <= 32, no action masks; anything else returns an error.  Never selected
 * by default: the reference's goldens gate it at the default tolerances (tests/test_gpu_split16.py), bench.py reports it as its own
 * row (--split16).  |values| >= 65504 overflow fp16 (far outside what this path sees) and show up as inf / nan in the loss. */
int marlhip_dqn_loss_grad_split16(const marlhip_net_shape* s, const float* params, const float* target_params,
                                  const marlhip_batch* batch, float gamma, int32_t double_q, void* workspace,
                                  int64_t workspace_bytes, float* grad, float* loss, void* stream);
int marlhip_idqn_update_n_split16(const marlhip_idqn_learner* L, int32_t n_updates, int32_t length, uint64_t seed,
                                  uint32_t counter0, int64_t* adam_step, int64_t* updates, int64_t* last_target_update,
                                  void* stream);

/* cfg.standardise_returns (QNetwork._compute_loss, model.py:146-158; RunningMeanStd, marlbase/utils/standardise_stream.py):
 * device-resident running statistics, one (mean, var) pair per agent and the shared count (initialise mean 0, var 1,
 * count 1e-4).  Each call de-standardises the bootstrap values with the CURRENT statistics, updates them with all T*B
 * returns of every agent (filled or not, as the reference does), and standardises the returns with the UPDATED ones.
 *
 * VDNetwork / QMixNetwork (model.py:221-222,256-264 / 357-358,415-422) build `RunningMeanStd(shape=(1,))` and feed it the [T, B]
 * returns: `update` takes the moments over dim 0 and broadcasts the (1,)-shaped state against the [B] results, so from the first
 * update on the state is ONE (mean, var) PER BATCH COLUMN, updated with T samples per call (count += T).  Reproduced as is:
 * columns = B, mean / var [B] (initialise mean 0, var 1, count 1e-4); the batch size is then fixed for the run, as it is there. */
/* Data-parallel training (C-ABI 208): with one process per GPU the batch moments of the per-agent statistics have to be those of the
 * GLOBAL batch, or each rank's RunningMeanStd drifts apart while the parameters stay equal (standardise_stream.py:15-20 on the
 * concatenated batch).  When `exchange` is set the library folds the local batch into `moments` = {sum_p, sum of squares_p}[P], n
 * (2P + 1 doubles, device memory of the caller), calls exchange(ctx, moments, 2P + 1, stream) - which must leave the SUM over all
 * ranks there, ordered on `stream` like marlhip_exchange_fn - and updates (mean, var, count) from the global moments, identically
 * on every rank.  NULL: single process, the local batch is the batch.  Per-batch-column statistics (columns > 0) are never
 * exchanged: a rank's columns are its own part of the global batch. */
typedef int (*marlhip_exchange_f64_fn)(void* ctx, double* buf, int64_t count, void* stream);
typedef struct marlhip_ret_stats {
    float* mean;     /* [P]; per-column statistics: [columns] */
    float* var;      /* [P]; per-column statistics: [columns] */
    double* count;   /* [1] */
    int32_t columns; /* 0: one pair per agent (QNetwork); B: one pair per batch column (VDNetwork, QMixNetwork) */
    marlhip_exchange_f64_fn exchange; /* NULL: single process */
    void* exchange_ctx;
    double* moments; /* [2P + 1] device scratch, needed when `exchange` is set */
} marlhip_ret_stats;

/* mode: 0 = IDQN (QNetwork, stats->columns = 0), 1 = VDN (VDNetwork, stats->columns = batch).  QMIX takes its statistics through
 * marlhip_qmix_mixer.ret_stats. */
int marlhip_dqn_loss_grad_std(const marlhip_net_shape* s, const float* params, const float* target_params,
                              const marlhip_batch* batch, float gamma, int32_t double_q, int32_t mode,
                              const marlhip_ret_stats* stats, void* workspace, int64_t workspace_bytes, float* grad, float* loss,
                              void* stream);
int marlhip_dqn_loss_grad_std_replay(const marlhip_net_shape* s, const float* params, const float* target_params,
                                     const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* idx,
                                     int32_t batch, int32_t length, uint64_t seed, uint32_t counter, int32_t* idx_out,
                                     float gamma, int32_t double_q, int32_t mode, const marlhip_ret_stats* stats, void* workspace,
                                     int64_t workspace_bytes, float* grad, float* loss, void* stream);

/* ------------------------------------------------------------------------------------------
 * QMIX learner.  Replaces QMixNetwork._compute_loss (marlbase/dqn/model.py:374-427) with QMixer.forward
 * (model.py:313-331) for both the online and the target mixer: state = concatenation of all agents'
 * observations (model.py:389,412), reward of agent 0 (model.py:379), Double-Q bootstrap mixed by the TARGET
 * mixer on obs[1:].  Runs as agent-forward -> mixer stage (4 MFMA kernels) -> agent-backward.
 * Flat mixer block = mixer.parameters() order: hyper_w_1.{0,2}.{weight,bias}, hyper_w_final.{0,2}.{weight,bias},
 * hyper_b_1.{weight,bias}, V.{0,2}.{weight,bias} (model.py:283-312).  mixing = {embed_dim 64, hypernet_layers 2, hypernet_embed 32}
 * (configs/algorithm/qmix.yaml:14-17) on the (agents, observation) pairs of the supported envs runs on the fused MFMA kernels of
 * csrc/qmix.h; since C-ABI 212 every other QMixer the reference builds (model.py:283-301: hypernet_layers 1 - hyper_w_1 / hyper_w_final
 * are then ONE Linear each, `hyper_w_1.{weight,bias}`, `hyper_w_final.{weight,bias}` - or 2, any embed_dim / hypernet_embed up to 1024,
 * any (agents <= 16, observation) pair) runs on the generic stage of csrc/qmix_gen.hip: the hypernet layers as f32 MFMA GEMMs over all
 * rows, the mixing network one wave per row, split-K weight gradients folded in fixed order.  Workspace: the *_workspace_bytes_mx queries.
 * QNetwork.update clips the CRITIC gradient only (model.py:169-170): call marlhip_dqn_clip_adam on the critic block
 * with max_norm and on the mixer block with max_norm = 0 (same step count; one torch Adam over both lists).
 * ---------------------------------------------------------------------------------------- */
typedef struct marlhip_qmix_mixer {
    const float* mixer;        /* [marlhip_qmix_nparams] */
    const float* target_mixer; /* same layout */
    float* mixer_grad;         /* out: d loss / d mixer */
    int32_t embed_dim, hypernet_layers, hypernet_embed;
    const struct marlhip_ret_stats* ret_stats; /* cfg.standardise_returns: per-batch-column statistics (columns = batch), or NULL */
    int32_t l1_fp16; /* opt-in DEVIATION from the reference's fp32 mixer (BASELINE config 5: "fp16 mixer on MFMA"): the state-fed first layers of
                      * both mixers - 2 x state_dim x 192 MACs per row, the mixer stage's largest GEMM - run on v_mfma_f32_16x16x16_f16 with
                      * weights rounded to fp16 (states are small integers: exact), fp32 accumulation; since C-ABI 210 also the mixer's
                      * weight-gradient products (operands rounded to bf16 - per-row gradients need fp32's exponent range -, one
                      * v_mfma_f32_16x16x16_bf16 per tile pair, fp32 accumulation).  The backward pass to the agents, the bias gradients
                      * and the master parameters stay fp32.  0 = the reference's arithmetic. */
} marlhip_qmix_mixer;

int marlhip_qmix_nparams(const marlhip_net_shape* s, int32_t embed_dim, int32_t hypernet_layers, int32_t hypernet_embed);
/* scratch for marlhip_qmix_loss_grad: the agent-network workspace + first-layer activations and backward operands (mixing = {64, 2, 32}) */
int64_t marlhip_qmix_workspace_bytes(const marlhip_net_shape* s, int32_t max_len, int32_t batch);
/* C-ABI 212: the same for the mixing configuration in mixer->{embed_dim, hypernet_layers, hypernet_embed} (only those three fields are read) */
int64_t marlhip_qmix_workspace_bytes_mx(const marlhip_net_shape* s, const marlhip_qmix_mixer* mixer, int32_t max_len, int32_t batch);
/* grad[P][nparams], mixer->mixer_grad, loss[0] = value, loss[1] = sum(filled) */
int marlhip_qmix_loss_grad(const marlhip_net_shape* s, const float* params, const float* target_params,
                           const marlhip_qmix_mixer* mixer, const marlhip_batch* batch, float gamma, int32_t double_q,
                           void* workspace, int64_t workspace_bytes, float* grad, float* loss, void* stream);
/* the same with the B episodes gathered in-kernel from the replay (see marlhip_dqn_loss_grad_replay) */
int marlhip_qmix_loss_grad_replay(const marlhip_net_shape* s, const float* params, const float* target_params,
                                  const marlhip_qmix_mixer* mixer, const marlhip_replay_shape* rs,
                                  const marlhip_replay_buffers* rb, const int32_t* idx, int32_t batch, int32_t length,
                                  uint64_t seed, uint32_t counter, int32_t* idx_out, float gamma, int32_t double_q,
                                  void* workspace, int64_t workspace_bytes, float* grad, float* loss, void* stream);

/* The same n-updates loop for QMIX (C-ABI 210): n x (marlhip_qmix_loss_grad_replay -> [exchange of the joint gradient] -> clip + step on
 * the critic block -> step on the mixer block) with QNetwork.update's bookkeeping, i.e. what QMixNetwork.update + update_target do per
 * update (marlbase/dqn/model.py:165-185, 429-443: clip_grad_norm_ over the CRITIC parameters only, one optimiser step count for critic
 * and mixer, hard copy / Polyak of target AND target mixer).  `base` carries the agent side exactly as for marlhip_idqn_update_n (its
 * workspace sized by marlhip_qmix_workspace_bytes, mode ignored); `mixer.mixer` / `.target_mixer` are written (they alias mixer_rw /
 * target_mixer_rw).  exchange != NULL (data-parallel): base.grad and mixer.mixer_grad must be ONE allocation [critic | mixer] - the
 * exchange gets it as one message - and world the number of ranks; NULL: single process.  optimizer as marlhip_dqn_clip_step. */
typedef struct marlhip_qmix_learner {
    marlhip_idqn_learner base;
    marlhip_qmix_mixer mixer;
    float *mixer_rw, *target_mixer_rw;           /* the same blocks as mixer.mixer / mixer.target_mixer, writable */
    float *mixer_exp_avg, *mixer_exp_avg_sq;     /* [marlhip_qmix_nparams] each */
    float* mixer_scratch;                        /* >= ceil(n_mixer / 256) + 1 floats */
    int32_t optimizer;
} marlhip_qmix_learner;
int marlhip_qmix_update_n(const marlhip_qmix_learner* L, int32_t n_updates, int32_t length, uint64_t seed, uint32_t counter0,
                          int64_t* adam_step, int64_t* updates, int64_t* last_target_update, marlhip_exchange_fn exchange,
                          void* exchange_ctx, int32_t world, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement aid (bench.py roofline leg): when enabled, the named kernels are bracketed by HIP
 * events on the stream they are launched on.  ids: 0 loss/grad kernel, 1 fused collector,
 * 2 replay sample gather, 3 env step, 4 QMIX mixer stage, 5 the gradient exchange of the data-parallel update loops (the reduce launch
 * with the in-library exchange inside, or the exchange callback).  marlhip_timing_read waits for the recorded events,
 * returns launches and summed milliseconds, and clears the slot.
 * ---------------------------------------------------------------------------------------- */
int marlhip_timing_enable(int32_t on);
int marlhip_timing_read(int32_t id, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* MARLHIP_H */
