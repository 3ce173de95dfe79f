// Shared host-side plumbing of libmarlhip.so: error text, launch checks, shape dispatch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/marlhip.h"
#include "lbf_core.h"
#include "rware_core.h"

namespace marl {

void set_error(const char* fmt, ...);

// optional per-kernel HIP-event timing (bench.py's roofline leg): off by default
enum { TIMER_LOSSGRAD = 0, TIMER_COLLECT = 1, TIMER_SAMPLE = 2, TIMER_ENVSTEP = 3, TIMER_QMIX = 4,
       TIMER_EXCHANGE = 5,  // data-parallel updates: the reduce launch with the in-library exchange inside, or reduce + the exchange callback
       TIMER_COUNT = 6 };
void timing_begin(int id, hipStream_t st);
void timing_end(int id, hipStream_t st);

// Pack scratch of the forward-only entry points (collectors, act, sequence forwards): CALLER memory.  An extern "C" entry point binds
// the workspace it was given for the duration of the call (thread-local, so calls on different host threads never meet);
// collect_pack_scratch hands out its start (one pack set alive at a time per call, as the launches of a call are stream-ordered)
// and fails with the size it wanted when the region is missing or too small.  Nothing is allocated, nothing outlives the call.
void scratch_bind(void* base, int64_t bytes);
struct ScratchScope {
    ScratchScope(void* base, int64_t bytes) { scratch_bind(base, bytes); }
    ~ScratchScope() { scratch_bind(nullptr, 0); }
    ScratchScope(const ScratchScope&) = delete;
    ScratchScope& operator=(const ScratchScope&) = delete;
};
int64_t scratch_avail();
float* collect_pack_scratch(size_t bytes, hipStream_t st);  // nullptr + marlhip_last_error() text when the bound region cannot hold it

// Second pass of a rollout: the reference keeps stepping the envs whose episode ended (auto-reset vector env, the policy acts on
// them too) until the LAST env's first episode ends, and appends the info of every further episode that finishes meanwhile
// (ac/train.py:71,101-110); with env.standardise_rewards those steps also move the envs' running reward statistics.  The first
// pass ignores finished envs, so the envs that finished before t_stop run again here - from the auto-reset state (reset stream
// 2 * round + 1, then + 2, ...), from their own finishing step t_start to t_stop, same action noise (keyed on the global step),
// no batch writes - and leave (returns, length, finishing step) of up to `cap` further episodes each.  env_ids == NULL: first pass.
struct AcGhost {
    const int32_t* env_ids;  // [n] env index of lane n (Philox streams, reward statistics)
    const int32_t* t_start;  // [n] first step of the second pass = length of the env's first episode
    int t_stop, cap;
    float* ret;              // [n][cap][P]
    int32_t* meta;           // [n][cap][2] = (episode length, finishing step)
    int32_t* cnt;            // [n] episodes recorded (<= cap)
};
const AcGhost& ac_ghost_current();        // the calling host thread's pass description (api.hip); env_ids == NULL outside a second pass
void ac_ghost_bind(const AcGhost* g);
struct AcGhostScope {
    explicit AcGhostScope(const AcGhost& g) { ac_ghost_bind(&g); }
    ~AcGhostScope() { ac_ghost_bind(nullptr); }
    AcGhostScope(const AcGhostScope&) = delete;
    AcGhostScope& operator=(const AcGhostScope&) = delete;
};

// The actors' forward pass kept by the rollout (round 5).  A2C updates once per rollout, on the parameters the rollout was sampled with
// (ac/train.py:203-212 -> ac/model.py:189-246): the logits and both hidden layers the learner step computes for every batch row
// (mlp_rows_fwd_kernel<S, HS> over T*B rows per agent) are the values the collector had in registers when it sampled that row's action -
// the same packs, the same operand order, bit for bit.  With a record bound, the fused collector writes them where the learner step
// reads them (AcWs::logits, AcWs::rec_a) and marlhip_ac_config.actor_forward_kept skips the pass.  hid == NULL: nothing is kept.
struct AcKeep {
    float* logits;           // [P][T*B][A], row = t * B + env
    float* hid;              // 16-byte tiles [64 lanes] in the MFMA C layout: tile index ((p T + t) bpt + env / 16) stride + tile of the layer
    int T, B, bpt, stride;   // bpt row-block slots per time step, `stride` tiles per slot
    int64_t off_h1, off_h2;  // tile offsets of the layers' tile 0 (hidden 128: the h1 record behind the h2 record, tp_bwd_kernel<STORED1>;
                             // hidden 64: h1 | h2 inside a slot, dqn_lossgrad_kernel<MODE 4, STORED>)
};
const AcKeep& ac_keep_current();  // the calling host thread's binding (api.hip)
void ac_keep_bind(const AcKeep* k);
struct AcKeepScope {
    explicit AcKeepScope(const AcKeep& k) { ac_keep_bind(&k); }
    ~AcKeepScope() { ac_keep_bind(nullptr); }
    AcKeepScope(const AcKeepScope&) = delete;
    AcKeepScope& operator=(const AcKeepScope&) = delete;
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: one slot per device ordinal, raised monotonically (a racing second
// call sets the same value)
struct LdsAttr {
    static constexpr int SLOTS = 64;
    size_t set[SLOTS] = {};
    static int dev() {
        int d = 0;
        (void)hipGetDevice(&d);
        return d;
    }
    // device ordinals beyond the table never alias another device's slot: they set the attribute on every call (cheap, idempotent)
    bool need(size_t bytes = 1) const { const int d = dev(); return d < 0 || d >= SLOTS || set[d] < bytes; }  // if (a.need(n)) { hipFuncSetAttribute...; a.done(n); }
    void done(size_t bytes = 1) { const int d = dev(); if (d >= 0 && d < SLOTS) set[d] = bytes; }
};

#define MARL_CHECK_LAUNCH(what)                                                   \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            marl::set_error("%s: %s", what, hipGetErrorString(e_));               \
            return -2;                                                            \
        }                                                                         \
    } while (0)

#define MARL_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            marl::set_error(__VA_ARGS__);    \
            return -1;                       \
        }                                    \
    } while (0)

inline LbfParams to_params(const marlhip_lbf_config* c) {
    LbfParams q;
    q.n_envs = c->n_envs; q.n_agents = c->n_agents; q.n_food = c->n_food;
    q.rows = c->rows; q.cols = c->cols; q.sight = c->sight;
    q.max_episode_steps = c->max_episode_steps; q.time_limit = c->time_limit;
    q.force_coop = c->force_coop; q.min_player_level = c->min_player_level; q.max_player_level = c->max_player_level;
    q.min_food_level = c->min_food_level; q.max_food_level = c->max_food_level;
    q.normalize_reward = c->normalize_reward; q.cooperative = c->cooperative;
    q.penalty = c->penalty; q.seed = c->seed;
    q.reward_stats = c->reward_stats;
    q.observe_id = c->observe_id != 0;
    return q;
}

// (players, foods) shapes with compiled kernels.  X(P, F)
#define MARL_LBF_SHAPES(X) X(2, 2) X(2, 3) X(3, 3) X(3, 5) X(4, 3) X(4, 5) X(8, 5)

// agent counts with compiled warehouse kernels.  X(P)
#define MARL_RW_SHAPES(X) X(2) X(4) X(8)

inline bool lbf_shape_supported(int P, int F) {
#define X(p, f) if (P == p && F == f) return true;
    MARL_LBF_SHAPES(X)
#undef X
    return false;
}

inline int lbf_validate(const marlhip_lbf_config* c) {
    MARL_REQUIRE(c != nullptr, "lbf config is NULL");
    MARL_REQUIRE(lbf_shape_supported(c->n_agents, c->n_food),
                 "no LBF kernel for %dp-%df (add it to MARL_LBF_SHAPES in csrc/common.h and rebuild)", c->n_agents, c->n_food);
    MARL_REQUIRE(c->n_envs > 0, "n_envs must be > 0");
    MARL_REQUIRE(c->rows >= 3 && c->cols >= 3 && c->rows <= 255 && c->cols <= 255, "field size %dx%d out of range", c->rows, c->cols);
    MARL_REQUIRE(c->min_player_level >= 1 && c->max_player_level >= c->min_player_level && c->max_player_level <= 20, "player level range");
    MARL_REQUIRE(c->max_episode_steps > 0 && c->max_episode_steps < 65535, "max_episode_steps");
    return 0;
}

inline RwParams to_rw_params(const marlhip_rware_config* c) {
    RwParams q;
    q.n_envs = c->n_envs; q.n_agents = c->n_agents;
    q.column_height = c->column_height;
    q.rows = (c->column_height + 1) * c->shelf_rows + 2;
    q.cols = 3 * c->shelf_columns + 1;
    q.n_shelves = rw_count_shelves(q);
    q.queue_size = c->request_queue_size;
    q.max_steps = c->max_steps; q.max_inactivity_steps = c->max_inactivity_steps; q.time_limit = c->time_limit;
    q.reward_type = c->reward_type; q.cooperative = c->cooperative; q.seed = c->seed;
    q.reward_stats = c->reward_stats; q.observe_id = c->observe_id != 0;
    return q;
}

inline int rw_validate(const marlhip_rware_config* c) {
    MARL_REQUIRE(c != nullptr, "rware config is NULL");
    bool ok = false;
#define X(p) ok = ok || c->n_agents == p;
    MARL_RW_SHAPES(X)
#undef X
    MARL_REQUIRE(ok, "no rware kernel for %d agents (add it to MARL_RW_SHAPES in csrc/common.h and rebuild)", c->n_agents);
    MARL_REQUIRE(c->n_envs > 0, "n_envs must be > 0");
    MARL_REQUIRE(c->shelf_columns >= 1 && c->shelf_columns % 2 == 1, "rware: only an odd number of shelf columns is supported");
    MARL_REQUIRE(c->shelf_rows >= 1 && c->shelf_rows <= 5 && c->column_height >= 1, "rware: shelf_rows (1..5) / column_height");
    const RwParams q = to_rw_params(c);
    MARL_REQUIRE(q.rows <= 255 && q.cols <= 255 && q.n_shelves <= 255, "rware: grid %dx%d with %d shelves does not fit the byte state", q.rows,
                 q.cols, q.n_shelves);
    MARL_REQUIRE(c->request_queue_size >= 1 && c->request_queue_size <= 2 * c->n_agents && c->request_queue_size < q.n_shelves,
                 "rware: request_queue_size %d (1..%d supported)", c->request_queue_size, 2 * c->n_agents);
    MARL_REQUIRE(q.rows * q.cols >= c->n_agents, "rware: more agents than cells");
    MARL_REQUIRE(c->max_steps >= 0 && c->max_steps < 65535 && c->max_inactivity_steps >= 0 && c->max_inactivity_steps < 65535, "rware: step limits");
    MARL_REQUIRE(c->reward_type >= 0 && c->reward_type <= 2, "rware: reward_type %d", c->reward_type);
    return 0;
}

// agent -> parameter block (identity for independent networks); passed to kernels by value
struct AgentMap {
    int nblk;         // number of parameter blocks (networks)
    int8_t net[16];
    int8_t depth;     // recurrent networks: stacked GRU layers, len(layers) - 1 (gru_stack.h); unused by the feed-forward kernels
};

inline AgentMap agent_map(const marlhip_net_shape* s) {
    AgentMap m;
    m.nblk = s->n_networks > 0 ? s->n_networks : s->n_agents;
    for (int i = 0; i < 16; ++i) m.net[i] = (int8_t)(s->n_networks > 0 ? (i < s->n_agents ? s->net_of[i] : 0) : i);
    const int depth = (s->n_hidden > 0 ? s->n_hidden : 2) - 1;
    m.depth = (int8_t)(depth < 1 ? 1 : (depth > 16 ? 16 : depth));
    return m;
}

// any_depth: the caller runs the GEMM path (1 .. 4 hidden layers); every fused kernel is built for two
inline int agent_map_validate(const marlhip_net_shape* s, bool any_depth = false) {
    MARL_REQUIRE(s->n_agents >= 1 && s->n_agents <= 16, "net shape: %d agents (1..16 supported)", s->n_agents);
    MARL_REQUIRE(s->n_hidden == 0 || s->n_hidden == 2 || (any_depth && s->n_hidden >= 1 && s->n_hidden <= 16),
                 "net shape: %d hidden layers (the fused kernels implement 2; 1..16 run on the GEMM path, marlhip_wide_*)", s->n_hidden);
    if (s->n_networks > 0) {
        MARL_REQUIRE(s->n_networks <= s->n_agents, "net shape: %d networks for %d agents", s->n_networks, s->n_agents);
        for (int i = 0; i < s->n_agents; ++i)
            MARL_REQUIRE(s->net_of[i] >= 0 && s->net_of[i] < s->n_networks, "net shape: net_of[%d] = %d out of range", i, s->net_of[i]);
    }
    return 0;
}

// network shapes (D, H, A) with compiled MFMA kernels.  X(D, H, A)
//   LBF obs dims: 2p2f 12, 2p3f 15, 3p3f 18, 3p5f 24, 4p3f 21, 4p5f 27, 8p5f 39
//   with env.observe_id (+P): 14, 17, 21, 27, 25, 31, 47
#define MARL_NET_SHAPES(X)                                                                         \
    X(12, 64, 6) X(15, 64, 6) X(18, 64, 6) X(21, 64, 6) X(24, 64, 6) X(27, 64, 6) X(39, 64, 6)     \
    X(12, 128, 6) X(15, 128, 6) X(18, 128, 6) X(21, 128, 6) X(24, 128, 6) X(27, 128, 6) X(39, 128, 6) \
    X(14, 64, 6) X(17, 64, 6) X(25, 64, 6) X(31, 64, 6) X(47, 64, 6)                               \
    X(14, 128, 6) X(17, 128, 6) X(25, 128, 6) X(31, 128, 6) X(47, 128, 6)                          \
    X(71, 64, 5) X(71, 128, 5) /* rware: 71 floats, 5 actions */

}  // namespace marl
