// TEST TOOL (not product): g++ build of codebase_amd/csrc/lbf_core.h so the integer env logic
// that the HIP kernels inline can be checked against oracle/lbf.py on a machine with no GPU.
#include <stdint.h>
#include <string.h>
#include <vector>
#include <string.h>

#include "../../codebase_amd/csrc/lbf_core.h"

using namespace marl;

struct HostCfg {
    int32_t n_envs, n_agents, n_food, rows, cols, sight, max_episode_steps, time_limit;
    int32_t force_coop, min_player_level, max_player_level, min_food_level, max_food_level;
    int32_t normalize_reward, cooperative;
    double penalty;
    uint64_t seed;
    float* reward_stats;
};

static LbfParams conv(const HostCfg* c) {
    LbfParams q;
    q.n_envs = c->n_envs; q.n_agents = c->n_agents; q.n_food = c->n_food; q.rows = c->rows; q.cols = c->cols;
    q.sight = c->sight; q.max_episode_steps = c->max_episode_steps; q.time_limit = c->time_limit;
    q.force_coop = c->force_coop; q.min_player_level = c->min_player_level; q.max_player_level = c->max_player_level;
    q.min_food_level = c->min_food_level; q.max_food_level = c->max_food_level; q.normalize_reward = c->normalize_reward;
    q.cooperative = c->cooperative; q.penalty = c->penalty; q.seed = c->seed;
    q.reward_stats = c->reward_stats;
    q.observe_id = 0;
    return q;
}

template <int P, int F>
static void run_reset(const LbfParams& q, uint8_t* state, const uint32_t* episode, float* obs) {
    const int stride = lbf_state_stride(P, F), D = 3 * (P + F);
    for (int n = 0; n < q.n_envs; ++n) {
        LbfState<P, F> s;
        DrawStream rng;
        rng.init(q.seed, (uint32_t)n, episode[n], STREAM_RESET);
        lbf_reset(q, s, rng);
        lbf_store(state + (size_t)n * stride, s);
        for (int p = 0; p < P; ++p) {
            LbfObs<P, F> o;
            lbf_observe(q, s, p, o);
            for (int d = 0; d < D; ++d) obs[((size_t)p * q.n_envs + n) * D + d] = o.v[d];
        }
    }
}

template <int P, int F>
static void run_step(const LbfParams& q, uint8_t* state, const int32_t* actions, float* obs, float* rew, double* raw_out,
                     uint8_t* done, uint8_t* trunc) {
    const int stride = lbf_state_stride(P, F), D = 3 * (P + F);
    for (int n = 0; n < q.n_envs; ++n) {
        LbfState<P, F> s;
        lbf_load(state + (size_t)n * stride, s);
        int a[P];
        double raw[P];
        float rw[P];
        bool d = false;
        for (int p = 0; p < P; ++p) a[p] = actions[(size_t)p * q.n_envs + n];
        lbf_step(q, s, a, raw, d);
        lbf_wrap_rewards<P>(q, (uint32_t)n, raw, rw);
        lbf_store(state + (size_t)n * stride, s);
        done[n] = d;
        trunc[n] = q.time_limit > 0 && s.step >= q.time_limit;
        for (int p = 0; p < P; ++p) {
            rew[(size_t)p * q.n_envs + n] = rw[p];
            raw_out[(size_t)p * q.n_envs + n] = raw[p];
            LbfObs<P, F> o;
            lbf_observe(q, s, p, o);
            for (int dd = 0; dd < D; ++dd) obs[((size_t)p * q.n_envs + n) * D + dd] = o.v[dd];
        }
    }
}

#define SHAPES(X) X(2, 2) X(2, 3) X(3, 3) X(3, 5) X(4, 3) X(4, 5) X(8, 5)

extern "C" int host_lbf_stride(int P, int F) { return lbf_state_stride(P, F); }

extern "C" int host_lbf_reset(const HostCfg* c, uint8_t* state, const uint32_t* episode, float* obs) {
    const LbfParams q = conv(c);
#define X(p, f) if (c->n_agents == p && c->n_food == f) { run_reset<p, f>(q, state, episode, obs); return 0; }
    SHAPES(X)
#undef X
    return -1;
}

extern "C" int host_lbf_step(const HostCfg* c, uint8_t* state, const int32_t* actions, float* obs, float* rew, double* raw,
                             uint8_t* done, uint8_t* trunc) {
    const LbfParams q = conv(c);
#define X(p, f) if (c->n_agents == p && c->n_food == f) { run_step<p, f>(q, state, actions, obs, rew, raw, done, trunc); return 0; }
    SHAPES(X)
#undef X
    return -1;
}

extern "C" void host_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    U4 c; c.x = ctr[0]; c.y = ctr[1]; c.z = ctr[2]; c.w = ctr[3];
    U4 r = philox4x32_10(c, key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// ---- multi-robot warehouse core (csrc/rware_core.h) against oracle/rware.py ----
#include "../../codebase_amd/csrc/rware_core.h"

struct HostRwCfg {
    int32_t n_envs, n_agents, rows, cols, column_height, n_shelves, queue_size, max_steps, max_inactivity_steps, time_limit;
    int32_t reward_type, cooperative;
    uint64_t seed;
};

static RwParams rw_conv(const HostRwCfg* c) {
    RwParams q;
    q.n_envs = c->n_envs; q.n_agents = c->n_agents; q.rows = c->rows; q.cols = c->cols; q.column_height = c->column_height;
    q.n_shelves = c->n_shelves; q.queue_size = c->queue_size; q.max_steps = c->max_steps;
    q.max_inactivity_steps = c->max_inactivity_steps; q.time_limit = c->time_limit; q.reward_type = c->reward_type;
    q.cooperative = c->cooperative; q.seed = c->seed; q.reward_stats = nullptr; q.observe_id = 0;
    return q;
}

template <int P>
static void rw_obs_all(const RwParams& q, const RwState<P>& s, const RwGrid& grid, int n, float* obs) {
    for (int p = 0; p < P; ++p) {
        int code[9];
        rw_window(q, s, grid, p, code);
        for (int d = 0; d < RW_OBS_DIM; ++d) obs[((size_t)p * q.n_envs + n) * RW_OBS_DIM + d] = rw_obs_elem(q, s, p, code, d);
    }
}

template <int P>
static void rw_run_reset(const RwParams& q, uint8_t* state, const uint32_t* episode, float* obs) {
    const int stride = rw_state_stride(P, q.rows, q.cols), cells = q.rows * q.cols;
    for (int n = 0; n < q.n_envs; ++n) {
        RwState<P> s;
        uint8_t* rec = state + (size_t)n * stride;
        const RwGrid grid{rec, 1};
        DrawStream rng;
        rng.init(q.seed, (uint32_t)n, episode[n], STREAM_RESET);
        rw_reset(q, s, grid, rng);
        rw_store(rec, cells, s);
        rw_obs_all<P>(q, s, grid, n, obs);
    }
}

template <int P>
static void rw_run_step(const RwParams& q, uint8_t* state, const uint32_t* episode, const int32_t* actions, float* obs, float* rew,
                        uint8_t* done, uint8_t* trunc) {
    const int stride = rw_state_stride(P, q.rows, q.cols), cells = q.rows * q.cols;
    for (int n = 0; n < q.n_envs; ++n) {
        RwState<P> s;
        uint8_t* rec = state + (size_t)n * stride;
        const RwGrid grid{rec, 1};
        rw_load(rec, cells, s);
        int a[P];
        double raw[P];
        float rw[P];
        bool d = false;
        for (int p = 0; p < P; ++p) a[p] = actions[(size_t)p * q.n_envs + n];
        DrawStream req;
        req.init(q.seed, (uint32_t)n, episode[n], STREAM_REQUEST);
        rw_step(q, s, grid, a, raw, d, req);
        lbf_wrap_rewards<P>(q, (uint32_t)n, raw, rw);
        rw_store(rec, cells, s);
        done[n] = d;
        trunc[n] = q.time_limit > 0 && s.steps >= q.time_limit;
        for (int p = 0; p < P; ++p) rew[(size_t)p * q.n_envs + n] = rw[p];
        rw_obs_all<P>(q, s, grid, n, obs);
    }
}

extern "C" int host_rw_stride(int P, int rows, int cols) { return rw_state_stride(P, rows, cols); }

extern "C" int host_rw_count_shelves(const HostRwCfg* c) { return rw_count_shelves(rw_conv(c)); }

extern "C" int host_rw_reset(const HostRwCfg* c, uint8_t* state, const uint32_t* episode, float* obs) {
    const RwParams q = rw_conv(c);
#define X(p) if (c->n_agents == p) { rw_run_reset<p>(q, state, episode, obs); return 0; }
    X(2) X(4) X(8)
#undef X
    return -1;
}

extern "C" int host_rw_step(const HostRwCfg* c, uint8_t* state, const uint32_t* episode, const int32_t* actions, float* obs, float* rew,
                            uint8_t* done, uint8_t* trunc) {
    const RwParams q = rw_conv(c);
#define X(p) if (c->n_agents == p) { rw_run_step<p>(q, state, episode, actions, obs, rew, done, trunc); return 0; }
    X(2) X(4) X(8)
#undef X
    return -1;
}

// The fused collectors keep the request queue as a bit set (RwRequested) ACROSS steps and rebuild it only when rw_step reports a delivery
// (env_traits.h RwEnvT::step).  rq_io [n_envs][4]: the carried words, in: as left by the previous call (build_only != 0: just build them
// from the state, as RwEnvT::reset does); returns the number of envs whose carried set differs from a fresh build after the step.
extern "C" int host_rw_step_carried_rq(const HostRwCfg* c, const uint8_t* state_before, const uint8_t* state_after, const uint32_t* episode,
                                       const int32_t* actions, uint64_t* rq_io, int build_only) {
    const RwParams q = rw_conv(c);
    int bad = 0;
#define X(p)                                                                                                          \
    if (c->n_agents == p) {                                                                                           \
        const int stride = rw_state_stride(p, q.rows, q.cols), cells = q.rows * q.cols;                               \
        for (int n = 0; n < q.n_envs; ++n) {                                                                          \
            RwState<p> s;                                                                                             \
            std::vector<uint8_t> rec(state_before + (size_t)n * stride, state_before + (size_t)(n + 1) * stride);     \
            const RwGrid grid{rec.data(), 1};                                                                         \
            rw_load(rec.data(), cells, s);                                                                            \
            RwRequested<p> rq;                                                                                        \
            if (build_only) {                                                                                         \
                rq.build(q, s);                                                                                       \
            } else {                                                                                                  \
                for (int k = 0; k < 4; ++k) rq.w[k] = rq_io[(size_t)n * 4 + k];                                       \
                int a[p];                                                                                             \
                double raw[p];                                                                                        \
                bool d = false;                                                                                       \
                for (int i = 0; i < p; ++i) a[i] = actions[(size_t)i * q.n_envs + n];                                 \
                DrawStream req;                                                                                       \
                req.init(q.seed, (uint32_t)n, episode[n], STREAM_REQUEST);                                            \
                if (rw_step(q, s, grid, a, raw, d, req)) rq.build(q, s);                                              \
                rw_store(rec.data(), cells, s);                                                                       \
                bad += memcmp(rec.data(), state_after + (size_t)n * stride, stride) != 0; /* the same step as host_rw_step took */ \
                RwRequested<p> fresh;                                                                                 \
                fresh.build(q, s);                                                                                    \
                bad += memcmp(fresh.w, rq.w, sizeof(rq.w)) != 0;                                                      \
            }                                                                                                         \
            for (int k = 0; k < 4; ++k) rq_io[(size_t)n * 4 + k] = rq.w[k];                                           \
        }                                                                                                             \
        return bad;                                                                                                   \
    }
    X(2) X(4) X(8)
#undef X
    return -1;
}

// the collectors' packed-window observation route against the per-cell one
extern "C" int host_rw_obs_word_check(const HostRwCfg* c, const uint8_t* state) {
    const RwParams q = rw_conv(c);
    int bad = 0;
#define X(p)                                                                                                     \
    if (c->n_agents == p) {                                                                                      \
        const int stride = rw_state_stride(p, q.rows, q.cols), cells = q.rows * q.cols;                          \
        for (int n = 0; n < q.n_envs; ++n) {                                                                     \
            RwState<p> s;                                                                                        \
            uint8_t* rec = const_cast<uint8_t*>(state) + (size_t)n * stride;                                     \
            const RwGrid grid{rec, 1};                                                                           \
            rw_load(rec, cells, s);                                                                              \
            RwRequested<p> rq;                                                                                   \
            rq.build(q, s);                                                                                      \
            for (int a = 0; a < p; ++a) {                                                                        \
                int code[9];                                                                                     \
                rw_window(q, s, grid, a, code);                                                                  \
                const uint64_t w = rw_window_word(q, s, grid, rq, a);                                            \
                const uint64_t bits = rw_obs_bits(q, s, grid, rq, a);                                            \
                for (int d = 0; d < RW_OBS_DIM; ++d) bad += rw_obs_elem(q, s, a, code, d) != rw_obs_elem_word(q, s, a, w, d); \
                for (int d = 0; d < RW_OBS_DIM; ++d) bad += rw_obs_elem(q, s, a, code, d) != rw_obs_elem_bits(q, s, a, bits, d); \
                bad += (bits >> 63) != 0;                                                                        \
            }                                                                                                    \
        }                                                                                                        \
        return bad;                                                                                              \
    }
    X(2) X(4) X(8)
#undef X
    return -1;
}

// the movement-conflict rule alone: edges as (start cell, target cell) per agent -> committed-agent bit mask
extern "C" int host_rw_resolve(int P, const int32_t* start, const int32_t* target) {
    uint32_t nxt = 0;
    int tc[8];
    for (int p = 0; p < P; ++p) {
        int o = 0xF;
        for (int k = 0; k < P; ++k)
            if (start[k] == target[p]) o = k;
        nxt = rw_set_nib(nxt, p, o);
        tc[p] = target[p];
    }
    if (P == 2) return (int)rw_resolve<2>(nxt, tc);
    if (P == 4) return (int)rw_resolve<4>(nxt, tc);
    if (P == 8) return (int)rw_resolve<8>(nxt, tc);
    return -1;
}
