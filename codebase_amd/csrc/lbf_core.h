// Level-Based Foraging dynamics on a packed entity list (no dense grid).
//
// Replaces the reference's L0 env arithmetic: the third-party
// lbforaging ForagingEnv.reset/step/_make_gym_obs reached through
//   marlbase/utils/envs.py:90-97,111  (gym.make -> TimeLimit -> RecordEpisodeStatistics)
//   marlbase/dqn/train.py:203,217     (env.reset / env.step)
// plus the per-step wrapper arithmetic marlbase/utils/wrappers.py:31-45,106-108.
//
// One env = F food triples (row-major order, level 0 = eaten/absent), P player
// triples, the step counter and the spawned-food total.  Everything is a pure
// function of (state, joint action); only reset draws random numbers.  Header is
// host+device so the integer logic can be exercised by a g++ build in tests/.
#pragma once
#include <math.h>

#include "philox.h"

namespace marl {

enum : int { ACT_NONE = 0, ACT_NORTH = 1, ACT_SOUTH = 2, ACT_WEST = 3, ACT_EAST = 4, ACT_LOAD = 5 };

struct LbfParams {
    int n_envs, n_agents, n_food, rows, cols, sight;
    int max_episode_steps;  // upstream registration: 50 -> `done`
    int time_limit;         // gymnasium TimeLimit (utils/envs.py:96) -> `truncated`; 0 = none
    int force_coop, min_player_level, max_player_level;
    int min_food_level, max_food_level;  // max_food_level <= 0: sum of the 3 lowest player levels
    int normalize_reward;
    int cooperative;  // CooperativeReward wrapper (utils/wrappers.py:106-108)
    double penalty;
    uint64_t seed;
    float* reward_stats;  // StandardiseReward wrapper state (utils/wrappers.py:111-142), [n_envs][3P+1] fp32, or nullptr
    int observe_id;       // ObserveID wrapper (utils/wrappers.py:73-103): observations carry a one-hot agent-index prefix
};

template <int P, int F>
struct LbfState {
    int fr[F], fc[F], fl[F];
    int pr[P], pc[P], pl[P];
    int step, spawned;
};

// bytes per env record in HBM: 3F + 3P + 2 (step u16) + 2 (spawned u16), padded to 4
MARL_HD int lbf_state_stride(int P, int F) { return (3 * F + 3 * P + 4 + 3) & ~3; }

template <int P, int F>
MARL_HD void lbf_load(const uint8_t* rec, LbfState<P, F>& s) {
#pragma unroll
    for (int f = 0; f < F; ++f) { s.fr[f] = rec[3 * f]; s.fc[f] = rec[3 * f + 1]; s.fl[f] = rec[3 * f + 2]; }
#pragma unroll
    for (int p = 0; p < P; ++p) { s.pr[p] = rec[3 * F + 3 * p]; s.pc[p] = rec[3 * F + 3 * p + 1]; s.pl[p] = rec[3 * F + 3 * p + 2]; }
    const uint8_t* t = rec + 3 * F + 3 * P;
    s.step = t[0] | (t[1] << 8);
    s.spawned = t[2] | (t[3] << 8);
}

template <int P, int F>
MARL_HD void lbf_store(uint8_t* rec, const LbfState<P, F>& s) {
#pragma unroll
    for (int f = 0; f < F; ++f) { rec[3 * f] = (uint8_t)s.fr[f]; rec[3 * f + 1] = (uint8_t)s.fc[f]; rec[3 * f + 2] = (uint8_t)s.fl[f]; }
#pragma unroll
    for (int p = 0; p < P; ++p) { rec[3 * F + 3 * p] = (uint8_t)s.pr[p]; rec[3 * F + 3 * p + 1] = (uint8_t)s.pc[p]; rec[3 * F + 3 * p + 2] = (uint8_t)s.pl[p]; }
    uint8_t* t = rec + 3 * F + 3 * P;
    t[0] = (uint8_t)(s.step & 0xFF); t[1] = (uint8_t)(s.step >> 8);
    t[2] = (uint8_t)(s.spawned & 0xFF); t[3] = (uint8_t)(s.spawned >> 8);
}

// level of the food standing on (r,c), 0 if none  (== upstream field[r,c])
template <int P, int F>
MARL_HD int lbf_field(const LbfState<P, F>& s, int r, int c) {
    int v = 0;
#pragma unroll
    for (int f = 0; f < F; ++f) v += (s.fl[f] > 0 && s.fr[f] == r && s.fc[f] == c) ? s.fl[f] : 0;
    return v;
}

MARL_HD int imin(int a, int b) { return a < b ? a : b; }
MARL_HD int imax(int a, int b) { return a > b ? a : b; }
MARL_HD int iabs(int a) { return a < 0 ? -a : a; }

// keep the food list in np.nonzero (row-major) order - the order observations use
template <int P, int F>
MARL_HD void lbf_sort_food(LbfState<P, F>& s, int cols) {
#pragma unroll
    for (int i = 1; i < F; ++i) {
#pragma unroll
        for (int j = F - 1; j >= 1; --j) {
            if (j <= i) {
                // absent foods (level 0) sort last
                const int ka = s.fl[j - 1] > 0 ? s.fr[j - 1] * cols + s.fc[j - 1] : 0x7FFFFFFF;
                const int kb = s.fl[j] > 0 ? s.fr[j] * cols + s.fc[j] : 0x7FFFFFFF;
                if (kb < ka) {
                    int t;
                    t = s.fr[j]; s.fr[j] = s.fr[j - 1]; s.fr[j - 1] = t;
                    t = s.fc[j]; s.fc[j] = s.fc[j - 1]; s.fc[j - 1] = t;
                    t = s.fl[j]; s.fl[j] = s.fl[j - 1]; s.fl[j - 1] = t;
                }
            }
        }
    }
}

// ForagingEnv.reset: spawn_players then spawn_food (rejection sampling, <=1000
// attempts each), draws taken from `rng` in upstream's call order.
template <int P, int F>
MARL_HD void lbf_reset(const LbfParams& q, LbfState<P, F>& s, DrawStream& rng) {
#pragma unroll
    for (int f = 0; f < F; ++f) { s.fr[f] = 0; s.fc[f] = 0; s.fl[f] = 0; }
#pragma unroll
    for (int p = 0; p < P; ++p) { s.pr[p] = 0; s.pc[p] = 0; s.pl[p] = q.min_player_level; }
    // players: uniform cell until empty (field is still all-zero here).  The loop over p stays ROLLED (its body inlines three Philox
    // draws - the compiler refuses to unroll it eight times, and a run-time s.pr[p] would push the whole state into scratch memory):
    // every array access below has a compile-time index, the run-time p only appears in comparisons.
#pragma unroll 1
    for (int p = 0; p < P; ++p) {
        int attempts = 0;
        while (attempts < 1000) {
            const int row = rng.integers(0, q.rows);
            const int col = rng.integers(0, q.cols);
            bool empty = true;
#pragma unroll
            for (int o = 0; o < P; ++o) empty = empty && !(o < p && s.pr[o] == row && s.pc[o] == col);
            if (empty) {
                const int lvl = rng.integers(q.min_player_level, q.max_player_level + 1);
#pragma unroll
                for (int o = 0; o < P; ++o)
                    if (o == p) { s.pr[o] = row; s.pc[o] = col; s.pl[o] = lvl; }
                break;
            }
            ++attempts;
        }
    }
    // max food level: given, or the sum of the (up to) three lowest player levels
    int max_level = q.max_food_level;
    if (max_level <= 0) {
        int lv[P];
#pragma unroll
        for (int p = 0; p < P; ++p) lv[p] = s.pl[p];
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int j = 0; j < P - 1; ++j)
                if (lv[j + 1] < lv[j]) { const int t = lv[j]; lv[j] = lv[j + 1]; lv[j + 1] = t; }
        max_level = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) max_level += (p < 3) ? lv[p] : 0;
    }
    const int min_level = q.force_coop ? max_level : q.min_food_level;
    int food_count = 0, attempts = 0;
    while (food_count < F && attempts < 1000) {
        ++attempts;
        const int row = rng.integers(1, q.rows - 1);
        const int col = rng.integers(1, q.cols - 1);
        bool bad = false;
        // any food in the 3x3 block, or within 2 cells along the row / column
#pragma unroll
        for (int f = 0; f < F; ++f) {
            if (f < food_count) {
                const int dr = iabs(s.fr[f] - row), dc = iabs(s.fc[f] - col);
                bad = bad || (dr <= 1 && dc <= 1) || (dc == 0 && dr <= 2) || (dr == 0 && dc <= 2);
            }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) bad = bad || (s.pr[p] == row && s.pc[p] == col);
        if (bad) continue;
        const int lvl = (min_level == max_level) ? min_level : rng.integers(min_level, max_level + 1);
#pragma unroll
        for (int f = 0; f < F; ++f)
            if (f == food_count) { s.fr[f] = row; s.fc[f] = col; s.fl[f] = lvl; }
        ++food_count;
    }
    int total = 0;
#pragma unroll
    for (int f = 0; f < F; ++f) total += s.fl[f];
    s.spawned = total;
    s.step = 0;
    lbf_sort_food(s, q.cols);
}

// ForagingEnv.step.  rew[] are the env's own per-agent rewards (fp64, as
// upstream computes them); `done` = game over (no food left or step limit).
template <int P, int F>
MARL_HD void lbf_step(const LbfParams& q, LbfState<P, F>& s, const int* act, double* rew, bool& done) {
    s.step += 1;
    int tr[P], tc[P];
    bool load[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int a = act[p];
        const int r = s.pr[p], c = s.pc[p];
        // valid set of the current state (_gen_valid_moves at the end of the previous step); an invalid choice silently becomes NONE.
        // Written without a per-action switch (round 4): lanes of a wave hold different actions, so every case of a switch runs, each
        // with its own field look-ups - one look-up at the requested cell serves the four moves, one pass over the food list LOAD.
        const int dr = (a == ACT_SOUTH ? 1 : 0) - (a == ACT_NORTH ? 1 : 0), dc = (a == ACT_EAST ? 1 : 0) - (a == ACT_WEST ? 1 : 0);
        const int nr = r + dr, nc = c + dc;
        const bool move = (a >= ACT_NORTH) & (a <= ACT_EAST);
        const bool inside = (nr >= 0) & (nr <= q.rows - 1) & (nc >= 0) & (nc <= q.cols - 1);
        const bool ok_move = move & inside & (lbf_field(s, nr, nc) == 0);
        // LOAD: field[max(r-1,0)][c] + field[min(r+1,rows-1)][c] + field[r][max(c-1,0)] + field[r][min(c+1,cols-1)] > 0
        const int rn = imax(r - 1, 0), rs = imin(r + 1, q.rows - 1), cw = imax(c - 1, 0), ce = imin(c + 1, q.cols - 1);
        bool near = false;
#pragma unroll
        for (int f = 0; f < F; ++f)
            near = near | ((s.fl[f] > 0) & (((s.fc[f] == c) & ((s.fr[f] == rn) | (s.fr[f] == rs))) | ((s.fr[f] == r) & ((s.fc[f] == cw) | (s.fc[f] == ce)))));
        const bool ok = (a == ACT_NONE) | ok_move | ((a == ACT_LOAD) & near);
        a = ok ? a : ACT_NONE;
        tr[p] = r + (ok_move ? dr : 0);
        tc[p] = c + (ok_move ? dc : 0);
        load[p] = (a == ACT_LOAD);
        rew[p] = 0.0;
    }
    // a cell claimed by more than one player is reached by none of them
    bool sole[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int n = 0;
#pragma unroll
        for (int o = 0; o < P; ++o) n += (tr[o] == tr[p] && tc[o] == tc[p]) ? 1 : 0;
        sole[p] = (n == 1);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        s.pr[p] = sole[p] ? tr[p] : s.pr[p];
        s.pc[p] = sole[p] ? tc[p] : s.pc[p];
    }

    // loading, in player order (order-independent on reachable states: the spawn
    // spacing rule leaves no cell adjacent to two foods)
    bool pending[P];
    bool ate = false;
#pragma unroll
    for (int p = 0; p < P; ++p) pending[p] = load[p];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (!pending[p]) continue;
        pending[p] = false;
        const int r = s.pr[p], c = s.pc[p];
        // adjacent_food_location: N, S, W, E (upstream's `> 1` on N / W kept)
        int fr_, fc_;
        if (r > 1 && lbf_field(s, r - 1, c) > 0) { fr_ = r - 1; fc_ = c; }
        else if (r < q.rows - 1 && lbf_field(s, r + 1, c) > 0) { fr_ = r + 1; fc_ = c; }
        else if (c > 1 && lbf_field(s, r, c - 1) > 0) { fr_ = r; fc_ = c - 1; }
        else if (c < q.cols - 1 && lbf_field(s, r, c + 1) > 0) { fr_ = r; fc_ = c + 1; }
        else continue;
        const int food = lbf_field(s, fr_, fc_);
        bool adj[P];
        int adj_level = 0;
#pragma unroll
        for (int o = 0; o < P; ++o) {
            const bool next_to = (iabs(s.pr[o] - fr_) == 1 && s.pc[o] == fc_) || (iabs(s.pc[o] - fc_) == 1 && s.pr[o] == fr_);
            adj[o] = next_to && (o == p || pending[o]);
            adj_level += adj[o] ? s.pl[o] : 0;
            if (adj[o]) pending[o] = false;
        }
        if (adj_level < food) {
#pragma unroll
            for (int o = 0; o < P; ++o)
                if (adj[o]) rew[o] -= q.penalty;
            continue;
        }
#pragma unroll
        for (int o = 0; o < P; ++o) {
            if (adj[o]) {
                double v = (double)(s.pl[o] * food);
                if (q.normalize_reward) v = v / (double)(adj_level * s.spawned);
                rew[o] = v;
            }
        }
#pragma unroll
        for (int f = 0; f < F; ++f)
            if (s.fl[f] > 0 && s.fr[f] == fr_ && s.fc[f] == fc_) { s.fl[f] = 0; s.fr[f] = 0; s.fc[f] = 0; }
        ate = true;
    }
    // canonical record: live foods first (row-major), eaten slots all-zero at the tail
    if (ate) lbf_sort_food(s, q.cols);
    int left = 0;
#pragma unroll
    for (int f = 0; f < F; ++f) left += s.fl[f];
    done = (left == 0) || (q.max_episode_steps <= s.step);
}

// one element of agent `p`'s observation vector (ForagingEnv._make_gym_obs,
// vector form): F food triples in row-major order of the visible window, then
// self, then the other visible players in index order; (-1,-1,0) padding.
template <int P, int F>
struct LbfObs {
    float v[3 * (F + P)];
};

template <int P, int F>
MARL_HD void lbf_observe(const LbfParams& q, const LbfState<P, F>& s, int p, LbfObs<P, F>& o) {
    // `p` is a run-time value in the agent-per-wave collectors: s.pr[p] would be a dynamically indexed private array, i.e. the whole
    // state mirrored in scratch memory; a select chain over compile-time indices keeps it in registers (and folds when p is constant)
    int cr = 0, cc = 0, cl = 0;
#pragma unroll
    for (int a = 0; a < P; ++a) {
        const bool me = a == p;
        cr = me ? s.pr[a] : cr;
        cc = me ? s.pc[a] : cc;
        cl = me ? s.pl[a] : cl;
    }
    const int offr = imin(q.sight, cr) - cr, offc = imin(q.sight, cc) - cc;
#pragma unroll
    for (int i = 0; i < F + P; ++i) { o.v[3 * i] = -1.f; o.v[3 * i + 1] = -1.f; o.v[3 * i + 2] = 0.f; }
    int n = 0;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const bool vis = s.fl[f] > 0 && iabs(s.fr[f] - cr) <= q.sight && iabs(s.fc[f] - cc) <= q.sight;
        // slot n takes the food; written as selects on compile-time slots (a conditional store per slot is merged by the compiler
        // into ONE store at the run-time index n, which moves the observation into scratch memory)
#pragma unroll
        for (int i = 0; i < F; ++i) {
            const bool hit = vis && i == n;
            o.v[3 * i] = hit ? (float)(s.fr[f] + offr) : o.v[3 * i];
            o.v[3 * i + 1] = hit ? (float)(s.fc[f] + offc) : o.v[3 * i + 1];
            o.v[3 * i + 2] = hit ? (float)s.fl[f] : o.v[3 * i + 2];
        }
        n += vis ? 1 : 0;
    }
    // self first
    o.v[3 * F] = (float)(cr + offr); o.v[3 * F + 1] = (float)(cc + offc); o.v[3 * F + 2] = (float)cl;
    int m = 1;
#pragma unroll
    for (int a = 0; a < P; ++a) {
        if (a == p) continue;
        const int y = s.pr[a] + offr, x = s.pc[a] + offc;
        const bool vis = imin(y, x) >= 0 && imax(y, x) <= 2 * q.sight;
#pragma unroll
        for (int i = 1; i < P; ++i) {
            const bool hit = vis && i == m;
            o.v[3 * (F + i)] = hit ? (float)y : o.v[3 * (F + i)];
            o.v[3 * (F + i) + 1] = hit ? (float)x : o.v[3 * (F + i) + 1];
            o.v[3 * (F + i) + 2] = hit ? (float)s.pl[a] : o.v[3 * (F + i) + 2];
        }
        m += vis ? 1 : 0;
    }
}

// wrapper arithmetic applied to the env's rewards before they reach the learner:
// CooperativeReward = P * [python sum(reward)] in fp64, then one cast to fp32.
template <int P, class PARAMS>
MARL_HD void lbf_wrap_rewards(const PARAMS& q, uint32_t env_id, const double* raw, float* out, bool commit = true) {
    double r[P];
#pragma unroll
    for (int p = 0; p < P; ++p) r[p] = raw[p];
    if (q.reward_stats != nullptr) {
        // StandardiseReward.reward (utils/wrappers.py:118-142) with numpy's precisions: the running arrays are fp32, the
        // reward list promotes q, r and the result to fp64; per-env record = sumw[P] | wmean[P] | t[P] | n (int32 bits).
        float* st = q.reward_stats + (size_t)env_id * (3 * P + 1);
        int n;
        {
            const float nf = st[3 * P];
            n = *reinterpret_cast<const int*>(&nf) + 1;
        }
        float sumw[P], wmean[P], tacc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float sw = st[p], wm = st[P + p], ta = st[2 * P + p];
            const double qd = r[p] - (double)wm;
            const float tsw = sw + 1.0f;
            const double rr = qd * 1.0 / (double)tsw;
            wmean[p] = (float)((double)wm + rr);
            tacc[p] = (float)((double)ta + qd * rr * (double)sw);
            sumw[p] = tsw;
        }
        if (n > 1) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const float var = (tacc[p] * (float)n) / (sumw[p] * (float)(n - 1));
                r[p] = (r[p] - (double)wmean[p]) / (double)(sqrtf(var) + 1e-6f);
            }
        }
        if (commit) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                st[p] = sumw[p];
                st[P + p] = wmean[p];
                st[2 * P + p] = tacc[p];
            }
            st[3 * P] = *reinterpret_cast<const float*>(&n);
        }
    }
    if (q.cooperative) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < P; ++p) t += r[p];
#pragma unroll
        for (int p = 0; p < P; ++p) out[p] = (float)t;
    } else {
#pragma unroll
        for (int p = 0; p < P; ++p) out[p] = (float)r[p];
    }
}

}  // namespace marl
