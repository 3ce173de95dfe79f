// Multi-robot warehouse (rware) dynamics: agents in registers, the shelf layer as one byte per cell.
//
// Replaces the reference's L0 env arithmetic for config 4: the third-party rware Warehouse.reset / step / _make_obs
// (flattened observations, msg_bits 0, sensor_range 1) reached through
//   marlbase/utils/envs.py:27-37,82-97  (gym.make -> TimeLimit -> RecordEpisodeStatistics)
//   marlbase/ac/train.py:30,79          (envs.reset / envs.step)
// Restated in oracle/rware.py (parity unpinned: the package is absent; see that file's header for what is pinned).
//
// One env = a [rows*cols] byte grid (shelf id standing on / carried over the cell, 0 = none), P agents
// (x, y, direction, carried shelf id, has_delivered), the request queue (shelf ids) and two step counters.
// Movement conflicts: upstream's networkx graph (cell -> requested cell) has out-degree <= 1, so a component is a cycle
// with in-trees or an in-tree draining into a free cell; rw_resolve is the closed form of "commit the cycle (not a
// 2-cycle), else the longest path", ties between equally long feeder chains going to the lowest agent index.
// Header is host+device (tests/host_shim builds it with g++).
#pragma once
#include "philox.h"

namespace marl {

enum : int { RW_NOOP = 0, RW_FORWARD = 1, RW_LEFT = 2, RW_RIGHT = 3, RW_TOGGLE = 4 };
enum : int { RW_UP = 0, RW_DOWN = 1, RW_DLEFT = 2, RW_DRIGHT = 3 };
enum : int { RW_REWARD_GLOBAL = 0, RW_REWARD_INDIVIDUAL = 1, RW_REWARD_TWO_STAGE = 2 };
enum : uint32_t { STREAM_REQUEST = 3u };

struct RwParams {
    int n_envs, n_agents, rows, cols, column_height;
    int n_shelves, queue_size;
    int max_steps, max_inactivity_steps;  // upstream registration: 500 / none -> `done`
    int time_limit;                       // gymnasium TimeLimit -> `truncated`; 0 = none
    int reward_type;
    int cooperative;      // CooperativeReward wrapper (utils/wrappers.py:106-108)
    uint64_t seed;
    float* reward_stats;  // StandardiseReward wrapper state, [n_envs][3P+1] fp32, or nullptr
    int observe_id;       // ObserveID wrapper: one-hot agent-index prefix
};

template <int P>
struct RwState {
    static constexpr int MAXQ = 2 * P;  // "-easy" tasks queue 2 requests per agent
    int ax[P], ay[P], ad[P], ac[P], adel[P];
    int rq[MAXQ];
    int steps, inactive;
};

// shelf layer of one env: byte `cell` lives at g[cell * stride]  (stride 1 in an HBM record, 64 in a workgroup's LDS)
struct RwGrid {
    uint8_t* g;
    int stride;
    MARL_HD int get(int cell) const { return g[(size_t)cell * stride]; }
    MARL_HD void set(int cell, int v) const { g[(size_t)cell * stride] = (uint8_t)v; }
};

MARL_HD bool rw_is_highway(const RwParams& q, int x, int y) {
    // x % 3 == 0 as one bit of a constant (cols <= 16 in every registered layout; the modulo is a quarter-rate multiply), the cross
    // aisles y % (column_height + 1) == 0 and the bottom row as one bit of a mask that only depends on the layout (wave-uniform:
    // scalar registers, hoisted out of the rollout loop): rows = m * shelf_rows + 2 <= 47 with shelf_rows <= 5
    const int m = q.column_height + 1;
    const uint64_t rows_hw = 1ull | (1ull << m) | (1ull << (2 * m < 63 ? 2 * m : 63)) | (1ull << (3 * m < 63 ? 3 * m : 63)) |
                             (1ull << (4 * m < 63 ? 4 * m : 63)) | (1ull << (5 * m < 63 ? 5 * m : 63)) | (1ull << (q.rows - 1));
    const bool col3 = x < 32 ? ((0x49249249u >> x) & 1u) != 0 : x % 3 == 0;
    return col3 | (((rows_hw >> y) & 1ull) != 0) |
           ((y > q.rows - (q.column_height + 3)) & ((x == q.cols / 2 - 1) | (x == q.cols / 2)));
}

MARL_HD int rw_count_shelves(const RwParams& q) {
    int n = 0;
    for (int y = 0; y < q.rows; ++y)
        for (int x = 0; x < q.cols; ++x) n += rw_is_highway(q, x, y) ? 0 : 1;
    return n;
}

// bytes per env record in HBM: grid | P x (x, y, dir, carry, delivered) | 2P queue ids | steps u16 | inactive u16
MARL_HD int rw_state_stride(int P, int rows, int cols) { return (rows * cols + 5 * P + 2 * P + 4 + 3) & ~3; }

template <int P>
MARL_HD void rw_load(const uint8_t* rec, int cells, RwState<P>& s) {
    const uint8_t* a = rec + cells;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        s.ax[p] = a[5 * p]; s.ay[p] = a[5 * p + 1]; s.ad[p] = a[5 * p + 2]; s.ac[p] = a[5 * p + 3]; s.adel[p] = a[5 * p + 4];
    }
#pragma unroll
    for (int k = 0; k < 2 * P; ++k) s.rq[k] = a[5 * P + k];
    const uint8_t* t = a + 7 * P;
    s.steps = t[0] | (t[1] << 8);
    s.inactive = t[2] | (t[3] << 8);
}

template <int P>
MARL_HD void rw_store(uint8_t* rec, int cells, const RwState<P>& s) {
    uint8_t* a = rec + cells;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        a[5 * p] = (uint8_t)s.ax[p]; a[5 * p + 1] = (uint8_t)s.ay[p]; a[5 * p + 2] = (uint8_t)s.ad[p];
        a[5 * p + 3] = (uint8_t)s.ac[p]; a[5 * p + 4] = (uint8_t)s.adel[p];
    }
#pragma unroll
    for (int k = 0; k < 2 * P; ++k) a[5 * P + k] = (uint8_t)s.rq[k];
    uint8_t* t = a + 7 * P;
    t[0] = (uint8_t)(s.steps & 0xFF); t[1] = (uint8_t)(s.steps >> 8);
    t[2] = (uint8_t)(s.inactive & 0xFF); t[3] = (uint8_t)(s.inactive >> 8);
}

// index of the agent standing on (x, y), -1 if none  (== upstream grid[_LAYER_AGENTS] - 1)
template <int P>
MARL_HD int rw_agent_at(const RwState<P>& s, int x, int y) {
    int r = -1;
#pragma unroll
    for (int p = 0; p < P; ++p) r = (s.ax[p] == x && s.ay[p] == y) ? p : r;
    return r;
}

template <int P>
MARL_HD bool rw_requested(const RwParams& q, const RwState<P>& s, int shelf) {
    bool r = false;
#pragma unroll
    for (int k = 0; k < 2 * P; ++k) r = r || (k < q.queue_size && s.rq[k] == shelf);
    return r;
}

// Warehouse.reset: shelves on every non-highway cell (ids in row-major order), agents on distinct uniform cells with
// uniform directions, queue_size distinct requested shelves; draws from `rng` in that order.
template <int P>
MARL_HD void rw_reset(const RwParams& q, RwState<P>& s, const RwGrid& grid, DrawStream& rng) {
    int id = 0;
    for (int y = 0; y < q.rows; ++y)
        for (int x = 0; x < q.cols; ++x) grid.set(y * q.cols + x, rw_is_highway(q, x, y) ? 0 : ++id);
    const int cells = q.rows * q.cols;
    int cell[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        bool fresh;
        int c;
        do {
            c = rng.integers(0, cells);
            fresh = true;
#pragma unroll
            for (int o = 0; o < P; ++o) fresh = fresh && !(o < p && cell[o] == c);
        } while (!fresh);
        cell[p] = c;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        s.ax[p] = cell[p] % q.cols;
        s.ay[p] = cell[p] / q.cols;
        s.ad[p] = rng.integers(0, 4);
        s.ac[p] = 0;
        s.adel[p] = 0;
    }
#pragma unroll
    for (int k = 0; k < 2 * P; ++k) s.rq[k] = 0;
#pragma unroll
    for (int k = 0; k < 2 * P; ++k) {
        if (k < q.queue_size) {
            bool fresh;
            int c;
            do {
                c = rng.integers(1, q.n_shelves + 1);
                fresh = true;
#pragma unroll
                for (int o = 0; o < 2 * P; ++o) fresh = fresh && !(o < k && s.rq[o] == c);
            } while (!fresh);
            s.rq[k] = c;
        }
    }
    s.steps = 0;
    s.inactive = 0;
}

// nibble-packed per-agent tables (P <= 8: eight nibbles = ONE 32-bit word; 0xF = none).  32-bit on purpose: the 64-bit variable
// shifts of a wider table are multi-pass instructions on the vector ALU, and these run ~100 times per env step.
MARL_HD int rw_nib(uint32_t w, int i) { return (int)((w >> (4 * i)) & 0xFu); }
MARL_HD uint32_t rw_set_nib(uint32_t w, int i, int v) { return (w & ~(0xFu << (4 * i))) | ((uint32_t)(v & 0xF) << (4 * i)); }

// committed-agent bit mask from the per-agent "agent on my target cell" table (0xF = free cell, self = stationary) and the
// target cells; see the header comment and oracle/rware.py resolve_rule.  Written as selects, not branches: per-lane `if`s become
// exec-mask regions (s_and_saveexec + s_cbranch) whose 64-bit masks overflow the scalar registers (round 4: the step spent most
// of its time on v_readlane / v_writelane spill traffic and taken branches, not on arithmetic).
template <int P>
MARL_HD uint32_t rw_resolve(uint32_t nxt, const int* tcell) {
    static_assert(P <= 8, "nibble tables sized for <= 8 agents");
    uint32_t committed = 0, in_tree = 0;
    // height[j] = longest chain of feeders behind agent j.  It is only ever read for agents of in-trees (below), where it equals the
    // longest walk ending at j - so it is gathered on the chain walks themselves: the walk from i reaches its k-th successor after k
    // moves (upstream's fixed-point passes give the same numbers on in-trees; on cycles they differ, and nobody looks)
    uint32_t height = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        // follow the chain from i: back at i = on a cycle, whose length decides (a 2-cycle is a swap: nobody moves);
        // at a free cell = i sits in an in-tree; neither = i feeds a cycle it is not on (and cannot move)
        int j = rw_nib(nxt, i), n = 1;
        const bool self = j == i;  // stationary: feeds nobody
#pragma unroll
        for (int it = 0; it < P; ++it) {
            const bool at_agent = (j != 0xF) & !self & (j != i);
            const int jj = at_agent ? j : 0;
            const int hj = rw_nib(height, jj);
            height = (at_agent & (it + 1 > hj)) ? rw_set_nib(height, jj, it + 1) : height;
            const bool go = j != 0xF && j != i;
            const int nj = rw_nib(nxt, go ? j : 0);
            j = go ? nj : j;
            n += go ? 1 : 0;
        }
        committed |= (j == i && n != 2) ? 1u << i : 0u;
        in_tree |= (j == 0xF) ? 1u << i : 0u;
    }
    // dag_longest_path walked back from the free cell = at every cell the feeder with the longest chain behind it wins
    // (lowest index on ties), and a winner moves iff its target is free or the agent on it moves
    uint32_t win = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        bool w = ((in_tree >> i) & 1u) != 0;
        const int hi = rw_nib(height, i);
#pragma unroll
        for (int o = 0; o < P; ++o) {
            if (o != i) {
                const int ho = rw_nib(height, o);
                w = w & !((tcell[o] == tcell[i]) & (((in_tree >> o) & 1u) != 0) & ((ho > hi) | ((ho == hi) & (o < i))));
            }
        }
        win |= (w ? 1u : 0u) << i;
    }
    uint32_t mv = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) mv |= (((win >> i) & 1u) && rw_nib(nxt, i) == 0xF) ? 1u << i : 0u;
#pragma unroll
    for (int pass = 0; pass < P - 1; ++pass) {
#pragma unroll
        for (int i = 0; i < P; ++i) mv |= (((win >> i) & 1u) & ((mv >> rw_nib(nxt, i)) & 1u)) << i;
    }
    return committed | mv;
}

// Warehouse.step.  rew[] are the env's own per-agent rewards (fp64, as upstream's np.zeros accumulates them); `done` =
// max_steps / max_inactivity_steps reached.  `req` is the env's replacement-request stream (STREAM_REQUEST); it is
// positioned at word 8 * step here.
// Shape of the code (round 4): every grid byte the common path needs - each agent's requested cell and its own cell - is read up
// front in one batch of independent loads (upstream's TOGGLE too reads the shelf layer as the PREVIOUS step left it: its grid is
// only recalculated at the end of step()); targets, turns and toggles are selects; the only per-lane branches left are the grid
// writes of moving carriers and the delivery (rare).
// Returns whether a shelf was delivered in this step - the one event that changes the request queue (callers that cache something
// derived from the queue, env_traits.h's request bit set, rebuild it on exactly that).
template <int P>
MARL_HD bool rw_step(const RwParams& q, RwState<P>& s, const RwGrid& grid, const int* act_in, double* rew, bool& done, DrawStream& req) {
    int act[P], tx[P], ty[P], tcell[P], own[P], g_t[P], g_own[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int a = act_in[p];
        a = (a < 0 || a > RW_TOGGLE) ? RW_NOOP : a;
        const int d = s.ad[p], fw = a == RW_FORWARD ? 1 : 0;
        int x = s.ax[p] + fw * ((d == RW_DRIGHT ? 1 : 0) - (d == RW_DLEFT ? 1 : 0));
        int y = s.ay[p] + fw * ((d == RW_DOWN ? 1 : 0) - (d == RW_UP ? 1 : 0));
        x = x < 0 ? 0 : (x > q.cols - 1 ? q.cols - 1 : x);  // Agent.req_location clamps at the walls
        y = y < 0 ? 0 : (y > q.rows - 1 ? q.rows - 1 : y);
        act[p] = a; tx[p] = x; ty[p] = y;
        own[p] = s.ay[p] * q.cols + s.ax[p];
        tcell[p] = y * q.cols + x;
        rew[p] = 0.0;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        g_t[p] = grid.get(tcell[p]);
        g_own[p] = grid.get(own[p]);
    }
    uint32_t nxt = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        // a loaded agent cannot enter a cell with a standing shelf (one another agent carries may move away in time)
        const int o = rw_agent_at(s, tx[p], ty[p]);  // the agent on the requested cell (itself when it stays), -1 = nobody
        int oc = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) oc = k == o ? s.ac[k] : oc;
        const bool blocked = (s.ac[p] != 0) & (tcell[p] != own[p]) & (g_t[p] != 0) & (oc == 0);
        act[p] = blocked ? RW_NOOP : act[p];
        tx[p] = blocked ? s.ax[p] : tx[p];
        ty[p] = blocked ? s.ay[p] : ty[p];
        tcell[p] = blocked ? own[p] : tcell[p];
        nxt = rw_set_nib(nxt, p, blocked ? p : (o < 0 ? 0xF : o));
    }
    const uint32_t committed = rw_resolve<P>(nxt, tcell);
    // moves of loaded agents: clear every vacated cell first, then occupy (trains of carriers)
    bool carry_move[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        act[p] = ((committed >> p) & 1u) ? act[p] : RW_NOOP;  // only FORWARD requests can fail
        carry_move[p] = (act[p] == RW_FORWARD) & (s.ac[p] != 0) & (tcell[p] != own[p]);
        if (carry_move[p]) grid.set(own[p], 0);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int a = act[p], d = s.ad[p];
        const bool fwd = a == RW_FORWARD;
        if (fwd & (s.ac[p] != 0)) grid.set(tcell[p], s.ac[p]);
        const int ax0 = s.ax[p], ay0 = s.ay[p];
        s.ax[p] = fwd ? tx[p] : ax0;
        s.ay[p] = fwd ? ty[p] : ay0;
        // clockwise order UP, RIGHT, DOWN, LEFT; LEFT turns against it.  Direction codes: UP 0, DOWN 1, LEFT 2, RIGHT 3
        const int left = (0x0132 >> (4 * d)) & 3, right = (0x1023 >> (4 * d)) & 3;
        s.ad[p] = a == RW_LEFT ? left : (a == RW_RIGHT ? right : d);
        const bool toggle = a == RW_TOGGLE;  // a toggling agent did not move: (ax0, ay0) is its cell, g_own its shelf byte
        const bool pick = toggle & (s.ac[p] == 0);
        const bool drop = toggle & (s.ac[p] != 0) & !rw_is_highway(q, ax0, ay0);
        rew[p] += (drop & (s.adel[p] != 0) & (q.reward_type == RW_REWARD_TWO_STAGE)) ? 0.5 : 0.0;
        s.adel[p] = drop ? 0 : s.adel[p];
        s.ac[p] = pick ? g_own[p] : (drop ? 0 : s.ac[p]);
    }
    // deliveries at the two goal cells
    bool delivered = false;
    req.idx = 8u * (uint32_t)s.steps;
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int gx = q.cols / 2 - 1 + gi, gy = q.rows - 1;
        const int shelf = grid.get(gy * q.cols + gx);
        if (shelf == 0 || !rw_requested(q, s, shelf)) continue;
        delivered = true;
        // replacement: uniform over the shelves not in the queue, in id order
        int sorted[2 * P];
#pragma unroll
        for (int k = 0; k < 2 * P; ++k) sorted[k] = k < q.queue_size ? s.rq[k] : 0x7FFFFFFF;
#pragma unroll
        for (int i = 0; i < 2 * P; ++i)
#pragma unroll
            for (int j = 0; j < 2 * P - 1; ++j)
                if (sorted[j + 1] < sorted[j]) { const int t = sorted[j]; sorted[j] = sorted[j + 1]; sorted[j + 1] = t; }
        int id = req.integers(0, q.n_shelves - q.queue_size) + 1;
#pragma unroll
        for (int k = 0; k < 2 * P; ++k) id += (sorted[k] <= id) ? 1 : 0;
        bool replaced = false;
#pragma unroll
        for (int k = 0; k < 2 * P; ++k) {
            if (!replaced && k < q.queue_size && s.rq[k] == shelf) { s.rq[k] = id; replaced = true; }
        }
        const int carrier = rw_agent_at(s, gx, gy);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (q.reward_type == RW_REWARD_GLOBAL) rew[p] += 1.0;
            else if (p == carrier) {
                if (q.reward_type == RW_REWARD_INDIVIDUAL) rew[p] += 1.0;
                else { s.adel[p] = 1; rew[p] += 0.5; }
            }
        }
    }
    s.inactive = delivered ? 0 : s.inactive + 1;
    s.steps += 1;
    done = (q.max_inactivity_steps > 0 && s.inactive >= q.max_inactivity_steps) || (q.max_steps > 0 && s.steps >= q.max_steps);
    return delivered;
}

// arr[p] for a RUN-TIME agent index (the agent-per-wave collectors): a select chain over compile-time indices - a dynamically
// indexed private array would move the whole agent state into scratch memory; folds to arr[p] when p is a constant
template <int P>
MARL_HD int rw_pick(const int (&arr)[P], int p) {
    int v = 0;
#pragma unroll
    for (int k = 0; k < P; ++k) v = k == p ? arr[k] : v;
    return v;
}

// agent p's 3x3 sensor window, one code per cell (row-major): bit0 agent present, bits1-2 its direction, bit3 shelf,
// bit4 shelf requested; cells off the grid read as empty (np.pad with zeros)
template <int P>
MARL_HD void rw_window(const RwParams& q, const RwState<P>& s, const RwGrid& grid, int p, int (&code)[9]) {
    const int px = rw_pick<P>(s.ax, p), py = rw_pick<P>(s.ay, p);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int x = px - 1 + c % 3, y = py - 1 + c / 3;
        int v = 0;
        if (x >= 0 && y >= 0 && x < q.cols && y < q.rows) {
            const int o = rw_agent_at(s, x, y);
            if (o >= 0) {
                int d = 0;
#pragma unroll
                for (int k = 0; k < P; ++k) d = k == o ? s.ad[k] : d;
                v |= 1 | (d << 1);
            }
            const int shelf = grid.get(y * q.cols + x);
            if (shelf != 0) v |= 8 | (rw_requested(q, s, shelf) ? 16 : 0);
        }
        code[c] = v;
    }
}

constexpr int RW_OBS_DIM = 8 + 9 * 7;

// The collectors' cheaper route to the same observation.  The 9 window codes packed 5 bits apiece into one word, built from
// the entities instead of per cell: every agent drops its (present, direction) bits into the cell it occupies if that cell
// is inside the window, every in-grid window cell adds the shelf bits, the request test being one probe of a 256-bit set.
template <int P>
struct RwRequested {
    uint64_t w[4];
    MARL_HD void build(const RwParams& q, const RwState<P>& s) {
        w[0] = w[1] = w[2] = w[3] = 0;
#pragma unroll
        for (int k = 0; k < 2 * P; ++k) {
            if (k < q.queue_size) {
                const uint64_t bit = 1ull << (s.rq[k] & 63);
                const int hi = s.rq[k] >> 6;
                w[0] |= hi == 0 ? bit : 0; w[1] |= hi == 1 ? bit : 0; w[2] |= hi == 2 ? bit : 0; w[3] |= hi == 3 ? bit : 0;
            }
        }
    }
    MARL_HD int test(int shelf) const {
        const int hi = shelf >> 6;
        const uint64_t v = hi == 0 ? w[0] : (hi == 1 ? w[1] : (hi == 2 ? w[2] : w[3]));
        return (int)((v >> (shelf & 63)) & 1ull);
    }
};

template <int P>
MARL_HD uint64_t rw_window_word(const RwParams& q, const RwState<P>& s, const RwGrid& grid, const RwRequested<P>& rq, int p) {
    uint64_t word = 0;
    const int x0 = rw_pick<P>(s.ax, p) - 1, y0 = rw_pick<P>(s.ay, p) - 1;
#pragma unroll
    for (int o = 0; o < P; ++o) {
        const unsigned dx = (unsigned)(s.ax[o] - x0), dy = (unsigned)(s.ay[o] - y0);
        if (dx < 3u && dy < 3u) word |= (uint64_t)(1 | (s.ad[o] << 1)) << (5 * (dy * 3 + dx));
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int x = x0 + c % 3, y = y0 + c / 3;
        if (x >= 0 && y >= 0 && x < q.cols && y < q.rows) {
            const int shelf = grid.get(y * q.cols + x);
            if (shelf != 0) word |= (uint64_t)(8 | (rq.test(shelf) << 4)) << (5 * c);
        }
    }
    return word;
}

// element d (any runtime value in [0, 71)) of agent p's observation from the packed window
template <int P>
MARL_HD float rw_obs_elem_word(const RwParams& q, const RwState<P>& s, int p, uint64_t word, int d) {
    if (d < 8) {
        const int ax = rw_pick<P>(s.ax, p), ay = rw_pick<P>(s.ay, p);
        const int v = d == 0 ? ax : (d == 1 ? ay : (d == 2 ? (rw_pick<P>(s.ac, p) != 0) : (d == 7 ? (int)rw_is_highway(q, ax, ay) : (rw_pick<P>(s.ad, p) == d - 3))));
        return (float)v;
    }
    const int e = d - 8, c = (e * 37) >> 8, f = e - 7 * c;  // e / 7 and e % 7 for e < 63
    const int v = (int)(word >> (5 * c)) & 31;
    const int has = v & 1, dir = has ? (v >> 1) & 3 : 0;
    const int r = f == 0 ? has : (f <= 4 ? (dir == f - 1) : (f == 5 ? (v >> 3) & 1 : (v >> 4) & 1));
    return (float)r;
}



// The fused collectors' route (round 4): the 63 window features of agent p as ONE bit mask - bit 7c + f = feature f of window cell
// c, the order of the flattened observation after its 8 leading entries - so that a lane's element of a k-step is one bit-field
// extract.  Built from the entities: every cell starts as "no agent, direction one-hot(0)" (bit 7c + 1), an agent inside the window
// replaces that by (present, one-hot(direction)); the 9 shelf bytes are read UNCONDITIONALLY from clamped coordinates (nine
// independent loads, one wait - a load under `if (in grid)` is a dependent round trip each) and masked afterwards.
template <int P>
MARL_HD uint64_t rw_obs_bits(const RwParams& q, const RwState<P>& s, const RwGrid& grid, const RwRequested<P>& rq, int p) {
    const int x0 = rw_pick<P>(s.ax, p) - 1, y0 = rw_pick<P>(s.ay, p) - 1;
    int shelf[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int x = x0 + c % 3, y = y0 + c / 3;
        const int cx = x < 0 ? 0 : (x > q.cols - 1 ? q.cols - 1 : x), cy = y < 0 ? 0 : (y > q.rows - 1 ? q.rows - 1 : y);
        shelf[c] = grid.get(cy * q.cols + cx);
    }
    uint64_t bits = 0;
#pragma unroll
    for (int c = 0; c < 9; ++c) bits |= 2ull << (7 * c);
#pragma unroll
    for (int o = 0; o < P; ++o) {
        const unsigned dx = (unsigned)(s.ax[o] - x0), dy = (unsigned)(s.ay[o] - y0);
        const bool in = (dx < 3u) & (dy < 3u);
        const int sh = in ? 7 * (int)(dy * 3 + dx) : 0;
        // clear the default direction bit, then present + one-hot(direction) (direction 0 sets the same bit again)
        bits = in ? ((bits & ~(2ull << sh)) | ((uint64_t)(1u | (2u << s.ad[o])) << sh)) : bits;
    }
    uint32_t lo = 0, hi = 0;  // shelf present (f = 5) and requested (f = 6): bits 7c + 5, 7c + 6 - never straddling bit 32
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int x = x0 + c % 3, y = y0 + c / 3;
        const bool in = (x >= 0) & (y >= 0) & (x < q.cols) & (y < q.rows);
        const int sf = in ? shelf[c] : 0;
        const uint32_t two = (sf != 0 ? 1u : 0u) | ((uint32_t)rq.test(sf) << 1);  // shelf ids start at 1: bit 0 of the set is never set
        if (7 * c + 5 < 32) lo |= two << ((7 * c + 5) & 31);
        else hi |= two << ((7 * c + 5) & 31);
    }
    return bits | (uint64_t)lo | ((uint64_t)hi << 32);
}

// element d of agent p's observation from the feature mask (d any run-time value in [0, 71))
template <int P>
MARL_HD float rw_obs_elem_bits(const RwParams& q, const RwState<P>& s, int p, uint64_t bits, int d) {
    if (d < 8) {
        const int ax = rw_pick<P>(s.ax, p), ay = rw_pick<P>(s.ay, p);
        const int v = d == 0 ? ax : (d == 1 ? ay : (d == 2 ? (rw_pick<P>(s.ac, p) != 0) : (d == 7 ? (int)rw_is_highway(q, ax, ay) : (rw_pick<P>(s.ad, p) == d - 3))));
        return (float)v;
    }
    return (float)((bits >> (d - 8)) & 1ull);
}

// element d of agent p's flattened observation (Warehouse._make_obs, fast path):
//   x, y, carrying, one-hot direction[4], on highway, then per window cell: agent present, one-hot direction[4]
//   (an empty cell reads as direction 0: Discrete(4) flattens to one-hot(0)), shelf present, shelf requested
template <int P>
MARL_HD float rw_obs_elem(const RwParams& q, const RwState<P>& s, int p, const int (&code)[9], int d) {
    if (d < 8) {
        const int ax = rw_pick<P>(s.ax, p), ay = rw_pick<P>(s.ay, p);
        switch (d) {
            case 0: return (float)ax;
            case 1: return (float)ay;
            case 2: return rw_pick<P>(s.ac, p) != 0 ? 1.f : 0.f;
            case 7: return rw_is_highway(q, ax, ay) ? 1.f : 0.f;
            default: return rw_pick<P>(s.ad, p) == d - 3 ? 1.f : 0.f;
        }
    }
    const int c = (d - 8) / 7, f = (d - 8) % 7;
    int v = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) v = k == c ? code[k] : v;
    const bool has = (v & 1) != 0;
    const int dir = (v >> 1) & 3;
    if (f == 0) return has ? 1.f : 0.f;
    if (f <= 4) return (has ? dir : 0) == f - 1 ? 1.f : 0.f;
    if (f == 5) return (v & 8) ? 1.f : 0.f;
    return (v & 16) ? 1.f : 0.f;
}

}  // namespace marl
