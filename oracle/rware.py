"""CPU restatement of the multi-robot warehouse (`rware.warehouse`, class Warehouse) - config 4's env.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED.  `rware` is a third-party, un-vendored, unpinned dependency of the reference (README.md:73
`pip install -U lbforaging rware`); its source is not under /root/reference, it is not installed and cannot be
fetched, and the reference ships no tests or golden vectors for it.  This file restates the package's published
algorithm (Warehouse.__init__ layout, reset, step, _make_obs in the FLATTENED "fast" form, and the
`rware-{tiny,small,medium,large}-{n}ag[-easy|-hard]-v2` registration) in the structure upstream uses (two collision
layers in a dense int grid, Agent / Shelf objects, a request queue, a movement graph).  Parity is anchored on the
reference's call sites, which are the same as for Level-Based Foraging:
  marlbase/utils/envs.py:27-37,82-97   gym.make -> TimeLimit -> RecordEpisodeStatistics (vector and single env)
  marlbase/ac/train.py:30,79           envs.reset() / envs.step(actions)      (config 4's collector)
  marlbase/dqn/train.py:203,217        env.reset() / env.step(actions)

Movement conflicts.  Upstream builds a networkx DiGraph cell -> requested cell per agent, and commits, per weakly
connected component, the agents on its cycle (none for a 2-cycle = a swap) or on its `dag_longest_path`.
`resolve_networkx` below is that code, run on the real networkx (importable here and on the GPU box).
`resolve_rule` is the closed form the HIP kernel implements: every node has out-degree <= 1, so a component is either
one cycle with in-trees hanging off it, or an in-tree draining into one free cell; in the latter the longest path is
found by walking back from the free cell, at every merge taking the feeder with the longest chain behind it.
THE ONE DIFFERENCE: when two feeders tie, upstream's choice is the iteration order of a CPython set of (x, y) tuples
inside `G.subgraph(c).copy()` (it even depends on whether the component is less than half of the graph) - an
implementation accident, not a rule.  `resolve_rule` and the kernel give the cell to the LOWEST-INDEX agent.
tests/test_rware_oracle.py checks that the two resolvers agree on every tie-free case and that on ties the networkx
outcome is always the rule's outcome under some relabelling of the tied agents.

Randomness.  Upstream draws from numpy's global RNG (agent cells and directions and the request queue at reset; a
replacement request whenever a shelf is delivered) and, for the replacement, from `list(set(shelfs) - set(queue))`,
whose order follows object addresses - not reproducible even upstream.  Here, as for Level-Based Foraging, the draws
come from the path's own Philox streams: reset draws from stream 1 (cells, then directions, then requested shelves, by
rejection of duplicates), replacement requests from stream 3 at word 8 * step (candidates in shelf-id order).
"""
from time import perf_counter

import numpy as np

from oracle.philox import DrawStream

NOOP, FORWARD, LEFT, RIGHT, TOGGLE_LOAD = range(5)
UP, DOWN, DLEFT, DRIGHT = range(4)
REWARD_GLOBAL, REWARD_INDIVIDUAL, REWARD_TWO_STAGE = range(3)
STREAM_REQUEST = 3
_LAYER_AGENTS, _LAYER_SHELFS = 0, 1
_WRAP = [UP, DRIGHT, DOWN, DLEFT]  # clockwise

SIZES = {"tiny": (1, 3), "small": (2, 3), "medium": (2, 5), "large": (3, 5)}  # (shelf_rows, shelf_columns)
DIFFICULTY = {"easy": 2.0, "": 1.0, "hard": 0.5}


class Agent:
    def __init__(self, id_, x, y, dir_):
        self.id, self.x, self.y, self.dir = id_, x, y, dir_
        self.req_action = None
        self.carrying_shelf = None
        self.has_delivered = False

    def req_location(self, grid_size):
        if self.req_action != FORWARD:
            return self.x, self.y
        if self.dir == UP:
            return self.x, max(0, self.y - 1)
        if self.dir == DOWN:
            return self.x, min(grid_size[0] - 1, self.y + 1)
        if self.dir == DLEFT:
            return max(0, self.x - 1), self.y
        return min(grid_size[1] - 1, self.x + 1), self.y

    def req_direction(self):
        if self.req_action == RIGHT:
            return _WRAP[(_WRAP.index(self.dir) + 1) % 4]
        if self.req_action == LEFT:
            return _WRAP[(_WRAP.index(self.dir) - 1) % 4]
        return self.dir


class Shelf:
    def __init__(self, id_, x, y):
        self.id, self.x, self.y = id_, x, y


class Warehouse:
    """Restatement of upstream Warehouse (flattened observations, msg_bits = 0)."""

    def __init__(self, shelf_columns, column_height, shelf_rows, n_agents, sensor_range=1, request_queue_size=None,
                 max_inactivity_steps=None, max_steps=500, reward_type=REWARD_INDIVIDUAL, resolver="rule", rng=None):
        assert shelf_columns % 2 == 1, "Only odd number of shelf columns is supported"
        self.grid_size = ((column_height + 1) * shelf_rows + 2, (2 + 1) * shelf_columns + 1)  # (rows, cols)
        self.column_height = column_height
        R, C = self.grid_size
        self.grid = np.zeros((2, R, C), np.int32)
        self.goals = [(C // 2 - 1, R - 1), (C // 2, R - 1)]  # (x, y)
        self.highways = np.zeros((R, C), np.uint8)
        for x in range(C):
            for y in range(R):
                self.highways[y, x] = int(self.highway_func(x, y))
        self.n_agents = n_agents
        self.sensor_range = sensor_range
        self.request_queue_size = n_agents if request_queue_size is None else request_queue_size
        self.max_inactivity_steps = max_inactivity_steps
        self.max_steps = max_steps
        self.reward_type = reward_type
        self.resolver = resolver
        self.rng = rng
        self.req_rng = None
        self.obs_dim = 8 + (2 * sensor_range + 1) ** 2 * 7
        self.n_actions = 5
        self.agents, self.shelfs, self.request_queue = [], [], []
        self._cur_inactive_steps = 0
        self._cur_steps = 0

    def highway_func(self, x, y):
        R, C = self.grid_size
        return (x % 3 == 0 or y % (self.column_height + 1) == 0 or y == R - 1
                or (y > R - (self.column_height + 3) and (x == C // 2 - 1 or x == C // 2)))

    def _is_highway(self, x, y):
        return bool(self.highways[y, x])

    # ------------------------------------------------------------------ reset
    def reset(self, rng=None):
        rng = rng or self.rng
        R, C = self.grid_size
        self._cur_inactive_steps = 0
        self._cur_steps = 0
        self.shelfs = []
        for y in range(R):
            for x in range(C):
                if not self._is_highway(x, y):
                    self.shelfs.append(Shelf(len(self.shelfs) + 1, x, y))
        cells = []
        while len(cells) < self.n_agents:
            c = rng.integers(0, R * C)
            if c not in cells:
                cells.append(c)
        dirs = [rng.integers(0, 4) for _ in range(self.n_agents)]
        self.agents = [Agent(i + 1, c % C, c // C, d) for i, (c, d) in enumerate(zip(cells, dirs))]
        self._recalc_grid()
        ids = []
        while len(ids) < self.request_queue_size:
            s = rng.integers(1, len(self.shelfs) + 1)
            if s not in ids:
                ids.append(s)
        self.request_queue = [self.shelfs[s - 1] for s in ids]
        if isinstance(rng, DrawStream):
            seed = rng.key[0] | (rng.key[1] << 32)
            self.req_rng = DrawStream(seed, rng.env_id, rng.episode, STREAM_REQUEST)
        else:
            self.req_rng = rng
        return tuple(self._make_obs(a) for a in self.agents), {}

    def _recalc_grid(self):
        self.grid[:] = 0
        for s in self.shelfs:
            self.grid[_LAYER_SHELFS, s.y, s.x] = s.id
        for a in self.agents:
            self.grid[_LAYER_AGENTS, a.y, a.x] = a.id

    # ------------------------------------------------------------ observation
    def _make_obs(self, agent):
        sr = self.sensor_range
        pa = np.pad(self.grid[_LAYER_AGENTS], sr, mode="constant")
        ps = np.pad(self.grid[_LAYER_SHELFS], sr, mode="constant")
        agents = pa[agent.y:agent.y + 2 * sr + 1, agent.x:agent.x + 2 * sr + 1].reshape(-1)
        shelfs = ps[agent.y:agent.y + 2 * sr + 1, agent.x:agent.x + 2 * sr + 1].reshape(-1)
        out = [float(agent.x), float(agent.y), float(agent.carrying_shelf is not None)]
        d = [0.0] * 4
        d[agent.dir] = 1.0
        out += d
        out.append(float(self._is_highway(agent.x, agent.y)))
        for id_agent, id_shelf in zip(agents, shelfs):
            if id_agent == 0:
                out += [0.0, 1.0, 0.0, 0.0, 0.0]  # no agent; Discrete(4) direction flattens to one-hot(0)
            else:
                d = [0.0] * 4
                d[self.agents[id_agent - 1].dir] = 1.0
                out += [1.0] + d
            if id_shelf == 0:
                out += [0.0, 0.0]
            else:
                out += [1.0, float(self.shelfs[id_shelf - 1] in self.request_queue)]
        return np.array(out, np.float32)

    # ------------------------------------------------------------------- step
    def _edges(self):
        """(start, target) per agent after the loaded-agent-into-standing-shelf cancellation"""
        edges = []
        for agent in self.agents:
            start = agent.x, agent.y
            target = agent.req_location(self.grid_size)
            blocked = False
            if agent.carrying_shelf and start != target and self.grid[_LAYER_SHELFS, target[1], target[0]]:
                other = self.grid[_LAYER_AGENTS, target[1], target[0]]
                blocked = not (other and self.agents[other - 1].carrying_shelf)
            if blocked:
                agent.req_action = NOOP
                edges.append((start, start))
            else:
                edges.append((start, target))
        return edges

    def resolve_networkx(self, edges):
        import networkx as nx

        G = nx.DiGraph()
        for s, t in edges:
            G.add_edge(s, t)
        committed = set()
        for comp in [G.subgraph(c).copy() for c in nx.weakly_connected_components(G)]:
            try:
                cycle = nx.algorithms.find_cycle(comp)
                if len(cycle) == 2:
                    continue
                for edge in cycle:
                    agent_id = self.grid[_LAYER_AGENTS, edge[0][1], edge[0][0]]
                    if agent_id > 0:
                        committed.add(int(agent_id))
            except nx.NetworkXNoCycle:
                for x, y in nx.algorithms.dag_longest_path(comp):
                    agent_id = self.grid[_LAYER_AGENTS, y, x]
                    if agent_id:
                        committed.add(int(agent_id))
        return committed

    @staticmethod
    def resolve_rule(edges):
        """the kernel's closed form (csrc/rware_core.h rw_resolve); ids are 1-based like upstream's"""
        P = len(edges)
        start_of = {s: i for i, (s, _) in enumerate(edges)}
        nxt = [start_of.get(t, -1) for _, t in edges]  # agent standing on my target cell, -1 = free cell
        committed = set()
        # cycles: follow nxt at most P times; i is on a cycle iff it comes back to itself
        for i in range(P):
            j, n = nxt[i], 1
            while j != -1 and j != i and n <= P:
                j, n = nxt[j], n + 1
            if j == i and n != 2:
                committed.add(i + 1)
        # in-trees draining into a free cell: height = longest chain of feeders behind an agent
        height = [0] * P
        for _ in range(P):
            for i in range(P):
                if nxt[i] >= 0 and nxt[i] != i:
                    height[nxt[i]] = max(height[nxt[i]], height[i] + 1)
        on_cycle_comp = [False] * P
        for i in range(P):
            j, n = i, 0
            while j != -1 and n <= P:
                j, n = nxt[j], n + 1
            on_cycle_comp[i] = j != -1
        sinks = []
        for i, (_, t) in enumerate(edges):
            if nxt[i] == -1 and t not in sinks:
                sinks.append(t)
        for cell in sinks:
            feeders = [i for i, (_, t) in enumerate(edges) if t == cell and not on_cycle_comp[i]]
            while feeders:
                best = max(feeders, key=lambda i: (height[i], -i))
                committed.add(best + 1)
                feeders = [i for i in range(P) if nxt[i] == best and i != best]
        return committed

    def step(self, actions):
        assert len(actions) == len(self.agents)
        for agent, action in zip(self.agents, actions):
            assert 0 <= int(action) < 5
            agent.req_action = int(action)
        edges = self._edges()
        committed = self.resolve_networkx(edges) if self.resolver == "networkx" else self.resolve_rule(edges)
        for agent in self.agents:
            if agent.id not in committed:
                assert agent.req_action == FORWARD
                agent.req_action = NOOP
        rewards = np.zeros(self.n_agents)
        for agent in self.agents:
            if agent.req_action == FORWARD:
                agent.x, agent.y = agent.req_location(self.grid_size)
                if agent.carrying_shelf:
                    agent.carrying_shelf.x, agent.carrying_shelf.y = agent.x, agent.y
            elif agent.req_action in (LEFT, RIGHT):
                agent.dir = agent.req_direction()
            elif agent.req_action == TOGGLE_LOAD and not agent.carrying_shelf:
                shelf_id = self.grid[_LAYER_SHELFS, agent.y, agent.x]
                if shelf_id:
                    agent.carrying_shelf = self.shelfs[shelf_id - 1]
            elif agent.req_action == TOGGLE_LOAD and agent.carrying_shelf:
                if not self._is_highway(agent.x, agent.y):
                    agent.carrying_shelf = None
                    if agent.has_delivered and self.reward_type == REWARD_TWO_STAGE:
                        rewards[agent.id - 1] += 0.5
                    agent.has_delivered = False
        self._recalc_grid()

        shelf_delivered = False
        if isinstance(self.req_rng, DrawStream):
            self.req_rng.idx = 8 * self._cur_steps
        for gx, gy in self.goals:
            shelf_id = self.grid[_LAYER_SHELFS, gy, gx]
            if not shelf_id:
                continue
            shelf = self.shelfs[shelf_id - 1]
            if shelf not in self.request_queue:
                continue
            shelf_delivered = True
            queued = {s.id for s in self.request_queue}
            candidates = [s for s in self.shelfs if s.id not in queued]  # shelf-id order (see module docstring)
            new_request = candidates[self.req_rng.integers(0, len(candidates))]
            self.request_queue[self.request_queue.index(shelf)] = new_request
            if self.reward_type == REWARD_GLOBAL:
                rewards += 1
            elif self.reward_type == REWARD_INDIVIDUAL:
                rewards[self.grid[_LAYER_AGENTS, gy, gx] - 1] += 1
            else:
                agent_id = self.grid[_LAYER_AGENTS, gy, gx]
                self.agents[agent_id - 1].has_delivered = True
                rewards[agent_id - 1] += 0.5
        if shelf_delivered:
            self._cur_inactive_steps = 0
        else:
            self._cur_inactive_steps += 1
        self._cur_steps += 1
        done = bool((self.max_inactivity_steps and self._cur_inactive_steps >= self.max_inactivity_steps)
                    or (self.max_steps and self._cur_steps >= self.max_steps))
        return tuple(self._make_obs(a) for a in self.agents), list(rewards), done, False, {}

    # --------------------------------------------------------- parity injection
    def get_state(self):
        """the HIP env's per-env record (csrc/rware_core.h): shelf-layer grid, agents, queue, counters"""
        return dict(grid=self.grid[_LAYER_SHELFS].astype(np.uint8).copy(),
                    agents=np.array([[a.x, a.y, a.dir, a.carrying_shelf.id if a.carrying_shelf else 0, int(a.has_delivered)]
                                     for a in self.agents], np.uint8),
                    queue=np.array([s.id for s in self.request_queue], np.uint8),
                    steps=self._cur_steps, inactive=self._cur_inactive_steps)

    def set_state(self, grid, agents, queue, steps=0, inactive=0):
        R, C = self.grid_size
        n = int((np.asarray(grid) > 0).sum())
        self.shelfs = [None] * n
        for y in range(R):
            for x in range(C):
                if grid[y][x]:
                    self.shelfs[int(grid[y][x]) - 1] = Shelf(int(grid[y][x]), x, y)
        self.agents = []
        for i, (x, y, d, carry, deliv) in enumerate(np.asarray(agents).tolist()):
            a = Agent(i + 1, x, y, d)
            a.carrying_shelf = self.shelfs[carry - 1] if carry else None
            a.has_delivered = bool(deliv)
            self.agents.append(a)
        self.request_queue = [self.shelfs[int(s) - 1] for s in queue]
        self._cur_steps, self._cur_inactive_steps = int(steps), int(inactive)
        self._recalc_grid()


def parse_env_name(name):
    """'rware:rware-tiny-4ag[-easy|-hard]-v2' (also 'rware:tiny-4ag') -> upstream registration kwargs"""
    parts = [p for p in name.split(":")[-1].split("-") if p not in ("rware",) and not (p.startswith("v") and p[1:].isdigit())]
    size = next(p for p in parts if p in SIZES)
    agents = int(next(p for p in parts if p.endswith("ag") and p[:-2].isdigit())[:-2])
    diff = next((p for p in parts if p in ("easy", "hard")), "")
    return dict(column_height=8, shelf_rows=SIZES[size][0], shelf_columns=SIZES[size][1], n_agents=agents, sensor_range=1,
                request_queue_size=int(agents * DIFFICULTY[diff]), max_inactivity_steps=None, max_steps=500,
                reward_type=REWARD_INDIVIDUAL)
