"""CPU restatement of Level-Based Foraging (`lbforaging.foraging.environment`,
class ForagingEnv) plus the marlbase wrapper stack that sits on top of it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED.  `lbforaging` is a third-party, un-vendored, unpinned dependency
of the reference (README.md:73 `pip install -U lbforaging rware`; the `-v3` ids in
README.md:80,85 imply the Gymnasium-era package, lbforaging >= 2.0).  Its source is
not under /root/reference, it is not installed and cannot be fetched, and the
reference ships no tests or golden vectors for it.  This file restates the
package's published algorithm (ForagingEnv.reset / step / _make_gym_obs and the
`Foraging-{s}x{s}-{p}p-{f}f[-coop][-2s]-v3` registration) in the same structure
upstream uses (dense int field, Player objects, collision dict, loading set), so
that a later check against a real install is a line-by-line read.  Parity is
anchored on the reference's own call sites:
  marlbase/utils/envs.py:90-97,111   gym.make -> TimeLimit -> RecordEpisodeStatistics -> reset(seed)
  marlbase/dqn/train.py:203,217      env.reset() / env.step(actions)
  marlbase/utils/wrappers.py:13-45   RecordEpisodeStatistics
  marlbase/utils/wrappers.py:106-108 CooperativeReward
Every constant upstream fixes in its registration is a constructor argument here.

The dynamics are written against a tiny RNG surface (`integers(lo, hi)`,
`permutation(n)`): pass `numpy.random.default_rng(seed)` (what gymnasium's
`seeding.np_random` builds) for the upstream-style stream, or
`oracle.philox.DrawStream` for the stream the HIP reset kernel uses.
"""
from collections import defaultdict
from time import perf_counter

import numpy as np

NONE, NORTH, SOUTH, WEST, EAST, LOAD = range(6)


class Player:
    def __init__(self):
        self.position = None
        self.level = None
        self.score = 0
        self.reward = 0

    def setup(self, position, level):
        self.position = position
        self.level = level
        self.score = 0


class ForagingEnv:
    """Restatement of upstream ForagingEnv (vector observations only)."""

    def __init__(
        self,
        players,
        field_size,
        max_num_food,
        sight=None,
        max_episode_steps=50,
        force_coop=False,
        min_player_level=1,
        max_player_level=2,
        min_food_level=1,
        max_food_level=None,
        normalize_reward=True,
        penalty=0.0,
        rng=None,
    ):
        self.players = [Player() for _ in range(players)]
        self.n_agents = players
        self.field = np.zeros(field_size, np.int32)
        self.max_num_food = max_num_food
        self.sight = max(field_size) if sight is None else sight
        self._max_episode_steps = max_episode_steps
        self.force_coop = force_coop
        self.min_player_level = min_player_level
        self.max_player_level = max_player_level
        self.min_food_level = min_food_level
        self.max_food_level = max_food_level
        self._normalize_reward = normalize_reward
        self.penalty = penalty
        self._food_spawned = 0.0
        self._game_over = None
        self._valid_actions = None
        self.current_step = 0
        self.np_random = rng if rng is not None else np.random.default_rng()
        self.obs_dim = 3 * (max_num_food + players)

    # ---- geometry helpers (upstream names) -------------------------------
    @property
    def rows(self):
        return self.field.shape[0]

    @property
    def cols(self):
        return self.field.shape[1]

    def neighborhood(self, row, col, distance=1, ignore_diag=False):
        if not ignore_diag:
            return self.field[
                max(row - distance, 0) : min(row + distance + 1, self.rows),
                max(col - distance, 0) : min(col + distance + 1, self.cols),
            ]
        return (
            self.field[
                max(row - distance, 0) : min(row + distance + 1, self.rows), col
            ].sum()
            + self.field[
                row, max(col - distance, 0) : min(col + distance + 1, self.cols)
            ].sum()
        )

    def adjacent_food(self, row, col):
        return (
            self.field[max(row - 1, 0), col]
            + self.field[min(row + 1, self.rows - 1), col]
            + self.field[row, max(col - 1, 0)]
            + self.field[row, min(col + 1, self.cols - 1)]
        )

    def adjacent_food_location(self, row, col):
        # upstream tests `row > 1` / `col > 1` (not `> 0`); food never spawns on
        # the border, so the two are equivalent on reachable states.
        if row > 1 and self.field[row - 1, col] > 0:
            return row - 1, col
        elif row < self.rows - 1 and self.field[row + 1, col] > 0:
            return row + 1, col
        elif col > 1 and self.field[row, col - 1] > 0:
            return row, col - 1
        elif col < self.cols - 1 and self.field[row, col + 1] > 0:
            return row, col + 1
        return None

    def adjacent_players(self, row, col):
        return [
            p
            for p in self.players
            if (abs(p.position[0] - row) == 1 and p.position[1] == col)
            or (abs(p.position[1] - col) == 1 and p.position[0] == row)
        ]

    def _is_empty_location(self, row, col):
        if self.field[row, col] != 0:
            return False
        for a in self.players:
            if a.position and row == a.position[0] and col == a.position[1]:
                return False
        return True

    # ---- reset -------------------------------------------------------------
    def spawn_players(self):
        self.np_random.permutation(len(self.players))  # bounds arrays are uniform
        for player in self.players:
            attempts = 0
            player.reward = 0
            while attempts < 1000:
                row = int(self.np_random.integers(0, self.rows))
                col = int(self.np_random.integers(0, self.cols))
                if self._is_empty_location(row, col):
                    player.setup(
                        (row, col),
                        int(
                            self.np_random.integers(
                                self.min_player_level, self.max_player_level + 1
                            )
                        ),
                    )
                    break
                attempts += 1

    def spawn_food(self, max_num_food, min_level, max_level):
        food_count = 0
        attempts = 0
        if self.force_coop:
            min_level = max_level
        self.np_random.permutation(max_num_food)  # bounds arrays are uniform
        while food_count < max_num_food and attempts < 1000:
            attempts += 1
            row = int(self.np_random.integers(1, self.rows - 1))
            col = int(self.np_random.integers(1, self.cols - 1))
            if (
                self.neighborhood(row, col).sum() > 0
                or self.neighborhood(row, col, distance=2, ignore_diag=True) > 0
                or not self._is_empty_location(row, col)
            ):
                continue
            self.field[row, col] = (
                min_level
                if min_level == max_level
                else int(self.np_random.integers(min_level, max_level + 1))
            )
            food_count += 1
        self._food_spawned = int(self.field.sum())

    def reset(self, rng=None):
        if rng is not None:
            self.np_random = rng
        self.field = np.zeros(self.field.shape, np.int32)
        for p in self.players:
            p.position = None
        self.spawn_players()
        player_levels = sorted(p.level for p in self.players)
        self.spawn_food(
            self.max_num_food,
            self.min_food_level,
            self.max_food_level
            if self.max_food_level is not None
            else sum(player_levels[:3]),
        )
        self.current_step = 0
        self._game_over = False
        self._gen_valid_moves()
        return self._make_gym_obs(), {}

    # ---- step --------------------------------------------------------------
    def _is_valid_action(self, player, action):
        r, c = player.position
        if action == NONE:
            return True
        elif action == NORTH:
            return r > 0 and self.field[r - 1, c] == 0
        elif action == SOUTH:
            return r < self.rows - 1 and self.field[r + 1, c] == 0
        elif action == WEST:
            return c > 0 and self.field[r, c - 1] == 0
        elif action == EAST:
            return c < self.cols - 1 and self.field[r, c + 1] == 0
        elif action == LOAD:
            return self.adjacent_food(r, c) > 0
        raise ValueError("Undefined action")

    def _gen_valid_moves(self):
        self._valid_actions = {
            p: [a for a in range(6) if self._is_valid_action(p, a)]
            for p in self.players
        }

    def step(self, actions, pop_order=None):
        """`pop_order`: optional list of player indices fixing the order in which
        loading players are popped (upstream pops a python set, i.e. hash order);
        the outcome does not depend on it on reachable states (tested)."""
        self.current_step += 1
        for p in self.players:
            p.reward = 0

        actions = [
            int(a) if int(a) in self._valid_actions[p] else NONE
            for p, a in zip(self.players, actions)
        ]

        loading_players = []
        collisions = defaultdict(list)
        for player, action in zip(self.players, actions):
            r, c = player.position
            if action == NONE:
                collisions[(r, c)].append(player)
            elif action == NORTH:
                collisions[(r - 1, c)].append(player)
            elif action == SOUTH:
                collisions[(r + 1, c)].append(player)
            elif action == WEST:
                collisions[(r, c - 1)].append(player)
            elif action == EAST:
                collisions[(r, c + 1)].append(player)
            elif action == LOAD:
                collisions[(r, c)].append(player)
                loading_players.append(player)

        # a cell claimed by more than one player is reached by none of them
        for k, v in collisions.items():
            if len(v) > 1:
                continue
            v[0].position = k

        if pop_order is not None:
            loading_players = [
                self.players[i] for i in pop_order if self.players[i] in loading_players
            ]
        while loading_players:
            player = loading_players.pop(0)
            loc = self.adjacent_food_location(*player.position)
            if loc is None:  # unreachable upstream (would raise TypeError)
                continue
            frow, fcol = loc
            food = int(self.field[frow, fcol])
            adj_players = self.adjacent_players(frow, fcol)
            adj_players = [
                p for p in adj_players if p in loading_players or p is player
            ]
            adj_player_level = sum(a.level for a in adj_players)
            loading_players = [p for p in loading_players if p not in adj_players]
            if adj_player_level < food:
                for a in adj_players:
                    a.reward -= self.penalty
                continue
            for a in adj_players:
                a.reward = float(a.level * food)
                if self._normalize_reward:
                    a.reward = a.reward / float(adj_player_level * self._food_spawned)
            self.field[frow, fcol] = 0

        self._game_over = bool(
            self.field.sum() == 0 or self._max_episode_steps <= self.current_step
        )
        self._gen_valid_moves()
        for p in self.players:
            p.score += p.reward
        rewards = [p.reward for p in self.players]
        return self._make_gym_obs(), rewards, self._game_over, False, {}

    # ---- observations ------------------------------------------------------
    def _transform_to_neighborhood(self, center, sight, position):
        return (
            position[0] - center[0] + min(sight, center[0]),
            position[1] - center[1] + min(sight, center[1]),
        )

    def _make_gym_obs(self):
        nobs = []
        F = self.max_num_food
        for player in self.players:
            obs = np.zeros(self.obs_dim, dtype=np.float32)
            seen = []
            for a in self.players:
                pos = self._transform_to_neighborhood(
                    player.position, self.sight, a.position
                )
                if min(pos) >= 0 and max(pos) <= 2 * self.sight:
                    seen.append((pos, a.level, a is player))
            seen = [s for s in seen if s[2]] + [s for s in seen if not s[2]]
            field = self.neighborhood(*player.position, self.sight)
            for i in range(F):
                obs[3 * i : 3 * i + 3] = (-1, -1, 0)
            for i, (y, x) in enumerate(zip(*np.nonzero(field))):
                obs[3 * i : 3 * i + 3] = (y, x, field[y, x])
            for i in range(len(self.players)):
                obs[3 * F + 3 * i : 3 * F + 3 * i + 3] = (-1, -1, 0)
            for i, (pos, lvl, _) in enumerate(seen):
                obs[3 * F + 3 * i : 3 * F + 3 * i + 3] = (pos[0], pos[1], lvl)
            nobs.append(obs)
        return tuple(nobs)

    # ---- packed-state bridge (parity injection; not upstream) --------------
    def get_state(self):
        """(foods[F,3] row-major (r,c,lvl) padded with level 0, players[P,3],
        current_step, food_spawned) - the fields the HIP env keeps in HBM."""
        foods = np.zeros((self.max_num_food, 3), np.int32)
        for i, (y, x) in enumerate(zip(*np.nonzero(self.field))):
            foods[i] = (y, x, self.field[y, x])
        players = np.array(
            [(p.position[0], p.position[1], p.level) for p in self.players], np.int32
        )
        return foods, players, int(self.current_step), int(self._food_spawned)

    def set_state(self, foods, players, current_step, food_spawned):
        self.field = np.zeros(self.field.shape, np.int32)
        for r, c, l in np.asarray(foods).reshape(-1, 3):
            if l > 0:
                self.field[r, c] = l
        for p, (r, c, l) in zip(self.players, np.asarray(players).reshape(-1, 3)):
            p.position = (int(r), int(c))
            p.level = int(l)
            p.reward = 0
        self.current_step = int(current_step)
        self._food_spawned = int(food_spawned)
        self._game_over = False
        self._gen_valid_moves()


def parse_env_name(name):
    """'lbforaging:Foraging[-grid]-8x8-2p-3f[-coop][-2s][-pen]-v3' -> ctor kwargs
    (upstream registration table; max_player_level 2 for v3, 3 for v2)."""
    base = name.split(":")[-1]
    parts = base.split("-")
    assert parts[0].startswith("Foraging"), name
    version = parts[-1]
    size = next(p for p in parts if "x" in p and p.replace("x", "").isdigit())
    s = int(size.split("x")[0])
    p = int(next(q for q in parts if q.endswith("p") and q[:-1].isdigit())[:-1])
    f = int(next(q for q in parts if q.endswith("f") and q[:-1].isdigit())[:-1])
    return dict(
        players=p,
        field_size=(s, s),
        max_num_food=f,
        sight=2 if "2s" in parts else s,
        max_episode_steps=50,
        force_coop="coop" in parts,
        min_player_level=1,
        max_player_level=2 if version == "v3" else 3,
        penalty=0.1 if "pen" in parts else 0.0,
    )


class MarlbaseEnv:
    """ForagingEnv under the reference's wrapper stack (utils/envs.py:93-109):
    TimeLimit(time_limit) -> RecordEpisodeStatistics -> [CooperativeReward]."""

    def __init__(self, name, time_limit, cooperative=False, rng=None, standardise_rewards=False, observe_id=False, **overrides):
        if "rware" in name:  # the warehouse env sits under the same wrapper stack (utils/envs.py:82-97)
            from oracle import rware

            kw = rware.parse_env_name(name)
            kw.update(overrides)
            self.env = rware.Warehouse(rng=rng, **kw)
        else:
            kw = parse_env_name(name)
            kw.update(overrides)
            self.env = ForagingEnv(rng=rng, **kw)
        self.n_agents = self.env.n_agents
        self.time_limit = time_limit
        self.cooperative = cooperative
        self.standardise_rewards = standardise_rewards
        self.observe_id = observe_id  # ObserveID (utils/wrappers.py:73-103): np.eye(P) row in front of every observation
        # StandardiseReward state (utils/wrappers.py:111-117): fp32 arrays + a python int; lives as long as the env object
        self.sr_sumw = np.zeros(self.n_agents, np.float32)
        self.sr_wmean = np.zeros(self.n_agents, np.float32)
        self.sr_t = np.zeros(self.n_agents, np.float32)
        self.sr_n = 0
        self._elapsed = 0
        self.episode_reward = 0
        self.episode_length = 0
        self.t0 = perf_counter()

    def _ids(self, obs):
        if not self.observe_id:
            return obs
        eye = np.eye(self.n_agents, dtype=np.float32)
        return tuple(np.concatenate((eye[p], o)) for p, o in enumerate(obs))

    def reset(self, rng=None):
        obs, info = self.env.reset(rng)
        obs = self._ids(obs)
        self._elapsed = 0
        self.episode_reward = 0  # wrappers.py:26 (int 0, becomes float32 array)
        self.episode_length = 0
        self.t0 = perf_counter()
        return obs, info

    def step(self, actions):
        obs, reward, done, truncated, info = self.env.step(actions)
        obs = self._ids(obs)
        self._elapsed += 1  # gymnasium TimeLimit
        if self.time_limit and self._elapsed >= self.time_limit:
            truncated = True
        # wrappers.py:31-45 (sits inside CooperativeReward: raw per-agent rewards)
        self.episode_reward = self.episode_reward + np.array(reward, dtype=np.float32)
        self.episode_length += 1
        if done or truncated:
            info["episode_returns"] = self.episode_reward
            for i, r in enumerate(self.episode_reward):
                info[f"agent{i}/episode_returns"] = r
            info["episode_length"] = self.episode_length
            info["episode_time"] = perf_counter() - self.t0
        if self.standardise_rewards:  # wrappers.py:118-142 (streaming mean / variance, numpy dtype promotion as there)
            q = reward - self.sr_wmean
            sumw1 = self.sr_sumw + 1.0
            r = q * 1.0 / sumw1
            self.sr_wmean += r
            self.sr_t += q * r * self.sr_sumw
            self.sr_sumw = sumw1
            self.sr_n += 1
            if self.sr_n > 1:
                var = (self.sr_t * self.sr_n) / (self.sr_sumw * (self.sr_n - 1))
                reward = (reward - self.sr_wmean) / (np.sqrt(var) + 1e-6)
        if self.cooperative:  # wrappers.py:106-108
            reward = self.n_agents * [sum(reward)]
        return obs, reward, done, truncated, info
