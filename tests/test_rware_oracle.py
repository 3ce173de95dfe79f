"""Multi-robot warehouse (config 4's env; parity unpinned: the rware package is absent, see oracle/rware.py).
CPU: (1) the movement-conflict rule the kernel implements against upstream's networkx resolution run on the real
networkx - equal on every tie-free case, and on ties the networkx outcome is the rule's under some relabelling;
(2) 22 hand-derived known-answer transitions, each run through the oracle AND the C++ core; (3) the env core the HIP kernels inline (csrc/rware_core.h, built with
g++) against the oracle: reset (same Philox stream) and step (same state, same joint action) agree bit for bit -
state records, observations, rewards, done / truncated."""
import ctypes
import itertools
import random

import numpy as np
import pytest

from oracle import rware as rw
from oracle.lbf import MarlbaseEnv
from oracle.philox import DrawStream
from tests.helpers import host_shim, ptr

TINY4 = "rware:rware-tiny-4ag-v2"


def make(name=TINY4, **over):
    return rw.Warehouse(**dict(rw.parse_env_name(name), **over))


def test_registration_and_layout():
    kw = rw.parse_env_name(TINY4)
    assert kw == dict(column_height=8, shelf_rows=1, shelf_columns=3, n_agents=4, sensor_range=1, request_queue_size=4,
                      max_inactivity_steps=None, max_steps=500, reward_type=rw.REWARD_INDIVIDUAL)
    assert rw.parse_env_name("rware:tiny-4ag") == kw
    assert rw.parse_env_name("rware-small-2ag-easy-v2")["request_queue_size"] == 4
    assert rw.parse_env_name("rware-medium-8ag-hard-v2")["request_queue_size"] == 4
    e = make()
    assert e.grid_size == (11, 10) and e.obs_dim == 71 and e.goals == [(4, 10), (5, 10)]
    e.reset(DrawStream(0, 0, 0))
    assert len(e.shelfs) == 32
    # shelves stand on the two outer 2 x 8 blocks; the middle block is the delivery lane
    assert sorted({s.x for s in e.shelfs}) == [1, 2, 7, 8] and sorted({s.y for s in e.shelfs}) == list(range(1, 9))
    assert make("rware-small-2ag-v2").grid_size == (20, 10) and make("rware-large-2ag-v2").grid_size == (29, 16)


def _relabellings(edges):
    outs = set()
    for perm in itertools.permutations(range(len(edges))):
        c = rw.Warehouse.resolve_rule([edges[i] for i in perm])
        outs.add(frozenset(perm[k - 1] + 1 for k in c))
    return outs


def test_conflict_rule_matches_networkx():
    pytest.importorskip("networkx")
    rng = random.Random(1)
    ties = agree = 0
    for trial in range(600):
        e = make()
        e.reset(DrawStream(trial, 0, 0))
        x0, y0 = rng.randrange(0, 8), rng.randrange(0, 9)
        cells = rng.sample([(x0 + i, y0 + j) for i in range(3) for j in range(3)], 4)  # packed: conflicts are frequent
        st = e.get_state()
        ag = st["agents"].copy()
        for i, (x, y) in enumerate(cells):
            ag[i] = (x, y, rng.randrange(4), 0, 0)
        e.set_state(st["grid"], ag, st["queue"])
        for a in e.agents:
            a.req_action = rw.FORWARD if rng.random() < 0.8 else rng.randrange(5)
        edges = e._edges()
        nxc, rule = frozenset(e.resolve_networkx(edges)), frozenset(e.resolve_rule(edges))
        st_, tg_ = (np.array([c[0] + 10 * c[1] for c in col], np.int32) for col in zip(*edges))
        mask = host_shim().host_rw_resolve(4, ptr(st_), ptr(tg_))  # the C++ rule the kernels inline
        assert {i + 1 for i in range(4) if mask >> i & 1} == set(rule), (edges, mask, rule)
        outs = _relabellings(edges)
        assert nxc in outs, (edges, nxc, outs)
        if len(outs) == 1:
            assert nxc == rule, (edges, nxc, rule)
            agree += 1
        else:
            ties += 1
    assert agree > 400 and ties > 10


def test_conflict_rule_cpp_equals_python_rule_dense_8_agents():
    lib = host_shim()
    rng = random.Random(5)
    for trial in range(3000):
        cells = rng.sample([(i, j) for i in range(4) for j in range(3)], 8)  # 8 agents on 12 cells
        edges = []
        for (x, y) in cells:
            dx, dy = rng.choice([(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1), (1, 0), (0, 1)])
            edges.append(((x, y), (min(max(x + dx, 0), 3), min(max(y + dy, 0), 2))))
        st_, tg_ = (np.array([c[0] + 10 * c[1] for c in col], np.int32) for col in zip(*edges))
        mask = lib.host_rw_resolve(8, ptr(st_), ptr(tg_))
        assert {i + 1 for i in range(8) if mask >> i & 1} == rw.Warehouse.resolve_rule(edges), edges


def _state(e, agents, queue=(1, 2, 3, 4), grid=None, steps=0):
    e.reset(DrawStream(0, 0, 0))
    st = e.get_state()
    e.set_state(st["grid"] if grid is None else grid, np.array(agents, np.uint8), np.array(queue, np.uint8), steps=steps)
    e.req_rng = DrawStream(0, 0, 0, rw.STREAM_REQUEST)
    return e


def _pos(e):
    return [(a.x, a.y, a.dir, a.carrying_shelf.id if a.carrying_shelf else 0) for a in e.agents]


class RwCfg(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("n_envs", "n_agents", "rows", "cols", "column_height", "n_shelves", "queue_size",
                                               "max_steps", "max_inactivity_steps", "time_limit", "reward_type", "cooperative")] + [
        ("seed", ctypes.c_uint64)]


def pack_state(e, P):
    st = e.get_state()
    q = np.zeros(2 * P, np.uint8)
    q[:len(st["queue"])] = st["queue"]
    tail = np.array([st["steps"] & 255, st["steps"] >> 8, st["inactive"] & 255, st["inactive"] >> 8], np.uint8)
    rec = np.concatenate([st["grid"].reshape(-1), st["agents"].reshape(-1), q, tail])
    return np.concatenate([rec, np.zeros((-len(rec)) % 4, np.uint8)])


def both(e, acts):
    """e.step(acts) on the oracle AND the same transition through the C++ core the kernels inline (same state, same request
    stream): records, observations, rewards and done must agree bit for bit"""
    lib = host_shim()
    P = e.n_agents
    R, C = e.grid_size
    hc = RwCfg(n_envs=1, n_agents=P, rows=R, cols=C, column_height=e.column_height, n_shelves=len(e.shelfs), queue_size=e.request_queue_size,
               max_steps=e.max_steps or 0, max_inactivity_steps=e.max_inactivity_steps or 0, time_limit=0, reward_type=e.reward_type,
               cooperative=0, seed=0)
    state = pack_state(e, P)[None].copy()
    out = e.step(acts)
    obs, rew = np.zeros((P, 1, 71), np.float32), np.zeros((P, 1), np.float32)
    done, trunc, epi = np.zeros(1, np.uint8), np.zeros(1, np.uint8), np.zeros(1, np.uint32)
    a = np.array(acts, np.int32).reshape(P, 1)
    assert lib.host_rw_step(ctypes.byref(hc), ptr(state), ptr(epi), ptr(a), ptr(obs), ptr(rew), ptr(done), ptr(trunc)) == 0
    np.testing.assert_array_equal(state[0], pack_state(e, P))
    for p in range(P):
        np.testing.assert_array_equal(obs[p, 0], out[0][p])
    np.testing.assert_array_equal(rew[:, 0], np.array(out[1], np.float32))
    assert bool(done[0]) == out[2]
    return out


def test_known_answer_transitions():
    # hand-derived from the rules in oracle/rware.py's header; agents (x, y, dir, carried shelf, has_delivered)
    far = [(9, 0, rw.UP, 0, 0), (9, 2, rw.UP, 0, 0)]  # two bystanders facing the wall / a free cell
    # 1. two agents into the same free cell: the lower index gets it
    e = _state(make(), [(3, 0, rw.DRIGHT, 0, 0), (5, 0, rw.DLEFT, 0, 0)] + far)
    both(e, [1, 1, 0, 0])
    assert _pos(e)[:2] == [(4, 0, rw.DRIGHT, 0), (5, 0, rw.DLEFT, 0)]
    # 2. a swap (2-cycle) moves nobody
    e = _state(make(), [(3, 0, rw.DRIGHT, 0, 0), (4, 0, rw.DLEFT, 0, 0)] + far)
    both(e, [1, 1, 0, 0])
    assert _pos(e)[:2] == [(3, 0, rw.DRIGHT, 0), (4, 0, rw.DLEFT, 0)]
    # 3. a train: the follower moves into the cell its leader vacates
    e = _state(make(), [(3, 0, rw.DRIGHT, 0, 0), (4, 0, rw.DRIGHT, 0, 0)] + far)
    both(e, [1, 1, 0, 0])
    assert _pos(e)[:2] == [(4, 0, rw.DRIGHT, 0), (5, 0, rw.DRIGHT, 0)]
    # 4. the longer chain wins a merge: agents 1->2->free cell (4,0) against agent 0 alone
    e = _state(make(), [(4, 1, rw.UP, 0, 0), (2, 0, rw.DRIGHT, 0, 0), (3, 0, rw.DRIGHT, 0, 0), (9, 5, rw.UP, 0, 0)])
    both(e, [1, 1, 1, 0])
    assert _pos(e)[:3] == [(4, 1, rw.UP, 0), (3, 0, rw.DRIGHT, 0), (4, 0, rw.DRIGHT, 0)]
    # 5. a 4-cycle rotates
    e = _state(make(), [(3, 0, rw.DRIGHT, 0, 0), (4, 0, rw.DOWN, 0, 0), (4, 1, rw.DLEFT, 0, 0), (3, 1, rw.UP, 0, 0)])
    both(e, [1, 1, 1, 1])
    assert [p[:2] for p in _pos(e)] == [(4, 0), (4, 1), (3, 1), (3, 0)]
    # 6. FORWARD into the wall stays (and still blocks the agent behind it)
    e = _state(make(), [(0, 0, rw.DLEFT, 0, 0), (1, 0, rw.DLEFT, 0, 0)] + far)
    both(e, [1, 1, 0, 0])
    assert _pos(e)[:2] == [(0, 0, rw.DLEFT, 0), (1, 0, rw.DLEFT, 0)]
    # 7. turns: LEFT from UP faces LEFT, RIGHT from UP faces RIGHT, RIGHT from LEFT faces UP
    e = _state(make(), [(3, 0, rw.UP, 0, 0), (5, 0, rw.UP, 0, 0), (9, 0, rw.DLEFT, 0, 0), (9, 2, rw.DOWN, 0, 0)])
    both(e, [2, 3, 3, 2])
    assert [p[2] for p in _pos(e)] == [rw.DLEFT, rw.DRIGHT, rw.UP, rw.DRIGHT]
    # 8. load under a shelf, carry it out, cannot unload on a highway, unload in the rack
    e = _state(make(), [(1, 1, rw.UP, 0, 0), (5, 0, rw.UP, 0, 0)] + far)  # shelf 1 stands on (1,1)
    both(e, [4, 0, 0, 0])
    assert _pos(e)[0] == (1, 1, rw.UP, 1)
    both(e, [1, 0, 0, 0])  # up to (1,0): highway row
    assert _pos(e)[0] == (1, 0, rw.UP, 1) and e.grid[1, 0, 1] == 1 and e.grid[1, 1, 1] == 0
    both(e, [4, 0, 0, 0])
    assert _pos(e)[0][3] == 1  # still carrying
    # 9. a loaded agent cannot drive into a standing shelf; an unloaded one can
    e = _state(make(), [(1, 1, rw.DRIGHT, 1, 0), (7, 1, rw.DRIGHT, 0, 0)] + far)  # shelf 2 stands on (2,1), shelf 4 on (8,1)
    both(e, [1, 1, 0, 0])
    assert _pos(e)[:2] == [(1, 1, rw.DRIGHT, 1), (8, 1, rw.DRIGHT, 0)]
    # 10. delivery: a requested shelf carried onto a goal cell pays its carrier and is replaced in the queue
    g = make()
    g.reset(DrawStream(0, 0, 0))
    grid = g.get_state()["grid"].copy()
    grid[1, 1] = 0
    grid[9, 4] = 1  # shelf 1 is carried by agent 0 standing on (4, 9), one step above the goal (4, 10)
    e = _state(make(), [(4, 9, rw.DOWN, 1, 0), (5, 0, rw.UP, 0, 0)] + far, queue=(1, 2, 3, 4), grid=grid, steps=7)
    obs, rew, done, trunc, _ = both(e, [1, 0, 0, 0])
    assert rew == [1.0, 0.0, 0.0, 0.0] and not done
    k = DrawStream(0, 0, 0, rw.STREAM_REQUEST)
    k.idx = 8 * 7
    expect = [s for s in range(1, 33) if s not in (1, 2, 3, 4)][k.integers(0, 28)]
    assert [s.id for s in e.request_queue] == [expect, 2, 3, 4] and e._cur_inactive_steps == 0 and e._cur_steps == 8
    assert obs[0][:8].tolist() == [4.0, 10.0, 1.0, 0.0, 1.0, 0.0, 0.0, 1.0]
    # the centre cell of agent 0's window: itself (facing DOWN) on its shelf, which is no longer requested
    assert obs[0][8 + 4 * 7:8 + 5 * 7].tolist() == [1.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0]
    # cells below the grid read as empty: no agent -> direction one-hot(0)
    assert obs[0][8 + 7 * 7:8 + 8 * 7].tolist() == [0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    # 11. episode end at max_steps
    e = _state(make(max_steps=9), [(3, 0, rw.UP, 0, 0), (5, 0, rw.UP, 0, 0)] + far, steps=8)
    assert both(e, [0, 0, 0, 0])[2] is True


def test_known_answer_transitions_rewards_queue_and_carriers():
    far = [(9, 0, rw.UP, 0, 0), (9, 2, rw.UP, 0, 0)]
    g = make()
    g.reset(DrawStream(0, 0, 0))
    base = g.get_state()["grid"]

    def grid_with(moves):  # {(x, y): shelf id or 0}
        gr = base.copy()
        for (x, y), v in moves.items():
            gr[y, x] = v
        return gr

    # 12. unloading inside the rack puts the shelf down where the agent stands; the agent keeps standing under it
    e = _state(make(), [(1, 1, rw.UP, 1, 0), (5, 0, rw.UP, 0, 0)] + far)
    both(e, [4, 0, 0, 0])
    assert _pos(e)[0] == (1, 1, rw.UP, 0) and e.grid[1, 1, 1] == 1
    # 13. two carriers in a train: the follower's shelf enters the cell the leader's shelf leaves
    gr = grid_with({(1, 1): 0, (2, 1): 0, (3, 0): 1, (4, 0): 2})
    e = _state(make(), [(3, 0, rw.DRIGHT, 1, 0), (4, 0, rw.DRIGHT, 2, 0)] + far, grid=gr, queue=(5, 6, 7, 8))
    both(e, [1, 1, 0, 0])
    assert _pos(e)[:2] == [(4, 0, rw.DRIGHT, 1), (5, 0, rw.DRIGHT, 2)]
    assert e.grid[1, 0, 3] == 0 and e.grid[1, 0, 4] == 1 and e.grid[1, 0, 5] == 2
    # 14. ... but a carrier is stopped by a carrier that stays (its shelf is a standing obstacle for this step)
    e = _state(make(), [(3, 0, rw.DRIGHT, 1, 0), (4, 0, rw.DRIGHT, 2, 0)] + far, grid=gr, queue=(5, 6, 7, 8))
    both(e, [1, 0, 0, 0])
    assert _pos(e)[:2] == [(3, 0, rw.DRIGHT, 1), (4, 0, rw.DRIGHT, 2)]
    # 15. an unloaded agent drives under a standing shelf and can pick it up there next step
    e = _state(make(), [(0, 1, rw.DRIGHT, 0, 0), (5, 0, rw.UP, 0, 0)] + far)
    both(e, [1, 0, 0, 0])
    both(e, [4, 0, 0, 0])
    assert _pos(e)[0] == (1, 1, rw.DRIGHT, 1)
    # 16. a shelf that is NOT requested earns nothing on the goal and the queue is untouched
    gr = grid_with({(1, 1): 0, (4, 9): 1})
    e = _state(make(), [(4, 9, rw.DOWN, 1, 0), (5, 0, rw.UP, 0, 0)] + far, grid=gr, queue=(5, 6, 7, 8))
    _, rew, _, _, _ = both(e, [1, 0, 0, 0])
    assert rew == [0.0] * 4 and [s.id for s in e.request_queue] == [5, 6, 7, 8] and e._cur_inactive_steps == 1
    # 17. global rewards: every agent is paid for a delivery
    e = _state(make(reward_type=rw.REWARD_GLOBAL), [(4, 9, rw.DOWN, 1, 0), (5, 0, rw.UP, 0, 0)] + far, grid=gr, queue=(1, 6, 7, 8))
    assert both(e, [1, 0, 0, 0])[1] == [1.0] * 4
    # 18. two-stage rewards: half on delivery, the other half when the delivered shelf is put back into the rack
    e = _state(make(reward_type=rw.REWARD_TWO_STAGE), [(4, 9, rw.DOWN, 1, 0), (5, 0, rw.UP, 0, 0)] + far, grid=gr, queue=(1, 6, 7, 8))
    assert both(e, [1, 0, 0, 0])[1] == [0.5, 0.0, 0.0, 0.0] and e.agents[0].has_delivered
    assert both(e, [4, 0, 0, 0])[1] == [0.0] * 4 and e.agents[0].carrying_shelf is not None  # the goal row is a highway: no unloading
    gr2 = grid_with({(1, 1): 1})
    e = _state(make(reward_type=rw.REWARD_TWO_STAGE), [(1, 1, rw.UP, 1, 1), (5, 0, rw.UP, 0, 0)] + far, grid=gr2, queue=(5, 6, 7, 8))
    assert both(e, [4, 0, 0, 0])[1] == [0.5, 0.0, 0.0, 0.0] and not e.agents[0].has_delivered and e.agents[0].carrying_shelf is None
    # 19. both goal cells can deliver in the same step; each replacement excludes the shelves queued at that moment
    gr = grid_with({(1, 1): 0, (2, 1): 0, (4, 9): 1, (5, 9): 2})
    e = _state(make(), [(4, 9, rw.DOWN, 1, 0), (5, 9, rw.DOWN, 2, 0)] + far, grid=gr, queue=(1, 2, 3, 4), steps=3)
    _, rew, _, _, _ = both(e, [1, 1, 0, 0])
    k = DrawStream(0, 0, 0, rw.STREAM_REQUEST)
    k.idx = 8 * 3
    first = [s for s in range(1, 33) if s not in (1, 2, 3, 4)][k.integers(0, 28)]
    second = [s for s in range(1, 33) if s not in (first, 2, 3, 4)][k.integers(0, 28)]
    assert rew == [1.0, 1.0, 0.0, 0.0] and [s.id for s in e.request_queue] == [first, second, 3, 4]
    # 20. max_inactivity_steps ends the episode after that many delivery-free steps; a delivery resets the count
    e = _state(make(max_inactivity_steps=3), [(3, 0, rw.UP, 0, 0), (5, 0, rw.UP, 0, 0)] + far)
    assert [both(e, [0, 0, 0, 0])[2] for _ in range(3)] == [False, False, True]
    # 21. the time limit wrapper truncates, the env's own max_steps terminates (marlbase stores done | truncated)
    m = MarlbaseEnv(TINY4, 5, max_steps=7)
    m.reset(DrawStream(1, 0, 0))
    flags = [m.step([0, 0, 0, 0])[2:4] for _ in range(5)]
    assert flags[3] == (False, False) and flags[4] == (False, True)
    # 22. observation of a neighbour: agent 1 (facing LEFT, carrying requested shelf 2) sits right of agent 0
    gr = grid_with({(2, 1): 0, (4, 0): 2})
    e = _state(make(), [(3, 0, rw.UP, 0, 0), (4, 0, rw.DLEFT, 2, 0)] + far, grid=gr, queue=(2, 6, 7, 8))
    obs = both(e, [0, 0, 0, 0])[0]
    assert obs[0][8 + 5 * 7:8 + 6 * 7].tolist() == [1.0, 0.0, 0.0, 1.0, 0.0, 1.0, 1.0]  # window cell 5 = (x+1, y)
    assert obs[0][8:8 + 3 * 7].tolist() == [0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0] * 3       # the row above the grid is padding
    assert obs[1][:8].tolist() == [4.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 1.0]


@pytest.mark.parametrize("name,over,coop", [(TINY4, {}, False), ("rware:rware-tiny-2ag-easy-v2", {}, True),
                                            ("rware:rware-small-4ag-hard-v2", {"reward_type": rw.REWARD_TWO_STAGE}, False),
                                            ("rware:rware-tiny-8ag-v2", {"reward_type": rw.REWARD_GLOBAL, "max_inactivity_steps": 60}, False)])
def test_core_matches_oracle(name, over, coop):
    lib = host_shim()
    N, T, seed = 12, 120, 4321
    kw = dict(rw.parse_env_name(name), **over)
    P = kw["n_agents"]
    envs = [MarlbaseEnv(name, T, cooperative=coop, **over) for _ in range(N)]
    R, C = envs[0].env.grid_size
    hc = RwCfg(n_envs=N, n_agents=P, rows=R, cols=C, column_height=kw["column_height"], n_shelves=0, queue_size=kw["request_queue_size"],
               max_steps=kw["max_steps"] or 0, max_inactivity_steps=kw["max_inactivity_steps"] or 0, time_limit=T,
               reward_type=kw["reward_type"], cooperative=int(coop), seed=seed)
    hc.n_shelves = lib.host_rw_count_shelves(ctypes.byref(hc))
    S = lib.host_rw_stride(P, R, C)
    rng = np.random.default_rng(3)
    delivered = 0
    for episode in range(2):
        state = np.zeros((N, S), np.uint8)
        obs = np.zeros((P, N, 71), np.float32)
        epi = np.full(N, episode, np.uint32)
        assert lib.host_rw_reset(ctypes.byref(hc), ptr(state), ptr(epi), ptr(obs)) == 0
        for n, e in enumerate(envs):
            o, _ = e.reset(DrawStream(seed, n, episode))
            assert hc.n_shelves == len(e.env.shelfs)
            np.testing.assert_array_equal(pack_state(e.env, P), state[n])
            for p in range(P):
                np.testing.assert_array_equal(o[p], obs[p, n])
        if episode == 1:  # seed deliveries: put every requested shelf on its carrier one step above a goal
            for n, e in enumerate(envs):
                w = e.env
                a = w.agents[0]
                sh = w.request_queue[n % len(w.request_queue)]
                if any((b.x, b.y) == (C // 2 - 1, R - 2) for b in w.agents[1:]):
                    continue
                a.x, a.y, a.dir, a.carrying_shelf = C // 2 - 1, R - 2, rw.DOWN, sh
                sh.x, sh.y = a.x, a.y
                w._recalc_grid()
                state[n] = pack_state(w, P)
        # the collectors carry the request queue as a bit set across steps and rebuild it only when rw_step reports a delivery:
        # the carried set against a fresh build after EVERY step, on a copy of the same states
        rq = np.zeros((N, 4), np.uint64)
        assert lib.host_rw_step_carried_rq(ctypes.byref(hc), ptr(state), ptr(state), ptr(epi), None, ptr(rq), 1) == 0
        for t in range(T):
            acts = rng.choice(5, size=(P, N), p=[0.1, 0.5, 0.1, 0.1, 0.2]).astype(np.int32)
            if episode == 1 and t == 0:
                acts[0, :] = 1
            rew = np.zeros((P, N), np.float32)
            done = np.zeros(N, np.uint8)
            trunc = np.zeros(N, np.uint8)
            before = state.copy()
            assert lib.host_rw_step(ctypes.byref(hc), ptr(state), ptr(epi), ptr(acts), ptr(obs), ptr(rew), ptr(done), ptr(trunc)) == 0
            assert lib.host_rw_step_carried_rq(ctypes.byref(hc), ptr(before), ptr(state), ptr(epi), ptr(acts), ptr(rq), 0) == 0, f"stale request bits at step {t}"
            for n, e in enumerate(envs):
                o, r, d, tr, _ = e.step([int(a) for a in acts[:, n]])
                np.testing.assert_array_equal(pack_state(e.env, P), state[n], err_msg=f"env {n} step {t}")
                for p in range(P):
                    np.testing.assert_array_equal(o[p], obs[p, n], err_msg=f"obs env {n} step {t} agent {p}")
                np.testing.assert_array_equal(np.array(r, np.float32), rew[:, n])
                assert bool(done[n]) == d and bool(trunc[n]) == tr
                delivered += sum(r) > 0
            assert lib.host_rw_obs_word_check(ctypes.byref(hc), ptr(state)) == 0  # the collectors' packed-window route
            if done.any() or trunc.any():
                break
    assert delivered > 0
