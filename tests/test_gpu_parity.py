"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle and the golden traces.

Bit-exact: positions, budgets, masks, flags, winner, visit counts, timesteps, sampled actions.
Rewards: float64, bit-exact against the oracle when both use the same exp/coverage tables
(the engine keeps the reference's operation order); 1e-12 against the reference goldens.
Belief: float32 on device vs float64 oracle, absolute tolerance 1e-5 (BASELINE.json north_star).
"""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from tests.helpers import GOLDEN, engine_actions, load_trace, trace_index  # noqa: E402

pytestmark = pytest.mark.gpu

BELIEF_TOL = 1e-5


@pytest.fixture(scope="module")
def sy():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import student_mechanism_design_amd as sy_mod
    sy_mod._lib.load()  # fail loudly if the HIP library is missing
    return sy_mod


@pytest.fixture(scope="module")
def ol():
    from oracle import oracle_lib
    oracle_lib.load()
    return oracle_lib


def _np(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------------------------
# golden traces recorded from the unmodified reference
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [e["file"] for e in trace_index()])
def test_golden_trace_on_device(sy, name):
    tr = load_trace(name)
    N, P, A = int(tr["N"]), int(tr["P"]), int(tr["P"]) + 1
    board = sy.make_board(N, tr["edge_links"], tr["edge_w"])
    B = 5  # the same episode on several waves (incl. a partial launch block)
    env = sy.BatchedScotlandYardEnv(B, [board], P, int(tr["money0"]), tr["weights"], auto_reset=False,
                                    waves_per_block=4)
    env.reset_to(np.tile(tr["starts"], (B, 1)))
    np.testing.assert_array_equal(_np(env.action_mask), np.tile(tr["mask0"], (B, 1, 1)))
    T = tr["actions"].shape[0]
    for s in range(T):
        act = np.tile(engine_actions(tr["actions"][s]), (B, 1))
        env.step(torch.as_tensor(act, dtype=torch.int32, device=env.device))
        for b in (0, B - 1):
            np.testing.assert_array_equal(_np(env.pos)[b], tr["pos"][s], err_msg=f"pos step {s}")
            np.testing.assert_array_equal(_np(env.budget)[b], tr["money"][s], err_msg=f"money step {s}")
            assert bool(_np(env.terminated)[b]) == bool(tr["terminated"][s]), f"terminated step {s}"
            assert bool(_np(env.truncated)[b]) == bool(tr["truncated"][s]), f"truncated step {s}"
            assert int(_np(env.winner)[b]) == int(tr["winner"][s]), f"winner step {s}"
            assert int(_np(env.t)[b]) == int(tr["t_after"][s])
            np.testing.assert_array_equal(_np(env.visits)[b].astype(np.int32), tr["visits"][s], err_msg=f"visits {s}")
            np.testing.assert_array_equal(_np(env.action_mask)[b], tr["masks"][s], err_msg=f"masks step {s}")
            np.testing.assert_allclose(_np(env.reward)[b], tr["reward"][s], rtol=1e-12, atol=1e-12,
                                       err_msg=f"reward step {s}")
    env.close()


# ----------------------------------------------------------------------------------------------
# engine vs oracle on seeded random inputs
# ----------------------------------------------------------------------------------------------
def _make_pair(sy, ol, B, N, E, P, money, G, seed, **kw):
    boards = sy.sample_board_pool(G, N, E, seed=seed)
    rng = np.random.default_rng(seed)
    weights = rng.uniform(0.05, 0.95, 11)
    env = sy.BatchedScotlandYardEnv(B, boards, P, money, weights, seed=seed, **kw)
    graphs = [ol.OracleGraph(N, b.edge_links, b.edges.astype(np.int32)) for b in boards]
    tables = sy.reward_tables()
    okw = {k: v for k, v in kw.items() if k in ("reveal_interval", "police_evidence", "belief_init_onehot",
                                                "auto_reset", "env_id_offset", "max_t")}
    if "max_timestep" in kw:
        okw["max_t"] = kw["max_timestep"]
    orc = ol.OracleBatch(graphs, env.env_graph_host, B, P, money, node_stride=env.NS, weights=weights,
                         tables=tables, **okw)
    orc.reset(seed=seed)
    return env, orc, boards


def _compare_state(env, orc, what=""):
    N = env.N
    np.testing.assert_array_equal(_np(env.pos), orc.pos, err_msg=f"pos {what}")
    np.testing.assert_array_equal(_np(env.budget), orc.money, err_msg=f"budget {what}")
    np.testing.assert_array_equal(_np(env.t), orc.t, err_msg=f"t {what}")
    np.testing.assert_array_equal(_np(env.step_count).astype(np.uint32), orc.step_count, err_msg=f"step_count {what}")
    np.testing.assert_array_equal(_np(env.visits).astype(np.int32), orc.visits[:, :N], err_msg=f"visits {what}")
    np.testing.assert_array_equal(_np(env._mask), orc.mask, err_msg=f"mask {what}")
    np.testing.assert_array_equal(_np(env._terminated), orc.terminated, err_msg=f"terminated {what}")
    np.testing.assert_array_equal(_np(env._truncated), orc.truncated, err_msg=f"truncated {what}")
    np.testing.assert_array_equal(_np(env.winner), orc.winner, err_msg=f"winner {what}")
    np.testing.assert_array_equal(_np(env.reward), orc.reward, err_msg=f"reward {what}")  # bit-exact float64
    if env.belief is not None:
        np.testing.assert_allclose(_np(env.belief), orc.belief[:, :N], rtol=0, atol=BELIEF_TOL, err_msg=f"belief {what}")


def _random_actions(rng, env_pos, mask, N):
    """Mostly legal moves, some no-ops / illegal / occupied targets."""
    B, A = env_pos.shape
    act = np.full((B, A), -1, dtype=np.int32)
    u = rng.random((B, A))
    for b in range(B):
        for a in range(A):
            legal = np.nonzero(mask[b, a])[0]
            if u[b, a] < 0.7 and legal.size:
                act[b, a] = rng.choice(legal)
            elif u[b, a] < 0.8:
                act[b, a] = -1
            elif u[b, a] < 0.9:
                act[b, a] = rng.integers(0, N + 3)
            else:
                act[b, a] = env_pos[b, rng.integers(0, A)]
    return act


@pytest.mark.parametrize("cfg", [
    dict(B=64, N=200, E=400, P=4, money=20, G=2, seed=1, reveal_interval=5),
    dict(B=37, N=15, E=20, P=2, money=10, G=1, seed=2, police_evidence=True),
    dict(B=48, N=64, E=110, P=6, money=8, G=3, seed=3, reveal_interval=3, belief_init_onehot=True, max_timestep=20),
    dict(B=20, N=8, E=10, P=5, money=5, G=1, seed=4, auto_reset=False),
    dict(B=16, N=300, E=560, P=7, money=30, G=1, seed=5, reveal_interval=7, police_evidence=True),
])
def test_step_matches_oracle(sy, ol, cfg):
    cfg = dict(cfg)
    B, N, E, P, money, G, seed = (cfg.pop(k) for k in ("B", "N", "E", "P", "money", "G", "seed"))
    env, orc, _ = _make_pair(sy, ol, B, N, E, P, money, G, seed, **cfg)
    _compare_state(env, orc, "after reset")
    rng = np.random.default_rng(100 + seed)
    for s in range(120):
        act = _random_actions(rng, orc.pos, orc.mask[:, :, :N], N)
        env.step(torch.as_tensor(act, device=env.device))
        orc.step(act)
        _compare_state(env, orc, f"step {s}")
    env.close()


@pytest.mark.parametrize("cfg", [
    dict(B=128, N=200, E=400, P=4, money=20, G=4, seed=11, reveal_interval=5, T=96),
    dict(B=33, N=15, E=20, P=2, money=10, G=1, seed=12, T=300),
    dict(B=40, N=100, E=190, P=6, money=6, G=2, seed=13, police_evidence=True, reveal_interval=4, T=80),
    dict(B=24, N=40, E=70, P=3, money=50, G=1, seed=14, max_timestep=30, T=200),
    # BASELINE configs[3] shape on one rank's shard: 6 police, global env ids of rank 3 of 8
    dict(B=96, N=200, E=400, P=6, money=20, G=2, seed=15, reveal_interval=5, env_id_offset=3 * 4096, T=48),
    # BASELINE configs[4] stand-in: 199 nodes (no London topology offline), 5 police, reveal every 5
    dict(B=72, N=199, E=380, P=5, money=24, G=3, seed=16, reveal_interval=5, T=64),
    # odd block sizes / partial blocks: 7 envs per block, 3 belief waves + a half-used one
    dict(B=45, N=70, E=120, P=4, money=9, G=1, seed=17, reveal_interval=2, police_evidence=True, waves_per_block=7, T=40),
    dict(B=9, N=520, E=1000, P=2, money=6, G=1, seed=18, reveal_interval=6, waves_per_block=2, T=30),
    # a board too big for 16 episodes per block: the engine picks the block size that fits the LDS
    dict(B=37, N=520, E=1000, P=3, money=8, G=1, seed=19, reveal_interval=4, T=30),
])
def test_fused_rollout_matches_oracle(sy, ol, cfg):
    cfg = dict(cfg)
    B, N, E, P, money, G, seed, T = (cfg.pop(k) for k in ("B", "N", "E", "P", "money", "G", "seed", "T"))
    env, orc, _ = _make_pair(sy, ol, B, N, E, P, money, G, seed, **cfg)
    rec = env.rollout(T)
    ref = orc.rollout(T)
    for k in ("pos", "t", "action", "terminated", "truncated", "winner", "mask", "reward"):
        np.testing.assert_array_equal(_np(rec[k]), ref[k], err_msg=k)
    np.testing.assert_array_equal(_np(rec["budget"]), ref["money"], err_msg="budget")
    np.testing.assert_allclose(_np(rec["belief"]), ref["belief"], rtol=0, atol=BELIEF_TOL)
    _compare_state(env, orc, "after rollout")
    done = ref["terminated"] | ref["truncated"]
    assert done.any(), "the case must contain finished episodes (auto-reset path)"
    # a second rollout continues from the live state
    rec2 = env.rollout(7)
    ref2 = orc.rollout(7)
    np.testing.assert_array_equal(_np(rec2["action"]), ref2["action"])
    np.testing.assert_array_equal(_np(rec2["pos"]), ref2["pos"])
    env.close()


def _hub_board(sy, n, hub_degree, rng):
    """Ring of n nodes plus a hub (node 0) wired to `hub_degree` nodes in all: the pool's widest ELL row is exact."""
    links = [(i, (i + 1) % n) for i in range(n)]
    extra = rng.choice(np.arange(2, n - 1), size=hub_degree - 2, replace=False)
    links += [(0, int(v)) for v in extra]
    # a second, smaller hub so that wide rows are not a single-node affair
    links += [(n // 2, int(v)) for v in rng.choice(np.setdiff1d(np.arange(2, n - 1), [n // 2 - 1, n // 2, n // 2 + 1]),
                                                   size=max(hub_degree - 5, 1), replace=False)]
    links = sorted({(min(a, b), max(a, b)) for a, b in links if a != b})
    w = rng.integers(1, 5, size=len(links))
    return sy.make_board(n, np.array(links, dtype=np.int32), w)


@pytest.mark.parametrize("police,hub_degree", [(6, 9), (6, 10), (6, 12), (6, 16), (5, 10), (5, 12), (7, 9), (7, 14), (4, 13),
                                               (4, 12), (4, 7), (4, 6), (2, 16), (2, 11)])
def test_fused_rollout_scan_widths_and_passes(sy, ol, police, hub_degree):
    """Every lane mapping of the neighbour scan: exact widths 9 / 10 that save a pass for 6-7 agents, the
    two-pass slot scan (7 agents at width 12, 6 at 12, 8 at 16), the generic multi-pass fallback, and the half-wave
    scan of up to 5 agents (P = 4: 6 columns per agent and lane pair, rows of 6 / 7 / 12 neighbours, 13 falls back to
    the paired scan; P = 2: 10 columns, rows of 11 and 16)."""
    rng = np.random.default_rng(1000 * police + hub_degree)
    N, B, T = 48, 40, 60
    boards = [_hub_board(sy, N, hub_degree, rng) for _ in range(2)]
    weights = rng.uniform(0.05, 0.95, 11)
    env = sy.BatchedScotlandYardEnv(B, boards, police, 9, weights, seed=77, reveal_interval=3)
    assert env.max_degree == hub_degree
    graphs = [ol.OracleGraph(N, b.edge_links, b.edges.astype(np.int32)) for b in boards]
    orc = ol.OracleBatch(graphs, env.env_graph_host, B, police, 9, node_stride=env.NS, weights=weights,
                         tables=sy.reward_tables(), reveal_interval=3)
    orc.reset(seed=77)
    rec = env.rollout(T)
    ref = orc.rollout(T)
    for k in ("pos", "t", "action", "terminated", "truncated", "winner", "mask", "reward"):
        np.testing.assert_array_equal(_np(rec[k]), ref[k], err_msg=k)
    np.testing.assert_allclose(_np(rec["belief"]), ref["belief"], rtol=0, atol=BELIEF_TOL)
    _compare_state(env, orc, "after scan-width rollout")
    env.close()


@pytest.mark.parametrize("police,hub_degree", [(5, 10), (5, 11), (5, 15), (5, 16), (6, 8), (6, 12), (6, 13), (6, 16), (4, 12), (4, 16)])
def test_half_wave_scan_columns_on_mid_size_boards(sy, ol, police, hub_degree):
    """Boards of 129..256 nodes: 6 and 7 agents take the half-wave scan with 2..4 columns per lane (5 / 4 columns per
    agent and pass); every column count and the fallback to the paired scan (6 agents at rows of 16) against the oracle."""
    rng = np.random.default_rng(77 * police + hub_degree)
    N, B, T = 136, 48, 50
    boards = [_hub_board(sy, N, hub_degree, rng) for _ in range(2)]
    weights = rng.uniform(0.05, 0.95, 11)
    env = sy.BatchedScotlandYardEnv(B, boards, police, 9, weights, seed=31, reveal_interval=4)
    assert env.max_degree == hub_degree
    graphs = [ol.OracleGraph(N, b.edge_links, b.edges.astype(np.int32)) for b in boards]
    orc = ol.OracleBatch(graphs, env.env_graph_host, B, police, 9, node_stride=env.NS, weights=weights,
                         tables=sy.reward_tables(), reveal_interval=4)
    orc.reset(seed=31)
    rec = env.rollout(T)
    ref = orc.rollout(T)
    for k in ("pos", "t", "action", "terminated", "truncated", "winner", "mask", "reward"):
        np.testing.assert_array_equal(_np(rec[k]), ref[k], err_msg=k)
    np.testing.assert_allclose(_np(rec["belief"]), ref["belief"], rtol=0, atol=BELIEF_TOL)
    _compare_state(env, orc, "after half-wave scan rollout")
    env.close()


@pytest.mark.parametrize("N,P,B", [(200, 4, 33), (200, 4, 1), (200, 6, 17), (199, 5, 31), (136, 2, 5), (64, 4, 19), (200, 4, 4097)])
def test_odd_batches_match_oracle(sy, ol, N, P, B):
    """Batches that leave a pair of episodes half empty (the move / helper waves carry two episodes): the last wave of
    the last block simulates a duplicate whose stores are suppressed — every scan instance, one block and many."""
    boards = sy.sample_board_pool(3, N, 2 * N, seed=100 + B)
    weights = np.random.default_rng(100 + B).uniform(0.05, 0.95, 11)
    env = sy.BatchedScotlandYardEnv(B, boards, P, 12, weights, seed=100 + B, reveal_interval=4)
    graphs = [ol.OracleGraph(N, b.edge_links, b.edges.astype(np.int32)) for b in boards]
    orc = ol.OracleBatch(graphs, env.env_graph_host, B, P, 12, node_stride=env.NS, weights=weights,
                         tables=sy.reward_tables(), reveal_interval=4)
    orc.reset(seed=100 + B)
    rec = env.rollout(70)
    ref = orc.rollout(70)
    for k in ("pos", "t", "action", "terminated", "truncated", "winner", "mask", "reward"):
        np.testing.assert_array_equal(_np(rec[k]), ref[k], err_msg=k)
    np.testing.assert_array_equal(_np(rec["budget"]), ref["money"], err_msg="budget")
    np.testing.assert_allclose(_np(rec["belief"]), ref["belief"], rtol=0, atol=BELIEF_TOL)
    _compare_state(env, orc, "after an odd-batch rollout")
    assert env.status() == 0
    env.close()


def test_rollout_equals_stepping_its_own_actions(sy):
    """Fused sampling path == caller-action path: replaying the recorded actions through step()."""
    boards = sy.sample_board_pool(2, 60, 100, seed=21)
    w = np.linspace(0.1, 0.9, 11)
    a = sy.BatchedScotlandYardEnv(64, boards, 4, 12, w, seed=5, reveal_interval=5)
    b = sy.BatchedScotlandYardEnv(64, boards, 4, 12, w, seed=5, reveal_interval=5)
    rec = a.rollout(50)
    for s in range(50):
        np.testing.assert_array_equal(_np(b.pos), _np(rec["pos"][s]))
        np.testing.assert_array_equal(_np(b._mask), _np(rec["mask"][s]))
        # (the fused rollout renormalises the belief every few steps, the step kernel on every step: equal up to
        # float32 rounding, both within 1e-5 of the oracle)
        np.testing.assert_allclose(_np(b._belief), _np(rec["belief"][s]), rtol=0, atol=2e-6)
        b.step(rec["action"][s].contiguous())
        np.testing.assert_array_equal(_np(b.reward), _np(rec["reward"][s]))
        np.testing.assert_array_equal(_np(b._terminated), _np(rec["terminated"][s]))
    np.testing.assert_array_equal(_np(a.pos), _np(b.pos))
    a.close()
    b.close()


# ----------------------------------------------------------------------------------------------
# BASELINE.json full size (N=200, P=4, B=4096): size-independent properties
# ----------------------------------------------------------------------------------------------
def test_full_size_properties(sy):
    N, E, P, B, T = 200, 400, 4, 4096, 64
    boards = sy.sample_board_pool(8, N, E, seed=7)
    env = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=3, reveal_interval=5)
    rec = env.rollout(T)
    pos, bud, act = _np(rec["pos"]), _np(rec["budget"]), _np(rec["action"])
    mask, bel = _np(rec["mask"])[..., :N].astype(bool), _np(rec["belief"])[..., :N]
    done = (_np(rec["terminated"]) | _np(rec["truncated"])).astype(bool)
    # agents never share a node before a step; MrX's budget is constant
    srt = np.sort(pos, axis=-1)
    assert (np.diff(srt, axis=-1) > 0).all()
    assert (bud[..., 0] == 1000).all() and (bud[..., 1:] >= 0).all() and (bud[..., 1:] <= 20).all()
    # every sampled action is legal under the recorded mask (or -1 with an empty mask)
    tt, bb, aa = np.nonzero(act >= 0)
    assert mask[tt, bb, aa, act[tt, bb, aa]].all()
    assert (mask.sum(-1)[act < 0] == 0).all()
    # masks = affordable ELL neighbours of the recorded position
    ell = env.pool.ell
    g = env.env_graph_host
    for t_ in (0, T // 2, T - 1):
        for b_ in (0, 1, 2047, 4095):
            for a_ in range(P + 1):
                row = ell[g[b_], pos[t_, b_, a_]]
                nb, w = row & 0xFFFF, row >> 16
                legal = np.zeros(N, bool)
                legal[nb[(nb < N) & (w <= bud[t_, b_, a_])]] = True
                np.testing.assert_array_equal(mask[t_, b_, a_], legal)
    # beliefs are distributions
    np.testing.assert_allclose(bel.sum(-1), 1.0, atol=1e-4)
    assert (bel >= 0).all()
    # budgets never grow inside an episode; timestep advances by one or restarts at 0 after done
    t_arr = _np(rec["t"])
    cont = ~done[:-1]
    assert (bud[1:][cont] <= bud[:-1][cont]).all()
    assert (t_arr[1:][cont] == t_arr[:-1][cont] + 1).all()
    assert (t_arr[1:][done[:-1]] == 0).all()
    # fusing is idempotent: T steps in one launch == T launches of one step
    env2 = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=3, reveal_interval=5)
    for _ in range(8):
        env2.rollout(1, record=False)
    env3 = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=3, reveal_interval=5)
    env3.rollout(8, record=False)
    for name in ("pos", "budget", "t", "_mask", "_visits", "reward"):
        assert torch.equal(getattr(env2, name), getattr(env3, name)), name
    # the belief is renormalised when a launch ends and every 8th step inside one: equal up to float32 rounding
    assert torch.allclose(env2._belief, env3._belief, rtol=0, atol=2e-6)
    # different seeds / env_id_offset give different streams
    env4 = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=3, env_id_offset=B)
    env5 = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=3)
    env6 = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=4)
    assert not torch.equal(env4.pos, env5.pos) and not torch.equal(env6.pos, env5.pos)


# ----------------------------------------------------------------------------------------------
# Every benchmarked shape against the oracle AT ITS OWN SIZE: the bench.py workload (BASELINE configs[1]:
# same boards, weights, seed, block size, fused length T = 256 — the Philox step counter passes 255 and
# hundreds of episodes restart inside the launch) and the per-GPU shards of configs[3] / configs[4].
# The oracle (OpenMP over envs) does each in about a second.
# ----------------------------------------------------------------------------------------------
FULL_SIZE_CASES = {
    "configs1_bench": dict(B=4096, N=200, E=400, P=4, money=20, G=8, env_seed=1234, env_id_offset=0, T=256),
    "configs3_shard_rank3": dict(B=4096, N=200, E=400, P=6, money=20, G=8, env_seed=1234, env_id_offset=3 * 4096, T=256),
    "configs4_shard_rank5": dict(B=8192, N=199, E=400, P=5, money=20, G=8, env_seed=1234, env_id_offset=5 * 8192, T=256),
}


def compare_rollout_with_oracle(rec, ref, T, chunk=32):
    """Record of T fused steps vs the oracle's, compared in time slices (bounded host memory)."""
    for s0 in range(0, T, chunk):
        s1 = min(T, s0 + chunk)
        for k in ("pos", "t", "action", "terminated", "truncated", "winner", "mask", "reward"):
            np.testing.assert_array_equal(_np(rec[k][s0:s1]), ref[k][s0:s1], err_msg=f"{k} steps {s0}..{s1}")
        np.testing.assert_array_equal(_np(rec["budget"][s0:s1]), ref["money"][s0:s1], err_msg=f"budget steps {s0}..{s1}")
        if rec.get("belief") is not None:
            d = np.abs(_np(rec["belief"][s0:s1]).astype(np.float64) - ref["belief"][s0:s1]).max()
            assert d <= BELIEF_TOL, f"belief steps {s0}..{s1}: max abs diff {d}"


@pytest.mark.parametrize("name", list(FULL_SIZE_CASES))
def test_benchmarked_shape_matches_oracle_at_full_size(sy, ol, name):
    c = FULL_SIZE_CASES[name]
    B, N, P, T = c["B"], c["N"], c["P"], c["T"]
    boards = sy.sample_board_pool(c["G"], N, c["E"], seed=0)            # bench.py's boards
    weights = np.full(11, 0.5)                                          # bench.py's weights
    env = sy.BatchedScotlandYardEnv(B, boards, P, c["money"], weights, seed=c["env_seed"], reveal_interval=5,
                                    env_id_offset=c["env_id_offset"])
    assert env.waves_per_block == 16                                    # the default block size the bench runs
    graphs = [ol.OracleGraph(N, b.edge_links, b.edges.astype(np.int32)) for b in boards]
    orc = ol.OracleBatch(graphs, env.env_graph_host, B, P, c["money"], node_stride=env.NS, weights=weights,
                         tables=sy.reward_tables(), reveal_interval=5, env_id_offset=c["env_id_offset"],
                         threads=max(1, len(os.sched_getaffinity(0))))
    orc.reset(seed=c["env_seed"])
    _compare_state(env, orc, "after reset")
    out = env.alloc_rollout(T)
    for launch in range(2):            # the second launch continues from the live state, like the bench's timed loop
        rec = env.rollout(T, out=out)
        ref = orc.rollout(T)
        compare_rollout_with_oracle(rec, ref, T)
        _compare_state(env, orc, f"after launch {launch}")
        done = ref["terminated"] | ref["truncated"]
        assert done.sum() > B // 4, "restarts inside a launch must be exercised at scale"
        del ref
    env.close()


# ----------------------------------------------------------------------------------------------
# action-mask known answers and belief filter through their own entry points
# ----------------------------------------------------------------------------------------------
def test_action_mask_known_answers_on_device(sy):
    with open(os.path.join(GOLDEN, "action_mask_kats.json")) as f:
        cases = json.load(f)
    for c in cases:
        tolls = c["tolls"]
        if tolls is not None and not np.isscalar(tolls):
            tolls = np.array(tolls)
        w = None if c["edge_weights"] is None else np.array(c["edge_weights"], dtype=float)
        r = sy.compute_action_mask(np.array(c["adjacency"], dtype=float), c["current_node"], c["budget"],
                                   tolls=tolls, edge_weights=w)
        np.testing.assert_array_equal(r.mask, np.array(c["mask"], dtype=bool), err_msg=c["tag"])
        n = len(c["mask"])
        assert r.index_to_node == {i: i for i in range(n)} and r.node_to_index == r.index_to_node
        assert r.valid_actions == [i for i in range(n) if c["mask"][i]]
        assert r.num_valid_actions == sum(c["mask"])


def test_belief_tracker_on_device(sy, ol):
    with open(os.path.join(GOLDEN, "belief_reference.json")) as f:
        ref = json.load(f)
    # the reference's own test scenario (test/test_belief_update.py)
    adj = np.array(ref["ref_test"]["adjacency"])
    tr = sy.DeviceBeliefTracker(3, adj)
    b = _np(tr.update(adj, observation_hint=[1]))[0]
    np.testing.assert_allclose(b, [1 / 42, 40 / 42, 1 / 42], atol=BELIEF_TOL)
    assert np.isclose(b.sum(), 1.0)
    b = _np(tr.update(adj, reveal=2))[0]
    assert b.argmax() == 2 and np.isclose(b.sum(), 1.0)
    # scripted scenarios vs the float64 oracle filter (itself pinned to the particle tracker)
    for case in ref["monte_carlo"]:
        adj = np.array(case["adjacency"])
        n = adj.shape[0]
        links = [(i, j) for i in range(n) for j in range(i + 1, n) if adj[i, j]]
        g = ol.OracleGraph(n, np.array(links, np.int32).reshape(-1, 2), np.ones(len(links), np.int32))
        tr = sy.DeviceBeliefTracker(n, adj, num_beliefs=3)
        ob = np.full(n, 1.0 / n)
        for st in case["steps"]:
            if st["kind"] == "hint":
                ob = ol.belief_update(g, ob, hint=st["hint"])
                db = tr.update(observation_hint=st["hint"])
            elif st["kind"] == "reveal":
                ob = ol.belief_update(g, ob, reveal=st["reveal"])
                db = tr.update(reveal=st["reveal"])
            else:
                ob = ol.belief_update(g, ob)
                db = tr.update()
            for q in range(3):
                np.testing.assert_allclose(_np(db)[q], ob, atol=BELIEF_TOL)


def test_errors_are_reported_not_thrown_across_the_abi(sy):
    import ctypes as C
    lib = sy._lib.load()
    cfg = sy._lib.EnvConfig(0, 10, 2, 10, 250, 1, 16, 0, 0, 0, 1, 0, 0)
    h = C.c_void_p()
    assert lib.sy_env_create(C.byref(cfg), C.byref(h)) == -1
    assert b"num_envs" in lib.sy_last_error()
    cfg = sy._lib.EnvConfig(4, 10, 2, 10, 250, 1, 16, 0, 0, 0, 1, 0, 0)
    assert lib.sy_env_create(C.byref(cfg), C.byref(h)) == 0
    assert lib.sy_env_step(h, None, None) == -2  # graph pool / state not bound yet
    assert lib.sy_env_destroy(h) == 0
    with pytest.raises(ValueError):
        sy.BatchedScotlandYardEnv(4, [sy.sample_board(10, 14, rng=np.random.default_rng(0))], 2, 10, np.zeros(5))


def test_device_apsp_matches_host_and_reference_distances(sy, ol):
    """sy_build_apsp (Bellman-Ford per source on the GPU) == the ORACLE's shortest paths (`OracleGraph.dist`,
    pinned to the reference's Dijkstra by the golden float64 rewards) == host Floyd-Warshall — bit-exact integers."""
    boards = sy.sample_board_pool(5, 200, 400, seed=9)
    pool = sy.pack_pool(boards)
    dev = sy.device_all_pairs_shortest_paths(pool.ell, 200)
    np.testing.assert_array_equal(_np(dev).view(np.uint16), pool.apsp)
    for g, b in enumerate(boards):       # directly against the checker, not through product code
        og = ol.OracleGraph(200, b.edge_links, b.edges.astype(np.int32))
        np.testing.assert_array_equal(_np(dev)[g].view(np.uint16).astype(np.int32), og.dist)
    for n, e, seed in ((15, 20, 1), (64, 63, 2), (130, 250, 3), (520, 1000, 4)):
        bs = sy.sample_board_pool(2, n, e, seed=seed)
        pk = sy.pack_pool(bs)
        np.testing.assert_array_equal(_np(sy.device_all_pairs_shortest_paths(pk.ell, n)).view(np.uint16), pk.apsp)
    # golden board (N=200 reference trace) and a disconnected board (0xFFFF = unreachable)
    tr = load_trace("trace_s10_n200_p4_m20_random_valid_ep0.npz")
    b = sy.make_board(200, tr["edge_links"], tr["edge_w"])
    d200 = _np(sy.device_all_pairs_shortest_paths(sy.pack_ell(b)[None], 200))[0].view(np.uint16)
    np.testing.assert_array_equal(d200, sy.all_pairs_shortest_paths(b))
    np.testing.assert_array_equal(d200.astype(np.int32), ol.OracleGraph(200, tr["edge_links"], tr["edge_w"].astype(np.int32)).dist)
    two = sy.make_board(6, [[0, 1], [1, 2], [3, 4]], [2, 3, 1])
    d = _np(sy.device_all_pairs_shortest_paths(sy.pack_ell(two)[None], 6))[0].view(np.uint16)
    assert d[0, 2] == 5 and d[3, 4] == 1 and d[0, 3] == 0xFFFF and d[5, 0] == 0xFFFF and d[5, 5] == 0


def test_partial_reset_and_reseed_match_oracle(sy, ol):
    """sy_env_reset with an env selection (CustomEnvironment.reset of some envs) and with a new seed."""
    env, orc, _ = _make_pair(sy, ol, 40, 30, 50, 3, 9, 2, 31, reveal_interval=4, auto_reset=False)
    rng = np.random.default_rng(5)
    for s in range(12):
        act = _random_actions(rng, orc.pos, orc.mask[:, :, :30], 30)
        env.step(torch.as_tensor(act, device=env.device))
        orc.step(act)
    sel = (orc.terminated | orc.truncated).astype(bool) | (np.arange(40) % 3 == 0)
    env.reset(env_mask=torch.as_tensor(sel, device=env.device))
    orc.reset(seed=31, env_sel=sel.astype(np.uint8))
    _compare_state(env, orc, "after partial reset")
    for s in range(6):
        act = _random_actions(rng, orc.pos, orc.mask[:, :, :30], 30)
        env.step(torch.as_tensor(act, device=env.device))
        orc.step(act)
    _compare_state(env, orc, "steps after partial reset")
    env.reset(seed=77)
    orc.reset(seed=77)
    _compare_state(env, orc, "after reseed")
    rec, ref = env.rollout(20), orc.rollout(20)
    np.testing.assert_array_equal(_np(rec["action"]), ref["action"])
    env.close()


def test_rollout_metrics_on_device(sy):
    from student_mechanism_design_amd import metrics as M
    boards = sy.sample_board_pool(2, 60, 100, seed=3)
    env = sy.BatchedScotlandYardEnv(512, boards, 4, 12, np.full(11, 0.5), seed=1, reveal_interval=5)
    rec = env.rollout(96)
    m = M.rollout_metrics(rec, env.N, reveal_interval=5)
    done = (_np(rec["terminated"]) | _np(rec["truncated"])).astype(bool)
    assert int(m["num_episodes"]) == done.sum() > 0
    assert int(m["num_reveals"]) == (((_np(rec["t"]) + 1) % 5) == 0).sum() > 0
    assert 0.0 < float(m["mean_belief_ce"]) < np.log(env.N) + 1e-3 and float(m["mean_belief_ce_all_steps"]) > 0.0
    assert int(m["mrx_wins"]) + int(m["police_wins"]) == done.sum()
    assert 0.0 <= float(m["win_rate"]) <= 1.0 and float(m["mean_episode_length"]) >= 1.0
    # at reveal steps the belief is a delta on MrX: cross-entropy ~ 0 there
    t = _np(rec["t"])
    ce = _np(M.belief_cross_entropy(rec["belief"][..., : env.N], rec["pos"][..., 0]))
    reveal = (t > 0) & (t % 5 == 0)
    assert reveal.any() and np.abs(ce[reveal]).max() < 1e-5
    env.close()
    # the same reductions on the GPU against the goldens of the unmodified eval/metrics.py
    # (tests/golden/metrics_reference.json, oracle/capture_metrics.py) — not only in the CPU suite
    from tests.helpers import metrics_golden_record
    with open(os.path.join(GOLDEN, "metrics_reference.json")) as f:
        g = json.load(f)
    grec = metrics_golden_record(g, device="cuda")
    gm = M.rollout_metrics(grec, int(g["num_nodes"]), reveal_interval=int(g["reveal_interval"]))
    a = g["aggregated"]
    assert gm["win_rate"].is_cuda
    for k in ("num_episodes", "mrx_wins", "police_wins"):
        assert int(gm[k]) == int(a[k]), k
    for k in ("win_rate", "mean_episode_length", "mean_time_to_catch", "mean_survival_time"):
        assert abs(float(gm[k]) - a[k]) < 1e-4 * max(1.0, abs(a[k])), k
    assert abs(float(gm["mean_belief_ce"]) - a["mean_belief_ce"]) < 1e-9
    assert abs(float(gm["belief_ce_std"]) - a["belief_ce_std"]) < 1e-9
    for c in g["ce_cases"]:
        got = M.belief_cross_entropy(torch.tensor(c["belief"], dtype=torch.float64, device="cuda"),
                                     torch.tensor(c["true_index"], device="cuda"))
        assert abs(float(got) - c["ce"]) < 1e-10


def test_device_board_sampler_statistics_and_use(sy, ol):
    """sy_sample_boards: structure of ConnectedGraph.sample (graph_layout.py:9-80) — spanning tree first,
    extras only between nodes below the degree cap, no duplicate / self edges, weights 1..4, requested
    edge count — and the same degree statistics as the host sampler; the pool drives the engine."""
    N, E, G = 200, 400, 48
    pool = sy.sample_board_pool_device(G, N, E, seed=5)
    pk = pool.to_packed()
    assert pool.num_edges == E and len(pk.boards) == G
    dev_deg = []
    for b in pk.boards:
        assert b.num_edges == E and b.edges.min() >= 1 and b.edges.max() <= 4
        und = {tuple(sorted(x)) for x in b.edge_links.tolist()}
        assert len(und) == E and all(u != v for u, v in und)
        tree, extra = b.edge_links[: N - 1], b.edge_links[N - 1:]
        seen = {int(tree[0, 0])}
        for u, v in tree:                       # (visited, new) pairs: a spanning tree grown Prim-style
            assert int(u) in seen and int(v) not in seen
            seen.add(int(v))
        assert len(seen) == N
        deg = np.bincount(tree.reshape(-1), minlength=N)
        for u, v in extra:
            assert deg[u] < 4 and deg[v] < 4
            deg[u] += 1
            deg[v] += 1
        dev_deg.append(deg)
    np.testing.assert_array_equal(pk.ell, np.stack([sy.pack_ell(b) for b in pk.boards]))
    np.testing.assert_array_equal(pk.apsp, np.stack([sy.all_pairs_shortest_paths(b) for b in pk.boards]))
    np.testing.assert_allclose(pk.inv_deg, np.stack([sy.graph.inverse_degree(b, pool.node_stride) for b in pk.boards]))
    host = sy.sample_board_pool(G, N, E, seed=11)
    host_deg = [np.bincount(b.edge_links.reshape(-1), minlength=N) for b in host]
    hd = np.bincount(np.concatenate(dev_deg), minlength=17)[:17] / (G * N)
    hh = np.bincount(np.concatenate(host_deg), minlength=17)[:17] / (G * N)
    assert np.abs(hd - hh).max() < 0.02, (hd, hh)       # same degree distribution (sampling error ~0.005)
    assert abs(np.mean([d.max() for d in dev_deg]) - np.mean([d.max() for d in host_deg])) < 1.0
    assert len({b.edge_links.tobytes() for b in pk.boards}) == G      # boards differ
    # ... and both samplers against what the UNMODIFIED reference sampler realises (tests/golden/sampler_stats.json,
    # oracle/capture_sampler_stats.py): degree / tree-degree / max-degree / weight histograms, realised edge counts
    from student_mechanism_design_amd.graph import sample_boards_device_raw
    from tests.helpers import assert_sampler_stats_close, sampler_stats
    with open(os.path.join(GOLDEN, "sampler_stats.json")) as f:
        gold = json.load(f)
    for c in gold["configs"]:
        n, e = c["nodes"], c["edges_requested"]
        links, w, cnt = sample_boards_device_raw(min(c["boards"], 512), n, e, seed=900 + n)
        ok = cnt >= 0
        assert ok.mean() > 0.99                                        # a row wider than the ELL is a ~never event
        got = sampler_stats(n, [links[i, : cnt[i]] for i in np.nonzero(ok)[0]], [w[i, : cnt[i]] for i in np.nonzero(ok)[0]])
        assert_sampler_stats_close(c, got, what=f"device sampler N={n} E={e}")
    # unreachable edge counts: every board is redrawn to the first board's realised count
    sat = sy.sample_board_pool_device(6, 30, 70, seed=2)
    assert sat.num_edges < 70 and len(sat) == 6
    tree_only = sy.sample_board_pool_device(3, 12, None, seed=1)
    assert tree_only.num_edges == 11
    # the device-built pool runs the engine and still matches the oracle
    env = sy.BatchedScotlandYardEnv(64, pool, 4, 20, np.full(11, 0.5), seed=3, reveal_interval=5)
    graphs = [ol.OracleGraph(N, b.edge_links, b.edges.astype(np.int32)) for b in pk.boards]
    orc = ol.OracleBatch(graphs, env.env_graph_host, 64, 4, 20, node_stride=env.NS, weights=np.full(11, 0.5),
                         tables=sy.reward_tables(), reveal_interval=5)
    orc.reset(seed=3)
    rec, ref = env.rollout(40), orc.rollout(40)
    for k in ("pos", "action", "mask", "reward", "terminated"):
        np.testing.assert_array_equal(_np(rec[k]), ref[k], err_msg=k)
    env.close()


def _random_board(sy, rng, n, extra, wmax, wmin=0):
    """Connected board: random spanning tree + `extra` random edges (duplicates dropped), weights in
    [wmin, wmax] — including zero-weight edges, which the reference's weights never are but the ABI admits."""
    order = rng.permutation(n)
    links = set()
    for i in range(1, n):
        u, v = int(order[i]), int(order[rng.integers(0, i)])
        links.add((min(u, v), max(u, v)))
    deg = np.zeros(n, int)
    for u, v in links:
        deg[u] += 1
        deg[v] += 1
    tries = 0
    while extra > 0 and tries < 50 * n:
        tries += 1
        u, v = int(rng.integers(0, n)), int(rng.integers(0, n))
        e = (min(u, v), max(u, v))
        if u == v or e in links or deg[u] >= 15 or deg[v] >= 15:
            continue
        links.add(e)
        deg[u] += 1
        deg[v] += 1
        extra -= 1
    links = sorted(links)
    w = rng.integers(wmin, wmax + 1, size=len(links))
    return sy.make_board(n, np.array(links, dtype=np.int32), w)


@pytest.mark.parametrize("case", range(40))
def test_randomised_configurations_match_the_oracle(sy, ol, case):
    """Engine vs oracle over randomly drawn shapes and switches: node / police counts, tiny budgets, zero and
    heavy edge weights, short episode caps (timeouts), reveal schedules, police evidence, one-hot priors,
    board pools, odd batch sizes and block sizes — fused rollout, then caller-action steps from the live state."""
    rng = np.random.default_rng(4242 + case)
    N = int(rng.integers(8, 90))
    P = int(rng.integers(1, 8))
    if P + 1 > N - 2:
        P = max(1, N - 3)
    G = int(rng.integers(1, 4))
    money = int(rng.choice([0, 1, 2, 3, 5, 9, 30]))
    wmax = int(rng.choice([1, 2, 4, 7]))
    wmin = int(rng.choice([0, 1]))
    B = int(rng.integers(3, 70))
    T = int(rng.integers(20, 70))
    kw = dict(reveal_interval=int(rng.integers(0, 5)), police_evidence=bool(rng.integers(0, 2)),
              belief_init_onehot=bool(rng.integers(0, 2)), max_timestep=int(rng.choice([3, 8, 20, 250])),
              waves_per_block=int(rng.choice([0, 0, 2, 3, 4, 5, 6, 7, 8, 12, 16])))
    boards = [_random_board(sy, rng, N, int(rng.integers(0, 2 * N)), wmax, wmin) for _ in range(G)]
    weights = rng.uniform(0.0, 1.0, 11)
    weights[rng.integers(0, 11)] = 0.0
    env = sy.BatchedScotlandYardEnv(B, boards, P, money, weights, seed=case, **kw)
    graphs = [ol.OracleGraph(N, b.edge_links, b.edges.astype(np.int32)) for b in boards]
    orc = ol.OracleBatch(graphs, env.env_graph_host, B, P, money, max_t=kw["max_timestep"], node_stride=env.NS,
                         weights=weights, tables=sy.reward_tables(), reveal_interval=kw["reveal_interval"],
                         police_evidence=kw["police_evidence"], belief_init_onehot=kw["belief_init_onehot"])
    orc.reset(seed=case)
    _compare_state(env, orc, f"case {case} after reset")
    rec = env.rollout(T)
    ref = orc.rollout(T)
    for k in ("pos", "t", "action", "terminated", "truncated", "winner", "mask", "reward"):
        np.testing.assert_array_equal(_np(rec[k]), ref[k], err_msg=f"case {case} {k}")
    np.testing.assert_array_equal(_np(rec["budget"]), ref["money"], err_msg="budget")
    np.testing.assert_allclose(_np(rec["belief"]), ref["belief"], rtol=0, atol=BELIEF_TOL)
    _compare_state(env, orc, f"case {case} after rollout")
    for s in range(6):
        act = _random_actions(rng, orc.pos, orc.mask[:, :, :N], N)
        env.step(torch.as_tensor(act, device=env.device))
        orc.step(act)
        _compare_state(env, orc, f"case {case} step {s}")
    env.close()
