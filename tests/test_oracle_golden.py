"""Pin the CPU oracle (oracle/sy_oracle.c) against goldens recorded from the unmodified reference.

Bit-exact: positions, budgets, masks, terminated/truncated, winner, visit counts, timestep.
Rewards: float64, rel/abs 1e-12 (only libm-vs-numpy exp/log1p rounding may differ).
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle_lib as ol
from tests.helpers import GOLDEN, engine_actions, load_trace, trace_index

TRACES = [e["file"] for e in trace_index()]


@pytest.mark.parametrize("name", TRACES)
def test_trace_replay_matches_reference(name):
    tr = load_trace(name)
    N, P = int(tr["N"]), int(tr["P"])
    g = ol.OracleGraph(N, tr["edge_links"], tr["edge_w"])
    ep = ol.OracleEpisode(g, P, int(tr["money0"]), tr["starts"], tr["weights"])
    np.testing.assert_array_equal(ep.masks(), tr["mask0"])
    T = tr["actions"].shape[0]
    for s in range(T):
        ended = ep.step(engine_actions(tr["actions"][s]))
        np.testing.assert_array_equal(ep.pos, tr["pos"][s], err_msg=f"pos step {s}")
        np.testing.assert_array_equal(ep.money, tr["money"][s], err_msg=f"money step {s}")
        assert bool(ep.flags[0]) == bool(tr["terminated"][s]), f"terminated step {s}"
        assert bool(ep.flags[1]) == bool(tr["truncated"][s]), f"truncated step {s}"
        assert int(ep.winner[0]) == int(tr["winner"][s]), f"winner step {s}"
        assert int(ep.t[0]) == int(tr["t_after"][s])
        np.testing.assert_array_equal(ep.visits, tr["visits"][s], err_msg=f"visits step {s}")
        np.testing.assert_array_equal(ep.masks(), tr["masks"][s], err_msg=f"masks step {s}")
        np.testing.assert_allclose(ep.reward, tr["reward"][s], rtol=1e-12, atol=1e-12, err_msg=f"reward step {s}")
        assert ended == bool(tr["terminated"][s] or tr["truncated"][s])


def test_golden_set_covers_the_edge_cases():
    """The fixture set must actually contain the SURVEY Appendix-A situations."""
    seen = dict(capture=False, timeout=False, no_money=False, none_action=False, invalid=False,
                blocked_police=False, debit=False, money_zero=False, p6=False, n200=False)
    for e in trace_index():
        tr = load_trace(e["file"])
        A = tr["actions"].shape[1]
        seen["p6"] |= A == 7
        seen["n200"] |= int(tr["N"]) == 200
        seen["none_action"] |= bool((tr["actions"] == -2).any())
        seen["money_zero"] |= bool((tr["money"][:, 1:] == 0).any())
        if tr["truncated"][-1]:
            seen["timeout"] = True
            assert tr["actions"].shape[0] == 252 and tr["t_after"][-1] == 252  # fires on the 252nd step
        if tr["terminated"][-1] and tr["winner"][-1] == 1:
            seen["capture"] = True
        if tr["terminated"][-1] and tr["winner"][-1] == 2:
            seen["no_money"] = True
        prev_pos = tr["starts"]
        prev_money = np.array([1000] + [int(tr["money0"])] * (A - 1))
        for s in range(tr["actions"].shape[0]):
            act, pos, money = tr["actions"][s], tr["pos"][s], tr["money"][s]
            seen["debit"] |= bool((money[1:] < prev_money[1:]).any())
            for a in range(A):
                if act[a] >= 0 and pos[a] == prev_pos[a] and act[a] != prev_pos[a]:
                    seen["invalid"] = True  # asked to go somewhere, stayed
                    if a >= 1 and act[a] in list(pos[1:]):
                        seen["blocked_police"] = True
            prev_pos, prev_money = pos, money
    missing = [k for k, v in seen.items() if not v]
    assert not missing, f"golden traces miss: {missing}"


def test_env_test_case_outcome():
    """reference test/env_test.py:69-75: all actions -1 on step 0 -> terminated, MrX wins, rewards 1/0/0."""
    tr = load_trace("trace_s0_n15_p2_m10_noop_ep0.npz")
    assert tr["actions"].shape[0] == 1 and (tr["actions"] == -1).all()
    assert tr["terminated"][0] and not tr["truncated"][0] and tr["winner"][0] == 2
    np.testing.assert_array_equal(tr["reward"][0], [1.0, 0.0, 0.0])


def _kats():
    with open(os.path.join(GOLDEN, "action_mask_kats.json")) as f:
        return json.load(f)


def test_action_mask_known_answers():
    cases = _kats()
    assert sum(c["tag"].startswith("ref_") for c in cases) == 6
    for c in cases:
        adj = np.array(c["adjacency"], dtype=float)
        n = adj.shape[0]
        tolls = c["tolls"]
        tmat = ol.normalize_tolls(tolls if tolls is None or np.isscalar(tolls) else np.array(tolls), n)
        w = None if c["edge_weights"] is None else np.array(c["edge_weights"], dtype=float)
        got = ol.action_mask_dense(adj, c["current_node"], c["budget"], tolls=tmat, edge_weights=w)
        np.testing.assert_array_equal(got, np.array(c["mask"], dtype=bool), err_msg=c["tag"])


def test_mask_equals_possible_moves_on_traces():
    """yard.py:297-317 (dense compute_action_mask) == yard.py:420-472 node set, as the survey probed."""
    for name in TRACES[:12]:
        tr = load_trace(name)
        g = ol.OracleGraph(int(tr["N"]), tr["edge_links"], tr["edge_w"])
        for s in range(min(5, tr["pos"].shape[0])):
            for a in range(tr["pos"].shape[1]):
                nodes, _ = g.possible_moves(tr["pos"][s][a], tr["money"][s][a])
                np.testing.assert_array_equal(np.nonzero(tr["masks"][s][a])[0], nodes)


def _graph_from_adj(adj):
    adj = np.asarray(adj)
    n = adj.shape[0]
    links = [(i, j) for i in range(n) for j in range(i + 1, n) if adj[i, j]]
    return ol.OracleGraph(n, np.array(links, dtype=np.int32).reshape(-1, 2), np.ones(len(links), dtype=np.int32))


def test_belief_reference_seeded_test():
    """test/test_belief_update.py: 3-node path, uniform prior, hint [1] then reveal 2."""
    with open(os.path.join(GOLDEN, "belief_reference.json")) as f:
        ref = json.load(f)["ref_test"]
    g = _graph_from_adj(ref["adjacency"])
    b = np.full(3, 1 / 3)
    b = ol.belief_update(g, b, hint=[1])
    np.testing.assert_allclose(b, [1 / 42, 40 / 42, 1 / 42], rtol=1e-14)
    assert np.isclose(b.sum(), 1.0)
    # the stochastic tracker's realized 20-particle estimate is within sampling error of the filter
    assert np.argmax(ref["after_hint"]) == np.argmax(b)
    b = ol.belief_update(g, b, reveal=2)
    np.testing.assert_array_equal(b, ref["after_reveal"])
    assert b.argmax() == 2 and np.isclose(b.sum(), 1.0)


def test_belief_filter_matches_particle_tracker_monte_carlo():
    """Deterministic forward filter == expectation of ParticleBeliefTracker (4e5 particles, no resampling)."""
    with open(os.path.join(GOLDEN, "belief_reference.json")) as f:
        mc = json.load(f)["monte_carlo"]
    worst = 0.0
    for case in mc:
        g = _graph_from_adj(case["adjacency"])
        n = g.N
        b = np.full(n, 1.0 / n)
        for st in case["steps"]:
            if st["kind"] == "hint":
                b = ol.belief_update(g, b, hint=st["hint"])
            elif st["kind"] == "reveal":
                b = ol.belief_update(g, b, reveal=st["reveal"])
            else:
                b = ol.belief_update(g, b)
            ref = np.array(st["belief"])
            worst = max(worst, float(np.abs(b - ref).max()))
            assert np.isclose(b.sum(), 1.0)
    assert worst < 6e-3, worst  # Monte-Carlo error at 4e5 particles with un-resampled weights


def test_apsp_equals_reference_distances_implicitly():
    """Distances feed the golden rewards; spot-check symmetry/triangle on a golden graph."""
    tr = load_trace("trace_s10_n200_p4_m20_random_valid_ep0.npz")
    g = ol.OracleGraph(200, tr["edge_links"], tr["edge_w"])
    d = g.dist
    assert (d == d.T).all() and (np.diag(d) == 0).all() and d.max() < 4 * 199
    i, j, k = np.random.default_rng(0).integers(0, 200, (3, 2000))
    assert (d[i, j] <= d[i, k] + d[k, j]).all()
