#!/usr/bin/env python3
"""Record golden vectors from the UNMODIFIED reference (build container only).

TEST INFRASTRUCTURE — never imported by the product path, never run on the GPU box
(`/root/reference` does not exist there; only the small fixtures written here travel).

What it does
  * puts oracle/refstubs (shape-only gymnasium/pettingzoo stand-ins) and /root/reference/src on
    sys.path and imports `environment.yard.CustomEnvironment`, `environment.action_mask`,
    `environment.belief_module` exactly as they lie in the reference;
  * drives seeded episodes with scripted action policies that hit every edge case listed in
    SURVEY.md Appendix A, recording the *realized* graph, starts, actions and per-step outputs
    (positions, budgets, rewards, flags, winner, masks, police visit counts);
  * records compute_action_mask known answers (the reference's 5 test cases + random dense cases
    with scalar / vector / matrix tolls) and ParticleBeliefTracker outputs (the reference's seeded
    test + high-particle-count Monte-Carlo estimates that pin the deterministic forward filter).

Usage:  python oracle/capture_goldens.py [--out tests/golden] [--ref /root/reference]
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

WEIGHT_NAMES = [
    "Police_distance", "Police_group", "Police_position", "Police_time",
    "Mrx_closest", "Mrx_average", "Mrx_position", "Mrx_time",
    "Police_coverage", "Police_proximity", "Police_overlap_penalty",
]

NONE_ACTION = -2  # encoding of a Python `None` action in the fixtures (-1 is the env's own no-op)


class NullLogger:
    def log(self, *a, **k):
        pass

    def log_scalar(self, *a, **k):
        pass

    def log_plt(self, *a, **k):
        pass

    def log_weights(self, *a, **k):
        pass


VIS_OFF = {"visualize_game": False, "visualize_heatmap": False,
           "save_visualization": False, "save_dir": "/tmp/sy_vis"}


def make_weights(rng):
    return {k: float(np.round(rng.uniform(0.05, 0.95), 6)) for k in WEIGHT_NAMES}


def snapshot_masks(obs, agents, n):
    m = np.zeros((len(agents), n), dtype=np.bool_)
    for i, a in enumerate(agents):
        m[i] = obs[a]["action_mask"]
    return m


def visits_array(env, n):
    v = np.zeros(n, dtype=np.int32)
    for k, c in env.node_visit_counts.items():
        v[int(k)] = c
    return v


def choose_actions(kind, env, rng, P, N, step):
    """Scripted action policies (our own RNG stream; the env only sees the chosen ints)."""
    acts = []
    mrx = int(env.MrX_pos[0])
    pol = [int(p) for p in env.police_positions]
    for a in range(P + 1):
        pos = mrx if a == 0 else pol[a - 1]
        moves = [int(m) for m in env.get_possible_moves(a)]
        if kind == "noop":
            acts.append(-1)
        elif kind == "random_valid":
            acts.append(int(rng.choice(moves)) if moves else -1)
        elif kind == "timeout":
            # MrX wanders; police "act" with an invalid target (their own node): never skipped,
            # never move, never pay -> only the t>250 truncation can end the episode.
            if a == 0:
                acts.append(int(rng.choice(moves)) if moves else -1)
            else:
                acts.append(pos)
        elif kind == "chase":
            if a == 0:
                acts.append(int(rng.choice(moves)) if moves else -1)
            else:
                if moves:
                    d = [env.get_distance(m, mrx) for m in moves]
                    acts.append(moves[int(np.argmin(d))])
                else:
                    acts.append(-1)
        elif kind in ("mixed", "mixed_live"):
            u = rng.random()
            if u < 0.55 and moves:
                acts.append(int(rng.choice(moves)))
            elif u < 0.65:
                acts.append(-1)
            elif u < 0.70:
                acts.append(NONE_ACTION)
            elif u < 0.80:
                acts.append(int(rng.integers(0, N)))        # usually a non-neighbour
            elif u < 0.92:
                others = [mrx] + pol                          # walk onto somebody (block/capture)
                acts.append(int(rng.choice(others)))
            elif u < 0.96:
                acts.append(pos)                              # own node
            else:
                acts.append(int(rng.integers(N, N + 5)))      # out of range id
        elif kind == "swarm":
            # all police pick the same free neighbour of Police0 when possible (same-target races,
            # vacated-node moves), MrX random
            if a == 0:
                acts.append(int(rng.choice(moves)) if moves else -1)
            else:
                m0 = [int(m) for m in env.get_possible_moves(1)]
                tgt = int(rng.choice(m0)) if m0 else pol[0]
                acts.append(tgt if rng.random() < 0.7 else (int(rng.choice(moves)) if moves else -1))
        else:
            raise ValueError(kind)
    if kind == "mixed_live" and all(x in (-1, NONE_ACTION) for x in acts[1:]):
        acts[P] = pol[P - 1]  # keep the episode alive: an invalid (own-node) target is not a skip
    return acts


def record_trace(CustomEnvironment, seed, N, E, P, money, kind, max_steps, episodes=1):
    np.random.seed(seed)
    random.seed(seed)
    rng = np.random.default_rng(1000 + seed)
    weights = make_weights(rng)
    env = CustomEnvironment(number_of_agents=P, agent_money=money, reward_weights=weights,
                            logger=NullLogger(), epoch=0, graph_nodes=N, graph_edges=E,
                            vis_configs=VIS_OFF)
    traces = []
    for ep in range(episodes):
        obs, _ = env.reset(episode=ep)
        agents = list(env.possible_agents)
        n = env.board.nodes.shape[0]
        tr = {
            "N": n, "P": P, "money0": money,
            "edge_links": np.asarray(env.board.edge_links, dtype=np.int32),
            "edge_w": np.asarray(env.board.edges, dtype=np.int32),
            "weights": np.array([weights[k] for k in WEIGHT_NAMES], dtype=np.float64),
            "starts": np.array([env.MrX_pos[0]] + list(env.police_positions), dtype=np.int32),
            "mask0": snapshot_masks(obs, agents, n),
        }
        A = P + 1
        acts_l, pos_l, mon_l, rew_l, term_l, trunc_l, win_l, mask_l, vis_l, t_l = ([] for _ in range(10))
        for step in range(max_steps):
            acts = choose_actions(kind, env, rng, P, n, step)
            adict = {ag: (None if acts[i] == NONE_ACTION else acts[i]) for i, ag in enumerate(agents)}
            obs, rewards, terms, truncs, _ = env.step(adict)
            acts_l.append(acts)
            pos_l.append([int(env.MrX_pos[0])] + [int(p) for p in env.police_positions])
            mon_l.append([int(m) for m in env.agents_money])
            rew_l.append([float(rewards[a]) for a in agents])
            tv = [bool(terms[a]) for a in agents]
            uv = [bool(truncs[a]) for a in agents]
            assert all(x == tv[0] for x in tv) and all(x == uv[0] for x in uv)
            term_l.append(tv[0])
            trunc_l.append(uv[0])
            win_l.append({None: 0, "Police": 1, "MrX": 2}[env.current_winner])
            mask_l.append(snapshot_masks(obs, agents, n))
            vis_l.append(visits_array(env, n))
            t_l.append(int(env.timestep))
            if tv[0] or uv[0]:
                break
        tr.update(
            actions=np.array(acts_l, dtype=np.int32).reshape(-1, A),
            pos=np.array(pos_l, dtype=np.int32).reshape(-1, A),
            money=np.array(mon_l, dtype=np.int32).reshape(-1, A),
            reward=np.array(rew_l, dtype=np.float64).reshape(-1, A),
            terminated=np.array(term_l, dtype=np.bool_),
            truncated=np.array(trunc_l, dtype=np.bool_),
            winner=np.array(win_l, dtype=np.int8),
            masks=np.array(mask_l, dtype=np.bool_).reshape(-1, A, n),
            visits=np.array(vis_l, dtype=np.int32).reshape(-1, n),
            t_after=np.array(t_l, dtype=np.int32),
        )
        tr["kind"] = kind
        tr["seed"] = seed
        traces.append(tr)
    return traces


def capture_env(out, CustomEnvironment):
    plan = [
        # (seed, N, E, P, money, kind, max_steps, episodes)
        (0, 15, 20, 2, 10, "noop", 5, 1),            # == reference test/env_test.py sizes & actions
        (1, 15, 20, 2, 10, "random_valid", 80, 2),
        (2, 15, 20, 2, 10, "mixed", 120, 3),
        (3, 15, 20, 2, 3, "mixed", 120, 3),          # tiny budgets: money==0 skips, empty masks
        (4, 12, 11, 3, 6, "mixed", 120, 2),          # pure tree (E = N-1)
        (5, 20, 30, 4, 20, "chase", 60, 3),          # captures with debit
        (6, 10, 14, 2, 10, "timeout", 300, 1),       # t>250 truncation on the 252nd step
        (7, 25, 40, 6, 8, "swarm", 100, 2),          # P=6, same-target races
        (8, 30, 50, 4, 20, "mixed", 150, 2),
        (9, 8, 10, 5, 5, "mixed", 100, 3),           # crowded board (A=6 on 8 nodes)
        (10, 200, 400, 4, 20, "random_valid", 12, 1),  # BASELINE config-2 sizes (slow in reference)
        (11, 40, 70, 4, 2, "random_valid", 60, 2),   # budgets run dry -> no-money termination
        (12, 15, 20, 2, 10, "swarm", 80, 2),
        (13, 6, 7, 2, 10, "chase", 40, 3),
        (14, 15, 20, 2, 10, "mixed_live", 150, 3),
        (15, 24, 36, 4, 12, "mixed_live", 150, 3),
        (16, 18, 28, 3, 4, "mixed_live", 150, 2),
        (17, 50, 90, 6, 15, "mixed_live", 60, 1),
    ]
    index = []
    for (seed, N, E, P, money, kind, max_steps, episodes) in plan:
        t0 = time.time()
        traces = record_trace(CustomEnvironment, seed, N, E, P, money, kind, max_steps, episodes)
        for ep, tr in enumerate(traces):
            name = f"trace_s{seed}_n{N}_p{P}_m{money}_{kind}_ep{ep}.npz"
            meta = {k: tr.pop(k) for k in ("kind", "seed")}
            np.savez_compressed(os.path.join(out, name), **tr)
            T = tr["actions"].shape[0]
            index.append({"file": name, "N": int(tr["N"]), "E": int(tr["edge_links"].shape[0]),
                          "P": P, "money0": money, "steps": int(T),
                          "ended": bool(tr["terminated"][-1] or tr["truncated"][-1]),
                          "winner": int(tr["winner"][-1]), **meta})
        print(f"  trace seed={seed} N={N} P={P} {kind}: {len(traces)} episodes, {time.time()-t0:.1f}s")
    with open(os.path.join(out, "traces_index.json"), "w") as f:
        json.dump(index, f, indent=1)


def capture_masks(out, compute_action_mask):
    cases = []

    def run(adj, cur, budget, tolls=None, w=None, tag=""):
        r = compute_action_mask(np.asarray(adj), current_node=cur, budget=budget, tolls=tolls,
                                edge_weights=None if w is None else np.asarray(w))
        n = np.asarray(adj).shape[0]
        assert r.index_to_node == {i: i for i in range(n)} and r.node_to_index == r.index_to_node
        assert r.valid_actions == [int(i) for i in np.nonzero(r.mask)[0]]
        cases.append({
            "tag": tag, "adjacency": np.asarray(adj, dtype=float).tolist(), "current_node": int(cur),
            "budget": float(budget),
            "tolls": None if tolls is None else (float(tolls) if np.isscalar(tolls) else np.asarray(tolls, dtype=float).tolist()),
            "edge_weights": None if w is None else np.asarray(w, dtype=float).tolist(),
            "mask": [bool(b) for b in r.mask],
        })

    # the reference's own five known-answer cases (test/test_action_mask.py:9-124), inputs transcribed
    run([[0, 1, 1], [1, 0, 0], [1, 0, 0]], 0, 3, w=[[0, 2, 4], [2, 0, 0], [4, 0, 0]], tag="ref_budget_and_mapping")
    run(np.ones((2, 2)) - np.eye(2), 0, 0.5, tolls=0.25, tag="ref_scalar_toll_unaffordable")
    run(np.ones((2, 2)) - np.eye(2), 0, 1.5, tolls=0.25, tag="ref_scalar_toll_affordable")
    run([[0, 1, 1, 1], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]], 0, 100, tag="ref_fixed_mapping")
    run([[0, 1], [1, 0]], 0, 1, w=[[0, 100], [100, 0]], tag="ref_no_valid_moves")
    run([[0, 0, 0], [0, 0, 1], [0, 1, 0]], 0, 100, tag="ref_isolated_node")
    # random dense cases incl. vector / matrix tolls and inf off-edge weights (yard.py:404-418 layout)
    rng = np.random.default_rng(7)
    for i in range(40):
        n = int(rng.integers(2, 24))
        adj = (rng.random((n, n)) < 0.3).astype(float)
        adj = np.triu(adj, 1)
        adj = adj + adj.T
        w = np.where(adj > 0, rng.integers(1, 5, (n, n)).astype(float), np.inf)
        w = np.minimum(w, w.T)
        np.fill_diagonal(w, 0)
        cur = int(rng.integers(0, n))
        budget = float(rng.choice([0, 1, 2, 3, 4, 2.5, 1000]))
        mode = i % 4
        tolls = None
        if mode == 1:
            tolls = float(rng.choice([0.25, 1.0, 2.0]))
        elif mode == 2:
            tolls = rng.integers(0, 3, n).astype(float)
        elif mode == 3:
            tolls = rng.integers(0, 3, (n, n)).astype(float)
        run(adj, cur, budget, tolls=tolls, w=(None if i % 5 == 0 else w), tag=f"rand{i}")
    with open(os.path.join(out, "action_mask_kats.json"), "w") as f:
        json.dump(cases, f)
    print(f"  {len(cases)} action-mask cases")


def capture_belief(out, ParticleBeliefTracker):
    res = {}
    # (1) the reference's seeded test (test/test_belief_update.py:9-25), realized values
    adj3 = np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]])
    tr = ParticleBeliefTracker(num_nodes=3, num_particles=20, rng=np.random.default_rng(0))
    p0 = tr.state.particles.copy()
    b1 = tr.update(adj3, observation_hint=[1])
    b2 = tr.update(adj3, reveal=2)
    res["ref_test"] = {"adjacency": adj3.tolist(), "particles0": p0.tolist(),
                       "after_hint": b1.tolist(), "after_reveal": b2.tolist()}
    # (2) Monte-Carlo pins of the deterministic forward filter: many particles, several scenarios
    rng = np.random.default_rng(11)
    mc = []
    for case in range(6):
        n = int(rng.integers(5, 12))
        adj = (rng.random((n, n)) < 0.35).astype(int)
        adj = np.triu(adj, 1)
        adj = adj + adj.T
        if case == 3:
            adj[0, :] = 0
            adj[:, 0] = 0  # isolated node: particles there stay put
        K = 400_000
        tracker = ParticleBeliefTracker(num_nodes=n, num_particles=K, rng=np.random.default_rng(100 + case))
        steps = []
        script = [("none", None), ("hint", None), ("none", None), ("reveal", None), ("hint", None), ("none", None)]
        for kind, _ in script:
            if kind == "hint":
                hint = sorted(set(int(x) for x in rng.integers(0, n, size=int(rng.integers(1, 4)))))
                b = tracker.update(adj, observation_hint=hint)
                steps.append({"kind": "hint", "hint": hint, "belief": b.tolist()})
            elif kind == "reveal":
                node = int(rng.integers(0, n))
                b = tracker.update(adj, reveal=node)
                steps.append({"kind": "reveal", "reveal": node, "belief": b.tolist()})
            else:
                b = tracker.update(adj)
                steps.append({"kind": "none", "belief": b.tolist()})
        mc.append({"adjacency": adj.tolist(), "num_particles": K, "steps": steps})
        print(f"  belief MC case {case} (n={n}) done")
    res["monte_carlo"] = mc
    with open(os.path.join(out, "belief_reference.json"), "w") as f:
        json.dump(res, f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="env,mask,belief")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    sys.path.insert(0, os.path.join(HERE, "refstubs"))
    sys.path.insert(0, os.path.join(args.ref, "src"))
    sys.dont_write_bytecode = True
    only = args.only.split(",")
    from environment.yard import CustomEnvironment
    from environment.action_mask import compute_action_mask
    from environment.belief_module import ParticleBeliefTracker
    if "mask" in only:
        capture_masks(out, compute_action_mask)
    if "belief" in only:
        capture_belief(out, ParticleBeliefTracker)
    if "env" in only:
        capture_env(out, CustomEnvironment)
    print("goldens written to", out)


if __name__ == "__main__":
    main()
