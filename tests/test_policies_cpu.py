"""CPU tests of the torch policies (device-agnostic code; the env side is covered by the GPU tests)."""
import numpy as np
import pytest
import torch

import student_mechanism_design_amd as sy
from student_mechanism_design_amd import policies as pol


def _fake_obs(B, N, P, seed=0):
    g = torch.Generator().manual_seed(seed)
    pos = torch.stack([torch.randperm(N, generator=g)[: P + 1] for _ in range(B)]).int()
    mask = torch.rand(B, P + 1, N, generator=g) < 0.05
    mask[0, 1] = False   # an agent without legal moves
    return {"MrX_pos": pos[:, 0], "Polices_pos": pos[:, 1:], "agent_position": pos, "action_mask": mask,
            "belief_map": torch.full((B, N), 1.0 / N)}


def test_mappo_policy_shapes_and_masking():
    B, N, P = 32, 40, 3
    obs = _fake_obs(B, N, P)
    net = pol.MappoPolicy(N, P, hidden_size=16)
    pr = net.probs(obs)
    assert pr.shape == (B, P + 1, N)
    torch.testing.assert_close(pr.sum(-1), torch.ones(B, P + 1))
    a, logp, v = net.act(obs, generator=torch.Generator().manual_seed(1))
    assert a.shape == (B, P + 1) and a.dtype == torch.int32 and logp.shape == (B, P + 1) and v.shape == (B,)
    legal = torch.gather(obs["action_mask"], -1, a.clamp_min(0).long().unsqueeze(-1)).squeeze(-1)
    assert bool((legal | (a < 0)).all()) and int(a[0, 1]) == -1
    # observations as the reference trainer builds them (mappo_trainer.py:173,197)
    mrx, police = net.observations(obs)
    assert mrx.sum(-1).eq(1).all() and police.sum(-1).eq(P).all()
    # one PPO step decreases nothing weird: losses finite, gradients flow to actors and critic
    adv = torch.randn(B, P + 1)
    newlp = torch.log(torch.gather(net.probs(obs), -1, a.clamp_min(0).long().unsqueeze(-1)).squeeze(-1) + 1e-8)
    al, cl = pol.ppo_loss(newlp, logp, adv, net.value(obs), torch.randn(B))
    (al + cl).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.parametrize("fixture", ["mappo_networks_reference.npz", "mappo_networks_reference_h128.npz"])
def test_mappo_policy_matches_the_reference_networks(fixture):
    """tests/golden/mappo_networks_reference*.npz (hidden 8, and the reference's default hidden 128): weights, inputs and outputs of the UNMODIFIED AgentPolicy /
    CentralCritic (oracle/capture_mappo_networks.py).  MappoPolicy loaded with those weights must reproduce the
    reference's action probabilities and values, through both the reference-shaped and the one-hot-free path."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture))
    N, P, H = int(g["N"]), int(g["P"]), int(g["H"])
    net = pol.MappoPolicy(N, P, hidden_size=H)
    with torch.no_grad():
        for k in range(P + 1):
            net.actors[k][0].weight.copy_(torch.from_numpy(g[f"actor{k}.actor.0.weight"]))
            net.actors[k][0].bias.copy_(torch.from_numpy(g[f"actor{k}.actor.0.bias"]))
            net.actors[k][2].weight.copy_(torch.from_numpy(g[f"actor{k}.actor.2.weight"]))
            net.actors[k][2].bias.copy_(torch.from_numpy(g[f"actor{k}.actor.2.bias"]))
        net.critic[0].weight.copy_(torch.from_numpy(g["critic.critic.0.weight"]))
        net.critic[0].bias.copy_(torch.from_numpy(g["critic.critic.0.bias"]))
        net.critic[2].weight.copy_(torch.from_numpy(g["critic.critic.2.weight"]))
        net.critic[2].bias.copy_(torch.from_numpy(g["critic.critic.2.bias"]))
    pos = torch.from_numpy(g["pos"])
    obs = {"MrX_pos": pos[:, 0], "Polices_pos": pos[:, 1:]}
    with torch.no_grad():
        np.testing.assert_allclose(net.probs(obs).numpy(), g["probs"], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(net.value(obs).numpy(), g["value"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(net.probs_fast(obs).numpy(), g["probs"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(net.value_fast(obs).numpy(), g["value"], rtol=1e-5, atol=1e-6)


def test_mappo_fast_inference_path_is_the_same_function():
    B, N, P = 48, 40, 4
    obs = _fake_obs(B, N, P, seed=3)
    net = pol.MappoPolicy(N, P, hidden_size=16)
    torch.testing.assert_close(net.probs_fast(obs), net.probs(obs), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(net.value_fast(obs), net.value(obs), rtol=1e-5, atol=1e-6)
    a1, lp1, v1 = net.act(obs, generator=torch.Generator().manual_seed(5))
    a2, lp2, v2 = net.act_fast(obs, generator=torch.Generator().manual_seed(5))
    assert torch.equal(a1 < 0, a2 < 0)
    legal = torch.gather(obs["action_mask"], -1, a2.clamp_min(0).long().unsqueeze(-1)).squeeze(-1)
    assert bool((legal | (a2 < 0)).all())
    torch.testing.assert_close(v1, v2, rtol=1e-5, atol=1e-6)


def _random_gnn(num_agents, with_belief, seed):
    torch.manual_seed(seed)
    net = pol.GnnQPolicy(num_agents, with_belief=with_belief)
    with torch.no_grad():                                 # spread the parameters so that every term of the layer matters
        for m in (net.mrx, net.police):
            for conv in (m.conv1, m.conv2):
                conv.bias.normal_(0.0, 0.5)
                conv.phi.weight.normal_(0.0, 0.8)
            m.out.weight.normal_(0.0, 1.0)
    return net


@pytest.mark.parametrize("directed", [True, False])
def test_gnn_model_on_gather_tables_matches_the_independent_restatement(directed):
    """`AntiSymmetricConvEll` / `GnnQModel` (gather over the <= 16 sources of a node) against oracle/gnn_oracle.py:
    float64, the propagation matrix D^-1/2 (A + I) D^-1/2 built DENSE, edge by edge, from the board's edge list — a
    different formulation of the published layer, not the same expression typed twice.  directed=True is the
    reference's data flow (every stored edge once, training/utils.py:170)."""
    from oracle import gnn_oracle as go
    N, P, B = 37, 3, 11
    boards = sy.sample_board_pool(3, N, 70, seed=2)
    tabs = pol.GcnTables(boards, directed=directed)
    env_graph = torch.tensor([0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1])
    obs = _fake_obs(B, N, P, seed=4)
    obs["belief_map"] = torch.rand(B, N)
    for with_belief in (False, True):
        net = _random_gnn(P + 1, with_belief, seed=7)
        with torch.no_grad():
            q = net.q_values(obs, tabs.for_envs(env_graph), N).double().numpy()           # [B, 2, N]
        pos = obs["agent_position"].numpy()
        x = go.node_features(pos, N, obs["belief_map"].numpy() if with_belief else None)
        for b in range(B):
            a_hat = go.propagation_matrix(N, boards[int(env_graph[b])].edge_links, directed=directed)
            for m, model in enumerate((net.mrx, net.police)):
                want = go.gnn_q(x[b: b + 1], a_hat, go.params_of(model))[0]
                np.testing.assert_allclose(q[b, m], want, rtol=0, atol=2e-6)
        # the propagation is not the identity: the board matters
        assert np.abs(q[0, 0] - q[0, 0].mean()).max() > 1e-3
    # the tables themselves: coefficients of a hand-checked path 0 -> 1 -> 2 (directed) / 0 - 1 - 2
    path = sy.make_board(3, [[0, 1], [1, 2]], [1, 1])
    t = pol.GcnTables([path], directed=directed)
    a_hat = go.propagation_matrix(3, path.edge_links, directed=directed)
    dense = np.zeros((3, 3))
    for v in range(3):
        dense[v, v] = float(t.self_coef[0, v])
        for k in range(16):
            u = int(t.nbr[0, v, k])
            if u >= 0:
                dense[v, u] += float(t.coef[0, v, k])
    np.testing.assert_allclose(dense, a_hat, rtol=1e-6)
    if directed:
        np.testing.assert_allclose(a_hat, [[1, 0, 0], [1 / np.sqrt(2), 0.5, 0], [0, 0.5, 0.5]], rtol=1e-12)


def test_gnn_policy_features_and_greedy_actions():
    from oracle import gnn_oracle as go
    B, N, P = 16, 30, 2
    obs = _fake_obs(B, N, P, seed=1)
    boards = sy.sample_board_pool(2, N, 50, seed=1)
    tabs = pol.GcnTables(boards)
    env_graph = torch.arange(B) % 2
    net = _random_gnn(P + 1, True, seed=3)
    x = net.features(obs, N)
    assert x.shape == (B, N, P + 2) and x[..., : P + 1].sum() == B * (P + 1)
    a, _, _ = net.act_greedy(obs, tabs.for_envs(env_graph))
    legal = torch.gather(obs["action_mask"], -1, a.clamp_min(0).long().unsqueeze(-1)).squeeze(-1)
    assert bool((legal | (a < 0)).all()) and bool((a[0, 1] == -1))                      # (agent without legal moves -> None)
    with torch.no_grad():
        q = net.q_values(obs, tabs.for_envs(env_graph), N).double().numpy()
    want, margin = go.greedy_actions(q[:, 0], q[:, 1], obs["action_mask"].numpy())
    decided = margin > 1e-6
    assert (a.numpy()[decided] == want[decided]).all()


def test_device_policy_and_sampler_refuse_to_run_without_a_gpu():
    """No CPU fallback: the HIP-backed policy pieces fail loudly on CPU tensors."""
    import pytest
    from student_mechanism_design_amd import collector as col
    net = pol.MappoPolicy(12, 2, hidden_size=8)
    with pytest.raises(sy.EngineError):
        pol.DeviceMappoPolicy(net)
    with pytest.raises(sy.EngineError):
        col.DeviceMaskedSampler(torch.device("cpu"))


def test_policy_oracle_philox_and_softmax_are_pinned():
    """oracle/policy_oracle.py (the restatement the GPU tests hold the in-kernel policy to) against independent
    implementations: its numpy Philox4x32-7 against the C oracle's (which the env parity tests pin through the action
    draws of the random policy), and its masked, renormalised log-probabilities against `masked_categorical_sample` on
    `MappoPolicy.probs` — the torch restatement that the goldens of the unmodified reference pin."""
    from oracle import oracle_lib as ol, policy_oracle as po
    from student_mechanism_design_amd import collector as col
    rng = np.random.default_rng(0)
    for _ in range(20):
        gid = int(rng.integers(0, 2**40))
        ctr, idx = int(rng.integers(0, 2**32)), int(rng.integers(0, 8))
        k0, k1 = int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32))
        want = ol.philox(gid & 0xFFFFFFFF, gid >> 32, ctr, (1 << 8) | idx, k0, k1)
        got = po.philox4x32_7(np.array([gid], dtype=np.uint64), np.array([ctr]), 1, np.array([idx]), k0, k1)[0]
        assert list(map(int, got)) == list(map(int, want))
    N, P, H, B = 30, 3, 16, 40
    import student_mechanism_design_amd as sy
    board = sy.sample_board(N, 55, rng=np.random.default_rng(4))
    ell = sy.pack_ell(board)
    torch.manual_seed(0)
    net = pol.MappoPolicy(N, P, hidden_size=H)
    with torch.no_grad():
        for a in net.actors:
            a[2].bias.normal_(0.0, 2.0)
    pos = np.stack([rng.permutation(N)[: P + 1] for _ in range(B)])[None]           # [1][B][A]
    budget = np.concatenate([np.full((1, B, 1), 1000), rng.integers(0, 5, size=(1, B, P))], axis=2)
    weights = {"W1": np.stack([a[0].weight.detach().numpy() for a in net.actors]), "b1": np.stack([a[0].bias.detach().numpy() for a in net.actors]),
               "W2": np.stack([a[2].weight.detach().numpy() for a in net.actors]), "b2": np.stack([a[2].bias.detach().numpy() for a in net.actors])}
    d = po.policy_draws(pos, budget, np.zeros(B, dtype=np.uint32), np.arange(B), lambda b: ell, weights, stream_key=77)
    # the torch path: masks from the ELL, the module's softmax, select_action's normalisation
    mask = np.zeros((B, P + 1, N), dtype=bool)
    for b in range(B):
        for a in range(P + 1):
            row = ell[pos[0, b, a]]
            ok = (row >> 16) <= budget[0, b, a]
            mask[b, a, (row & 0xFFFF)[ok]] = True
    assert (mask.sum(-1) == d["count"][0]).all()
    with torch.no_grad():
        probs = net.probs({"MrX_pos": torch.from_numpy(pos[0, :, 0]), "Polices_pos": torch.from_numpy(pos[0, :, 1:])})
        _, _, p = col.masked_categorical_sample(probs.double(), torch.from_numpy(mask), generator=torch.Generator().manual_seed(1))
    norm = (p / p.sum(-1, keepdim=True)).numpy()
    for b in range(B):
        for a in range(P + 1):
            for k in range(16):
                n = d["nodes"][0, b, a, k]
                if n >= 0:
                    assert abs(np.exp(d["logp_entries"][0, b, a, k]) - norm[b, a, n]) < 1e-6
    # arg-max consistency and the checker's own bookkeeping
    best = d["keys"][0].argmax(-1)
    has = d["count"][0] > 0
    assert (np.take_along_axis(d["nodes"][0], best[..., None], -1)[..., 0][has] == d["action"][0][has]).all()
    stats = po.check_recorded_policy_rollout(d["action"], d["log_prob"], d)
    assert stats["decided"] + stats["undecided"] + stats["in_underflow_band"] == int(has.sum())
    bad = d["action"].copy()
    pick = np.argwhere((d["count"] > 1) & (d["margin"] > 0.5))[0]
    others = [n for n in d["nodes"][tuple(pick)] if n >= 0 and n != d["action"][tuple(pick)]]
    bad[tuple(pick)] = others[0]
    lp_bad = d["log_prob"].copy()
    lp_bad[tuple(pick)] = d["logp_entries"][tuple(pick)][list(d["nodes"][tuple(pick)]).index(others[0])]
    with pytest.raises(AssertionError, match="decided draws differ"):
        po.check_recorded_policy_rollout(bad, lp_bad, d)


def test_mappo_updater_losses_are_the_masked_softmax_surrogate():
    """`update.MappoUpdater` computes the new log-probabilities from a log-sum-exp over the <= 16 affordable entries of the
    agent's ELL row.  Against the plain form — softmax over all N nodes (the pinned `MappoPolicy.probs`), times the mask,
    renormalised as `select_action` does (mappo_agent.py:112-134), clipped surrogate + critic MSE (:260-293) — the losses
    must agree, and an update must move the parameters without any NaN."""
    from student_mechanism_design_amd.update import MappoUpdater
    rng = np.random.default_rng(3)
    N, P, H, T, B = 40, 3, 16, 6, 10
    A = P + 1
    boards = sy.sample_board_pool(2, N, 70, seed=5)
    pool = sy.pack_pool(boards)
    ell = torch.from_numpy(pool.ell.view(np.int32).copy())
    env_graph = torch.arange(B) % 2
    torch.manual_seed(1)
    net = pol.MappoPolicy(N, P, hidden_size=H)
    pos = torch.stack([torch.stack([torch.randperm(N)[:A] for _ in range(B)]) for _ in range(T)]).int()     # [T, B, A]
    budget = torch.cat([torch.full((T, B, 1), 1000), torch.randint(0, 4, (T, B, P))], -1).int()
    mask = torch.zeros((T, B, A, N), dtype=torch.bool)
    for t in range(T):
        for b in range(B):
            for a in range(A):
                row = pool.ell[int(env_graph[b]), int(pos[t, b, a])]
                ok = (row >> 16) <= int(budget[t, b, a])
                mask[t, b, a, (row & 0xFFFF)[ok].astype(np.int64)] = True
    cnt = mask.sum(-1)
    act = torch.where(cnt > 0, torch.multinomial(mask.reshape(-1, N).float() + 1e-9, 1).reshape(T, B, A), torch.full((T, B, A), -1))
    act = torch.where(cnt > 0, act, torch.full_like(act, -1)).int()
    old_lp = torch.where(cnt > 0, -torch.log(cnt.float().clamp_min(1)) + 0.1 * torch.randn(T, B, A), torch.zeros(T, B, A))
    returns = torch.randn(T, B, A)
    rec = {"pos": pos, "budget": budget, "action": act, "log_prob": old_lp}
    up = MappoUpdater(net, ell, env_graph, minibatch=T * B)
    R = T * B
    adv = ((returns - returns.mean()) / (returns.std() + 1e-8)).reshape(R, A)
    al, cl = up._losses(pos.reshape(R, A).long(), budget.reshape(R, A).long(), act.reshape(R, A).long(), old_lp.reshape(R, A), adv,
                        returns.reshape(R, A).sum(-1), env_graph.repeat(T))
    # the plain form
    obs = {"MrX_pos": pos.reshape(R, A)[:, 0], "Polices_pos": pos.reshape(R, A)[:, 1:]}
    pm = net.probs(obs) * mask.reshape(R, A, N).float()
    pm = pm / pm.sum(-1, keepdim=True).clamp_min(1e-30)
    valid = (act.reshape(R, A) >= 0).float()
    new_lp = torch.log(torch.gather(pm, -1, act.reshape(R, A).long().clamp_min(0).unsqueeze(-1)).squeeze(-1).clamp_min(1e-30)) * valid
    al_ref, cl_ref = pol.ppo_loss(new_lp, old_lp.reshape(R, A) * valid, adv, net.value(obs), returns.reshape(R, A).sum(-1))
    torch.testing.assert_close(al, al_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cl, cl_ref, rtol=1e-5, atol=1e-6)
    before = [p.detach().clone() for p in net.parameters()]
    a2, c2 = up.update(rec, returns)
    assert torch.isfinite(a2) and torch.isfinite(c2)
    assert all(torch.isfinite(p).all() for p in net.parameters())
    assert all(not torch.equal(x, y) for x, y in zip(before, net.parameters()))
