"""CPU tests of the torch policies (device-agnostic code; the env side is covered by the GPU tests)."""
import numpy as np
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


def test_mappo_policy_matches_the_reference_networks():
    """tests/golden/mappo_networks_reference.npz: weights, inputs and outputs of the UNMODIFIED AgentPolicy /
    CentralCritic (oracle/capture_mappo_networks.py).  MappoPolicy loaded with those weights must reproduce the
    reference's action probabilities and values, through both the reference-shaped and the one-hot-free path."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mappo_networks_reference.npz"))
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


def test_antisymmetric_conv_matches_its_formula():
    torch.manual_seed(0)
    boards = sy.sample_board_pool(2, 12, 18, seed=1)
    pool = sy.pack_pool(boards)
    ell = torch.from_numpy(pool.ell.view(np.int32).copy())
    a_hat = pol.normalized_adjacency(ell, 12)
    # reference construction of D^-1/2 (A+I) D^-1/2
    for gi, b in enumerate(boards):
        A = np.zeros((12, 12))
        A[b.edge_links[:, 0], b.edge_links[:, 1]] = 1
        A[b.edge_links[:, 1], b.edge_links[:, 0]] = 1
        A += np.eye(12)
        d = A.sum(1) ** -0.5
        np.testing.assert_allclose(a_hat[gi].numpy(), d[:, None] * A * d[None, :], rtol=1e-6)
    conv = pol.AntiSymmetricConvDense(5)
    x = torch.randn(3, 12, 5)
    y = conv(x, a_hat[:1])
    W = conv.W.detach()
    ref = x + 0.1 * torch.tanh(x @ (W - W.t() - 0.1 * torch.eye(5)).t() + a_hat[:1] @ conv.phi(x) + conv.bias)
    torch.testing.assert_close(y, ref)
    # the weight matrix acting on x is anti-symmetric up to the -gamma*I damping (the layer's point)
    M = W - W.t()
    torch.testing.assert_close(M, -M.t())


def test_gnn_q_policy_greedy_respects_mask():
    B, N, P = 16, 12, 2
    obs = _fake_obs(B, N, P, seed=3)
    boards = sy.sample_board_pool(1, N, 18, seed=1)
    a_hat = pol.normalized_adjacency(torch.from_numpy(sy.pack_pool(boards).ell.view(np.int32).copy()), N)
    net = pol.GnnQPolicy(P + 1, with_belief=True)
    x = net.features(obs, N)
    assert x.shape == (B, N, P + 2) and x[..., : P + 1].sum() == B * (P + 1)
    a, _, _ = net.act_greedy(obs, a_hat)
    legal = torch.gather(obs["action_mask"], -1, a.clamp_min(0).long().unsqueeze(-1)).squeeze(-1)
    assert bool((legal | (a < 0)).all())


def test_device_policy_and_sampler_refuse_to_run_without_a_gpu():
    """No CPU fallback: the HIP-backed policy pieces fail loudly on CPU tensors."""
    import pytest
    from student_mechanism_design_amd import collector as col
    net = pol.MappoPolicy(12, 2, hidden_size=8)
    with pytest.raises(sy.EngineError):
        pol.DeviceMappoPolicy(net)
    with pytest.raises(sy.EngineError):
        col.DeviceMaskedSampler(torch.device("cpu"))
