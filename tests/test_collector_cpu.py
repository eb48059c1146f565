"""CPU tests of the collector's torch-side maths and of the one multi-GPU exchange (gloo, world 2)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from student_mechanism_design_amd import collector as col

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _select_action_goldens():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "select_action_reference.json")) as f:
        return json.load(f)["cases"]


def test_masked_sampling_matches_the_reference_goldens():
    """tests/golden/select_action_reference.json: outputs of the UNMODIFIED MappoAgent.select_action
    (oracle/capture_select_action.py).  The batched restatement must produce the same normalised
    probabilities, and the reference's log-prob of the action it drew."""
    for c in _select_action_goldens():
        probs, mask = torch.tensor(c["probs"]), torch.tensor(c["mask"], dtype=torch.float32)
        _, _, p = col.masked_categorical_sample(probs.unsqueeze(0), mask.unsqueeze(0))
        np.testing.assert_allclose(p[0].numpy(), np.array(c["current_probs"], dtype=np.float32), rtol=1e-6, atol=1e-9,
                                   err_msg=c["kind"])
        norm = p[0] / p[0].sum()
        np.testing.assert_allclose(float(torch.log(norm[c["action"]])), c["log_prob"], rtol=1e-5, atol=1e-6)
        assert c["kind"] == "empty_mask" or c["mask"][c["action"]] == 1


def test_masked_sampling_follows_mappo_select_action():
    """agent/mappo_agent.py:87-142: illegal actions get zero probability; all-zero product -> uniform
    over the mask; empty mask -> uniform over everything; result renormalised."""
    g = torch.Generator().manual_seed(0)
    probs = torch.softmax(torch.randn(6, 9, generator=g), -1)
    mask = torch.zeros(6, 9)
    mask[0, [1, 4]] = 1
    mask[1, :] = 1
    mask[2, 3] = 1
    mask[3, [0, 8]] = 1
    probs[3] = 0
    probs[3, 5] = 1.0          # all mass on an illegal action -> uniform over the mask
    # row 4: empty mask -> uniform over all 9; row 5: legal subset
    mask[5, [2, 6, 7]] = 1
    a, logp, p = col.masked_categorical_sample(probs, mask, generator=g)
    assert (p[0, [0, 2, 3, 5, 6, 7, 8]] == 0).all() and a[0].item() in (1, 4)
    exp0 = probs[0, [1, 4]] / (probs[0, [1, 4]].sum() + 1e-8)
    torch.testing.assert_close(p[0, [1, 4]], exp0)
    assert a[2].item() == 3
    torch.testing.assert_close(p[3, [0, 8]], torch.tensor([0.5, 0.5]))
    torch.testing.assert_close(p[4], torch.full((9,), 1 / 9))
    assert a[5].item() in (2, 6, 7)
    torch.testing.assert_close(logp.exp(), (p / p.sum(-1, keepdim=True)).gather(-1, a.unsqueeze(-1)).squeeze(-1))
    # frequencies follow the renormalised probabilities
    big = probs[1:2].expand(20000, -1)
    aa, _, _ = col.masked_categorical_sample(big, torch.ones(20000, 9), generator=g)
    freq = torch.bincount(aa, minlength=9).float() / 20000
    assert (freq - probs[1]).abs().max() < 0.02


def _ref_returns(rewards, dones, gamma):
    """mappo_agent.py:248-254 on a flat buffer."""
    out = np.zeros_like(rewards)
    run = 0.0
    for i in reversed(range(len(rewards))):
        run = rewards[i] + gamma * run * (1 - dones[i])
        out[i] = run
    return out


def test_returns_match_ppo_update_loop_and_gae_reduces_to_them():
    rng = np.random.default_rng(0)
    T, B = 37, 5
    r = rng.normal(size=(T, B))
    d = (rng.random((T, B)) < 0.15).astype(np.float64)
    R = col.discounted_returns(torch.tensor(r), torch.tensor(d), 0.99).numpy()
    for b in range(B):
        np.testing.assert_allclose(R[:, b], _ref_returns(r[:, b], d[:, b], 0.99), rtol=1e-12)
    v = torch.tensor(rng.normal(size=(T, B)))
    adv, ret = col.gae(torch.tensor(r), v, torch.tensor(d), torch.zeros(B, dtype=torch.float64), 0.99, 1.0)
    np.testing.assert_allclose(ret.numpy(), R, rtol=1e-10, atol=1e-12)      # lambda = 1, zero bootstrap
    np.testing.assert_allclose(adv.numpy(), R - v.numpy(), rtol=1e-10, atol=1e-12)
    sa = col.standardized_advantages(torch.tensor(R), v)
    ref = (R - v.numpy())
    ref = (ref - ref.mean()) / (ref.std(ddof=1) + 1e-8)                    # torch.std is unbiased, :258
    np.testing.assert_allclose(sa.numpy(), ref, rtol=1e-10)
    # per-agent rewards with per-env done flags broadcast over the agent axis
    r3 = torch.tensor(rng.normal(size=(T, B, 3)))
    R3 = col.discounted_returns(r3, torch.tensor(d), 0.9)
    np.testing.assert_allclose(R3[:, 2, 1].numpy(), _ref_returns(r3[:, 2, 1].numpy(), d[:, 2], 0.9), rtol=1e-12)


def _make_record(rank, T=4, B=3, A=2, NS=16):
    g = torch.Generator().manual_seed(100 + rank)
    return {
        "pos": torch.randint(0, 50, (T, B, A), generator=g, dtype=torch.int32),
        "reward": torch.randn(T, B, A, generator=g, dtype=torch.float64),
        "mask": torch.randint(0, 2, (T, B, A, NS), generator=g, dtype=torch.uint8),
        "belief": torch.rand(T, B, NS, generator=g),
        "terminated": torch.randint(0, 2, (T, B), generator=g, dtype=torch.uint8),
        "winner": torch.randint(0, 3, (T, B), generator=g, dtype=torch.int8),
        "skipped": None,
    }


def test_pack_unpack_roundtrip():
    rec = _make_record(0)
    buf, meta = col.pack_record(rec)
    assert buf.dtype == torch.uint8
    out = col.unpack_record(buf, meta, 1)
    for k, v in rec.items():
        if v is not None:
            assert torch.equal(out[k], v), k
    assert "skipped" not in out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec = _make_record(rank)
        out = col.gather_trajectories(rec)
        ok = True
        for r in range(world):
            ref = _make_record(r)
            for k, v in ref.items():
                if v is None:
                    continue
                B = v.shape[1]
                ok = ok and torch.equal(out[k][:, r * B:(r + 1) * B], v)
        ok = ok and out["pos"].shape[1] == world * 3
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _arena_record(rank, T=4, B=3, A=3, NS=16):
    from student_mechanism_design_amd.env import make_rollout_record, record_words
    rec = make_rollout_record(T, B, A, NS, record_words(A), "cpu", log_prob=True, value=True)
    g = torch.Generator().manual_seed(200 + rank)
    rec.arena.copy_(torch.randint(0, 256, (rec.arena.numel(),), generator=g, dtype=torch.uint8))
    return rec


def _worker_arena(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec = _arena_record(rank)
        ex = col.TrajectoryExchange(rec)
        ok = ex.world == world and ex.bytes_received_per_rank == (world - 1) * rec.arena.numel()
        for rep in range(2):                       # the receive buffer is reused across updates
            out = ex.gather()
            recv_ptr = ex._recv.data_ptr()
            for r in range(world):
                ref = _arena_record(r)
                for k, v in ref.items():
                    if v is None:
                        continue
                    got = out[k][r]
                    ok = ok and got.shape == v.shape and got.dtype == v.dtype
                    ok = ok and torch.equal(got.contiguous().view(torch.uint8), v.contiguous().view(torch.uint8))
                    # views of the receive buffer, not copies
                    ok = ok and recv_ptr <= out[k].data_ptr() < recv_ptr + ex._recv.numel()
            ok = ok and out["pos"].shape[:2] == (world, 4)
        # the data-parallel alternative: the fused update's gradient slab averaged across the ranks, in place, one collective
        slab = torch.full((4, 10), float(rank + 1))
        back = col.allreduce_slab(slab)
        ok = ok and back is slab and bool((slab == (1 + world) / 2).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_zero_copy_trajectory_exchange_gloo_world2():
    """The exchange of BASELINE configs[3] ("gather at PPO update") on arena-backed records: the arena is the
    send buffer, one all-gather into a reused [world, bytes] buffer, results are views of it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_arena, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_arena_record_layout_and_single_rank_gather():
    from student_mechanism_design_amd.env import make_rollout_record, record_words
    T, B, A, NS = 5, 4, 5, 32
    rec = make_rollout_record(T, B, A, NS, record_words(A), "cpu", log_prob=True)
    lo, hi = rec.arena.data_ptr(), rec.arena.data_ptr() + rec.arena.numel()
    for k, v in rec.items():
        if v is not None:
            assert lo <= v.data_ptr() < hi, k                        # every tensor is a view of the arena
    for name, dtype, shape, off, nbytes in rec.layout:
        assert off % 256 == 0 and rec[name].is_contiguous()
    rec["pos"][2, 1, 3] = 77
    assert int(rec["record"][2, 1, 2 * A + 3]) == 77                 # named fields alias the packed row
    rec["reward"][1, 0, 4] = -0.25
    assert rec["record"][1, 0, 8:10].view(torch.float64).item() == -0.25
    out = col.gather_trajectories(rec)                                # world 1: views of the arena itself
    assert out["pos"].shape == (1, T, B, A) and int(out["pos"][0, 2, 1, 3]) == 77
    assert out["mask"].data_ptr() == rec["mask"].data_ptr()


def test_allreduce_gradients_is_identity_without_a_group():
    slab = torch.arange(6.0).view(2, 3)
    assert col.allreduce_slab(slab) is slab and torch.equal(slab, torch.arange(6.0).view(2, 3))
    lin = torch.nn.Linear(3, 2)
    lin(torch.ones(1, 3)).sum().backward()
    g = lin.weight.grad.clone()
    assert col.allreduce_gradients(lin) == 0 and torch.equal(lin.weight.grad, g)


def test_gather_trajectories_gloo_world2():
    """The N>1 path: env shards are independent, one all-gather of the packed record at the update."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_env_shards_use_disjoint_rng_streams():
    """Sharding rule: rank r runs envs [r*B, (r+1)*B) of the global batch.  With the oracle engine:
    two shards of 8 envs (env_id_offset 0 and 8) == one batch of 16 envs."""
    from oracle import oracle_lib as ol
    import student_mechanism_design_amd as sy
    boards = sy.sample_board_pool(1, 20, 30, seed=2)
    g = [ol.OracleGraph(20, boards[0].edge_links, boards[0].edges.astype(np.int32))]
    w = np.linspace(0.1, 0.9, 11)
    whole = ol.OracleBatch(g, np.zeros(16, np.int32), 16, 3, 8, weights=w, node_stride=32)
    whole.reset(seed=5)
    rw = whole.rollout(30)
    for r in range(2):
        part = ol.OracleBatch(g, np.zeros(8, np.int32), 8, 3, 8, weights=w, node_stride=32, env_id_offset=8 * r)
        part.reset(seed=5)
        rp = part.rollout(30)
        for k in ("pos", "action", "reward", "terminated"):
            np.testing.assert_array_equal(rp[k], rw[k][:, 8 * r:8 * (r + 1)], err_msg=k)


def test_returns_and_advantages_match_the_unmodified_ppo_update():
    """tests/golden/ppo_returns_reference.npz: what mappo_agent.py:247-258 computes inside the unmodified
    `MappoAgent.ppo_update` (oracle/capture_ppo_returns.py).  The oracle's C restatement is bit-exact; the torch
    forms (`discounted_returns`, `standardized_advantages`) agree to float32 rounding of a different op order."""
    from oracle import oracle_lib as ol
    z = np.load(os.path.join(GOLDEN, "ppo_returns_reference.npz"))
    assert len(z["case_names"]) >= 7
    for name in z["case_names"]:
        r32 = z[f"{name}/reward"].astype(np.float32)
        d, gamma, val = z[f"{name}/done"], float(z[f"{name}/gamma"]), z[f"{name}/values"]
        ret, adv = ol.discounted_returns_f32(r32, d, gamma, val)
        np.testing.assert_array_equal(ret, z[f"{name}/returns"], err_msg=name)          # bit-exact float32
        t_ret = col.discounted_returns(torch.tensor(r32), torch.tensor(d), gamma)
        np.testing.assert_allclose(t_ret.numpy(), z[f"{name}/returns"], rtol=2e-6, atol=2e-6, err_msg=name)
        t_adv = col.standardized_advantages(torch.tensor(z[f"{name}/returns"]), torch.tensor(val))
        np.testing.assert_allclose(t_adv.numpy(), z[f"{name}/advantages"], rtol=1e-6, atol=1e-6, err_msg=name)
        # GAE at lambda = 1 with a zero bootstrap reduces to the reference's returns
        g_adv, g_ret = ol.gae_f64(z[f"{name}/reward"], d, val.astype(np.float64), gamma, 1.0)
        run, want = 0.0, np.zeros(len(r32))
        for i in range(len(r32) - 1, -1, -1):
            run = z[f"{name}/reward"][i] + gamma * run * (1.0 - d[i])
            want[i] = run
        np.testing.assert_allclose(g_ret, want, rtol=1e-12, atol=1e-12, err_msg=name)
