#!/usr/bin/env python3
"""Headline benchmark: agent-steps/s of the batched Scotland-Yard engine (BASELINE.json metric).

Workload = BASELINE.json configs[1]: 200-node board, 4 police (+MrX), 4096 parallel envs per GPU,
uniform-random policy, env kernels only.  One bench "step" = ONE fused rollout launch: T env-steps
of all B envs with the trajectory (obs, masks, belief, actions, rewards, flags) recorded to HBM.
E=400 edges and money=20 are SURVEY.md section 8 choices (BASELINE gives neither).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Timing: W untimed warm-up steps, then R >= 5 repeats of EXACTLY K steps, each repeat bracketed by a barrier +
`torch.cuda.synchronize()` on both sides and reduced with MAX over ranks; the line reports the MEDIAN repeat (min /
max beside it).  R is chosen so that the timed regions add up to >= ~1.6 s (--timed-seconds).  No launches precede the warm-up: the
clock / power transient of an idle device (its first ~8 ms under load) falls into the first repeat and the median
ignores it.  Kernel time = HIP events on the launch stream around every timed launch (median).

Episodes are independent: ranks own disjoint env shards (weak scaling, B per GPU fixed), no data-path collective in
the rollout.  The ONE exchange of the multi-GPU path — the trajectory all-gather at the PPO update, BASELINE
configs[3] — is timed separately for N > 1 (`gather`: zero-copy `collector.TrajectoryExchange`).
Rank 0 prints ONE JSON line with `roofline` (HBM), `cpu_baseline` (the CPU oracle, a port: all-core and one-thread)
and `verified` (the launch right after the timed region checked against the oracle).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_env_step(N, P):
    """reference observation layout, 1 byte/node masks, f32 belief."""
    A = P + 1
    R = 4 * A + 4 * A + 4 * P + 4 + 1 + 4 * N + 4 * P
    W = 4 * A + 4 * P + 4 + 2 * A + 4 * A + 4 * N + A * N + 4 * P
    return R, W


def usable_cores():
    """Host cores this process may really use: the cgroup CPU quota when one is set (a 1-GPU box gets a
    share of the host), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(np.ceil(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, int(np.ceil(q / int(f2.read())))))
            break
        except Exception:
            continue
    return n


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rank_environment(rank, world, port, base=None):
    """Environment of rank `rank` of a one-node job: what torch.distributed.run would set."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool
    return env


def spawn_ranks(n, argv, popen=None, timeout=None):
    """`python bench.py --gpus N` without a launcher: THIS process — which has not touched the GPU (no torch import, no
    HIP call; a process that has must never exec or fork GPU work) — starts the N ranks as child processes with the
    environment torch.distributed.run would give them, relays rank 0's single JSON line and returns the worst exit
    code.  If a rank fails, the others are stopped (by their own PIDs)."""
    import subprocess
    popen = popen or subprocess.Popen
    port = free_port()
    procs = []
    for r in range(n):
        procs.append(popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=rank_environment(r, n, port),
                           stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    t_end = None if timeout is None else time.time() + timeout
    worst, out0 = 0, b""
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            try:
                if r == 0:
                    o, _ = procs[0].communicate(timeout=0.2)
                    out0 += o or b""
                    rc = procs[0].returncode
                else:
                    rc = procs[r].wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            pending.discard(r)
            if rc != 0:
                worst = worst or rc
                for q in pending:          # a collective with a dead peer never returns: stop the others
                    procs[q].terminate()
        if t_end is not None and time.time() > t_end and pending:
            for q in pending:
                procs[q].kill()
            worst = worst or 124
    # ONE JSON line on stdout; whatever else rank 0 printed there (a backend's connection banner) goes to stderr
    for line in out0.decode("utf-8", "replace").splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return worst


def config3_record(args, boards, weights, device):
    """BASELINE configs[2] beside the headline (an extra key of the JSON line, measured AFTER it): the same 200-node /
    4-police / 4096-env workload with a LEARNED policy in the loop, on-device returns and a PPO update — one training
    iteration = collect T=64 steps (the MAPPO actors sampling inside the fused rollout) + `sy_returns_advantages` + one
    pass of minibatch PPO over the 262 144 env-steps (1.3 M agent transitions) + weight refresh.  Also the collect rate of
    the reference's other policy family, the GNN Q-policy (`sy_gnn_q_act` + `sy_env_step_record` per step, graph-replayed)."""
    import torch
    import student_mechanism_design_amd as sy
    from student_mechanism_design_amd import collector as col, policies as pol
    from student_mechanism_design_amd.update import MappoUpdater
    N, P, B, T = args.nodes, args.police, args.envs, 64
    A = P + 1
    env = sy.BatchedScotlandYardEnv(B, boards, P, args.money, weights, seed=args.seed + 1, reveal_interval=args.reveal, device=device)
    torch.manual_seed(0)
    net = pol.MappoPolicy(N, P, hidden_size=64).to(device)
    fused = pol.DeviceMappoPolicy(net, seed=3)
    env.set_policy(fused)
    out = env.alloc_rollout(T)
    up = MappoUpdater(net, env.ell, env.env_graph, minibatch=32768, use_graph=args.config3_graph)

    def timed(fn):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(device)
        return r, (time.perf_counter() - t0) * 1e3

    def iteration(split):
        ms = {}
        rec, ms["collect"] = timed(lambda: env.rollout(T, out=out)) if split else (env.rollout(T, out=out), 0.0)
        f = lambda: col.device_returns(rec["reward"], rec["terminated"], 0.99, done_b=rec["truncated"])   # noqa: E731
        (ret, _), ms["returns"] = timed(f) if split else (f(), 0.0)

        def upd():
            up.update(rec, ret)
            fused.refresh()
        _, ms["update"] = timed(upd) if split else (upd(), 0.0)
        return ms

    for _ in range(3):
        iteration(False)                      # warm-up (allocations, the update's graph capture)
    iters = 5
    parts = [iteration(True) for _ in range(iters)]
    _, whole_ms = timed(lambda: [iteration(False) for _ in range(iters)])
    env.check_status()
    al, cl = (float(x) for x in up.last_losses)
    med = lambda k: float(np.median([p_[k] for p_ in parts]))   # noqa: E731
    n = T * B * A
    rec3 = {"what": "BASELINE configs[2]: learned policy (MAPPO actors, hidden 64) sampling inside the fused rollout, T=64 x "
                    "%d envs, returns in one HIP launch, one minibatch-PPO pass (minibatch 32768 env-steps: %s%s), weights "
                    "refreshed; medians of %d iterations" % (B, "loss + gradient in one HIP kernel on resident parameters, Adam in its reduction launch: sy_mappo_ppo_grad"
                                                             if up.fused else "torch autograd", ", each step one HIP graph" if up.use_graph else "", iters),
            "kernel": env.rollout_kernel_name(), "update_path": "sy_mappo_ppo_grad" if up.fused else "torch", "agent_transitions_per_iteration": n,
            "collect_ms": med("collect"), "returns_ms": med("returns"), "update_ms": med("update"),
            "collect_agent_steps_per_s": n / (med("collect") * 1e-3),
            "iteration_ms_unsplit": whole_ms / iters, "iteration_agent_steps_per_s": n * iters / (whole_ms * 1e-3),
            "update_fraction_of_iteration": med("update") / (med("collect") + med("returns") + med("update")),
            "last_actor_loss": al, "last_critic_loss": cl, "losses_finite": bool(np.isfinite(al) and np.isfinite(cl))}
    # the reference's literal form: ONE step over the whole buffer (mappo_agent.py:260-293 has no minibatches)
    up_full = MappoUpdater(net, env.ell, env.env_graph, minibatch=T * B, use_graph=False)
    rec = env.rollout(T, out=out)
    ret, _ = col.device_returns(rec["reward"], rec["terminated"], 0.99, done_b=rec["truncated"])
    for _ in range(2):
        up_full.update(rec, ret)
    rec3["update_ms_one_full_batch_step"] = float(np.median([timed(lambda: up_full.update(rec, ret))[1] for _ in range(iters)]))
    env.close()
    # the GNN Q-policy driving the per-step collector (configs[2] names the GNN policy)
    env = sy.BatchedScotlandYardEnv(B, boards, P, args.money, weights, seed=args.seed + 2, reveal_interval=args.reveal, device=device)
    gnn = pol.GnnQPolicy(A).to(device)
    dgnn = pol.DeviceGnnPolicy(gnn, pol.GcnTables(env.pool.boards, device=device), env.env_graph, seed=4, explore_eps=0.1)
    c = col.RolloutCollector(env, dgnn.act, frames_per_batch=T, use_graph=True)
    for _ in range(3):
        c.collect()
    _, gms = timed(lambda: [c.collect() for _ in range(iters)])
    env.check_status()
    rec3["gnn_collect"] = {"what": "GNN Q-policy (2 x AntiSymmetricConv + Linear, epsilon-greedy 0.1) as one HIP kernel per step + "
                                   "sy_env_step_record, T=64 steps replayed as one HIP graph",
                           "collect_ms": gms / iters, "collect_agent_steps_per_s": n * iters / (gms * 1e-3)}
    env.close()
    return rec3


def make_oracle(args, boards, weights, env_graph, threads, env_id_offset=0):
    from oracle import oracle_lib as ol
    import student_mechanism_design_amd as sy
    graphs = [ol.OracleGraph(args.nodes, b.edge_links, b.edges.astype(np.int32)) for b in boards]
    return ol.OracleBatch(graphs, env_graph, args.envs, args.police, args.money, node_stride=(args.nodes + 15) // 16 * 16,
                          weights=weights, tables=sy.reward_tables(), reveal_interval=args.reveal, threads=threads,
                          env_id_offset=env_id_offset, with_belief=not args.no_belief)


def cpu_baseline(args, boards, weights, env_graph, A):
    """The CPU oracle (plain-C port of the reference algorithm, OpenMP over envs) on a bounded sample of the same
    workload — the same boards / sizes, T_cpu fused steps of B envs — once on all usable cores (`value`) and once
    on ONE thread (`single_thread_value`).  A reported baseline, not a target."""
    cores = usable_cores()
    B = args.envs
    out = {}
    for label, threads, budget in (("all", cores, args.cpu_seconds * 0.6), ("one", 1, args.cpu_seconds * 0.4)):
        orc = make_oracle(args, boards, weights, env_graph, threads)
        orc.reset(seed=1)
        orc.rollout(2, record=False)  # warm
        t_cpu, steps, T = 0.0, 0, 8 if threads > 1 else 2
        t0 = time.perf_counter()
        while t_cpu < budget:
            orc.rollout(T, record=True)
            steps += T
            t_cpu = time.perf_counter() - t0
        out[label] = (steps * B * A / t_cpu, steps, t_cpu)
    return {"value": out["all"][0], "unit": "agent-steps/s", "cores": cores, "kind": "port",
            "single_thread_value": out["one"][0],
            "sample": f"oracle/sy_oracle.c batched engine, trajectory recorded: B={B} envs x {out['all'][1]} fused steps in "
                      f"{out['all'][2]:.1f}s on {cores} OpenMP threads; x {out['one'][1]} steps in {out['one'][2]:.1f}s on 1 thread"}


def verify_last_launch(args, boards, weights, env, snap, out, T):
    """Outside the timed region: the oracle restarts from the state snapshot taken just before the LAST timed launch
    and must reproduce that launch's recorded trajectory (bit-exact; belief within 1e-5) and the live state."""
    orc = make_oracle(args, boards, weights, env.env_graph_host, usable_cores(), env_id_offset=env.env_id_offset)
    orc.reset(seed=env.stream_key)
    N = env.N
    orc.pos[:] = snap["pos"].cpu().numpy()
    orc.money[:] = snap["budget"].cpu().numpy()
    orc.t[:] = snap["t"].cpu().numpy()
    orc.step_count[:] = snap["step_count"].cpu().numpy().astype(np.uint32)
    orc.visits[:] = snap["visits"].cpu().numpy().astype(np.int32)
    orc.mask[:] = snap["mask"].cpu().numpy()
    if orc.belief is not None:
        orc.belief[:] = snap["belief"].cpu().numpy().astype(np.float64)
    ref = orc.rollout(T, record_mask=out.get("mask") is not None)
    ok, worst = True, 0.0
    for s0 in range(0, T, 32):
        s1 = min(T, s0 + 32)
        for k, rk in (("pos", "pos"), ("budget", "money"), ("t", "t"), ("action", "action"), ("terminated", "terminated"),
                      ("truncated", "truncated"), ("winner", "winner"), ("reward", "reward"), ("mask", "mask")):
            if out.get(k) is None or ref.get(rk) is None:
                continue
            ok = ok and np.array_equal(out[k][s0:s1].cpu().numpy(), ref[rk][s0:s1])
        if out.get("belief") is not None and ref.get("belief") is not None:
            worst = max(worst, float(np.abs(out["belief"][s0:s1].cpu().numpy().astype(np.float64) - ref["belief"][s0:s1]).max()))
    ok = ok and worst <= 1e-5
    ok = ok and np.array_equal(env.pos.cpu().numpy(), orc.pos) and np.array_equal(env._mask.cpu().numpy(), orc.mask)
    ok = ok and np.array_equal(env.reward.cpu().numpy(), orc.reward) and np.array_equal(env.t.cpu().numpy(), orc.t)
    return bool(ok), worst, int((ref["terminated"] | ref["truncated"]).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=0, help="timed repeats of K steps (0 = at least 5, enough for --timed-seconds)")
    ap.add_argument("--timed-seconds", type=float, default=1.6, help="target for the sum of the timed regions when --repeats is 0")
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--edges", type=int, default=400)
    ap.add_argument("--police", type=int, default=4)
    ap.add_argument("--money", type=int, default=20)
    ap.add_argument("--graphs", type=int, default=8, help="boards in the pool (SURVEY 8d variants: 1 and 64)")
    ap.add_argument("--fused", type=int, default=256, help="env steps per launch (T); 256 ~ the reference's 250-step episode cap")
    ap.add_argument("--reveal", type=int, default=5)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--wpb", type=int, default=0, help="waves (envs) per launch block, 0 = engine default")
    ap.add_argument("--no-record", action="store_true", help="do not write the trajectory (diagnostic only)")
    ap.add_argument("--no-belief", action="store_true", help="diagnostic: engine without the belief filter")
    ap.add_argument("--no-mask-record", action="store_true", help="diagnostic: do not record masks")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the last timed trajectory")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the separately timed trajectory all-gather")
    ap.add_argument("--step-api", action="store_true", help="also time the per-step sy_env_step launch path")
    ap.add_argument("--no-config3", action="store_true", help="skip the configs[2] sub-record (learned policy + update)")
    ap.add_argument("--no-belief-layout", action="store_true", help="diagnostic: belief scratch in node order (no bank-aware layout)")
    ap.add_argument("--config3-graph", action="store_true", help="configs[2] sub-record: replay every minibatch step as one HIP graph "
                    "(no gain on the fused update: a minibatch is two launches)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start the ranks ourselves, BEFORE anything touches the GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    import student_mechanism_design_amd as sy
    from student_mechanism_design_amd.collector import TrajectoryExchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SY_BENCH_BACKEND", "nccl")   # "gloo" only to rehearse the N>1 path on a 1-GPU box
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev if world > 1 else 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)

    N, P, A, B, T, K = args.nodes, args.police, args.police + 1, args.envs, args.fused, args.steps
    boards = sy.sample_board_pool(args.graphs, N, args.edges, seed=0)   # same synthetic boards on all ranks
    weights = np.full(11, 0.5)
    env = sy.BatchedScotlandYardEnv(B, boards, P, args.money, weights, seed=args.seed, reveal_interval=args.reveal,
                                    env_id_offset=rank * B, waves_per_block=args.wpb, device=device,
                                    with_belief=not args.no_belief, belief_layout=not args.no_belief_layout)
    out = None if args.no_record else env.alloc_rollout(T, record_mask=not args.no_mask_record)
    full = not (args.no_record or args.no_belief or args.no_mask_record)
    do_verify = full and not args.no_verify and rank == 0

    def one_step():
        env.rollout(T, out=out, record=not args.no_record)

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    sync_all()
    tw = time.perf_counter()
    for _ in range(args.warmup):
        one_step()
    sync_all()
    est_step = max((time.perf_counter() - tw) / max(args.warmup, 1), 1e-5)
    repeats = args.repeats        # 0: decided after the first timed repeat (see below)

    state_names = ("pos", "budget", "t", "step_count", "_visits", "_belief", "_mask")
    elapsed, events = [], []
    r = 0
    while True:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        sync_all()
        t0 = time.perf_counter()
        for i in range(K):
            ev[i][0].record()   # the engine launches on torch's current stream, so these bracket the kernel
            one_step()
            ev[i][1].record()
        sync_all()
        elapsed.append(time.perf_counter() - t0)
        events.append(ev)
        r += 1
        if repeats <= 0:
            # enough repeats for >= --timed-seconds of timed launches (a device sampler with a period of seconds then
            # sees the load), at least 5, decided from the first timed repeat; every rank must run the same number
            repeats = int(min(2000, max(5, np.ceil(args.timed_seconds / max(elapsed[0], 1e-5)))))
            if world > 1:
                rr = torch.tensor([repeats], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
                dist.all_reduce(rr, op=dist.ReduceOp.MAX)
                repeats = int(rr.item())
        if r >= repeats:
            break
    # the launch the oracle replays: one more step of the same stream of launches, right after the timed region
    # (untimed, so that the state snapshot — 7 device copies, ~10 MB — is in nobody's measurement)
    snap = None
    if do_verify:
        snap = {n.lstrip("_"): getattr(env, n).clone() for n in state_names if getattr(env, n) is not None}
        one_step()
        torch.cuda.synchronize(device)     # (rank 0 only: no barrier here)
    env.check_status()          # a launch that lost a hand-off reports it here instead of returning wrong data
    el = torch.tensor(elapsed, dtype=torch.float64, device=device if (world > 1 and backend == "nccl") else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = np.sort(el.cpu().numpy())
    med = float(np.median(el))
    kern = np.array([a.elapsed_time(b) for ev in events for a, b in ev])
    kern_ms = float(np.median(kern))

    env_steps = K * T * B * world
    value = env_steps * A / med
    R, W = algorithmic_bytes_per_env_step(N, P)
    bytes_per_launch = (R + W) * T * B
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # measured by tools/pmc_traffic.py under rocprofv3
    build_id = sy._lib.build_id()
    if os.path.exists(pmc):
        try:
            with open(pmc) as f:
                rec = json.load(f)
            if full and rec.get("config") == {"nodes": N, "police": P, "envs": B, "fused": T}:
                if rec.get("build_id") == build_id:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = ("stored: profiles/pmc_traffic.json, collected from this library build (%s) in separate rocprofv3 "
                                      "--pmc FETCH_SIZE / WRITE_SIZE passes of this workload, gfx950 x2 fetch correction; not "
                                      "measured in this run" % build_id)
                else:      # counters of another kernel are not this kernel's traffic
                    traffic_source = ("none: profiles/pmc_traffic.json was collected from library build %s, this run loaded %s"
                                      % (rec.get("build_id"), build_id))
        except Exception:
            traffic = None

    result = {
        "metric": "agent-steps/sec, 200-node graph, 4 police, 4096 parallel envs, 1/2/4/8 GPUs",
        "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": med / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32 state / u8 masks / f32 belief / f64 reward", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 200-node random board, 4 police + MrX, 4096 envs/GPU, "
                               "uniform-random policy in-kernel, env kernels only",
                   "nodes": N, "edges": boards[0].num_edges, "police": P, "agent_money": args.money,
                   "envs_per_gpu": B, "graphs_in_pool": args.graphs, "fused_env_steps_per_launch": T,
                   "reveal_interval": args.reveal, "trajectory_recorded": not args.no_record,
                   "waves_per_block": env.waves_per_block, "lds_bytes_per_block": env.lds_bytes,
                   "parallelism": f"env-shard x{world} (no data-path collective in the rollout)"},
        "timing": {"repeats": repeats, "statistic": "median over repeats of K steps, MAX over ranks per repeat",
                   "ms_per_step_min": float(el[0]) / K * 1e3, "ms_per_step_max": float(el[-1]) / K * 1e3,
                   "timed_region_ms_total": float(el.sum()) * 1e3, "launches_before_warmup": 0},
        "env_steps_per_s": value / A,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "hbm_frac_measured": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "kernel": env.rollout_kernel_name(record=not args.no_record), "library_build_id": build_id,
                     "kernel_ms": kern_ms, "kernel_ms_min": float(kern.min()), "kernel_ms_max": float(kern.max()),
                     "algorithmic_bytes_per_env_step": R + W, "algorithmic_read_bytes_per_env_step": R,
                     "read_only_frac": (R * T * B / (kern_ms * 1e-3) / 1e9) / HBM_PEAK_GBS,
                     "note": "frac = ALGORITHMIC bytes (SURVEY 8d) / kernel time / 8 TB/s; hbm_frac_measured = counter bytes "
                             "(what the fused kernel really moves: the trajectory writes) / kernel time / 8 TB/s"},
    }
    if do_verify and snap is not None:
        ok, worst, n_done = verify_last_launch(args, boards, weights, env, snap, out, T)
        result["verified"] = ok
        result["verification"] = {"what": "the launch that follows the last timed one (same stream of launches, untimed) replayed by the CPU oracle from a state snapshot: record "
                                          "(pos, budget, t, action, flags, winner, float64 reward, masks) and live state "
                                          "bit-exact, belief max abs diff", "belief_max_abs_diff": worst,
                                  "episodes_finished_in_launch": n_done}
    else:
        result["verified"] = None
    if world > 1 and not args.no_gather and out is not None:
        ex = TrajectoryExchange(out)
        ex.gather()                      # allocates the receive buffer, warms the communicator
        sync_all()
        gt = []
        for _ in range(5):
            sync_all()
            t0 = time.perf_counter()
            ex.gather()
            sync_all()
            gt.append(time.perf_counter() - t0)
        g = torch.tensor(gt, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(g, op=dist.ReduceOp.MAX)
        gms = float(np.median(g.cpu().numpy())) * 1e3
        result["gather"] = {"what": "the ONE exchange per PPO update: all_gather_into_tensor of each rank's rollout arena "
                                    "(zero-copy send, preallocated receive, results are views)",
                            "gather_ms": gms, "bytes_per_rank_sent": ex.nbytes, "bytes_per_rank_received": ex.bytes_received_per_rank,
                            "recv_GBps_per_rank": ex.bytes_received_per_rank / (gms * 1e-3) / 1e9,
                            "rollout_plus_gather_agent_steps_per_s": T * B * world * A / (med / K + gms * 1e-3),
                            "alternative": "gradient all-reduce of the MAPPO networks: 0.77 MB per update"}
    if args.step_api and rank == 0:
        act = env.rollout(1)["action"][0].contiguous()      # some legal actions, as the caller's own tensor
        for _ in range(20):
            env.step(act)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        n = 200
        for _ in range(n):
            env.step(act)
        torch.cuda.synchronize(device)
        result["step_api_agent_steps_per_s"] = n * B * A / (time.perf_counter() - t1)
        del act
    if rank == 0 and world == 1 and not args.no_config3 and full:
        try:
            result["config3"] = config3_record(args, boards, weights, device)
        except Exception as e:      # the sub-record must never take the headline down with it
            result["config3"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        if not args.no_cpu and world == 1:
            result["cpu_baseline"] = cpu_baseline(args, boards, weights, env.env_graph_host, A)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if result.get("verified") is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
