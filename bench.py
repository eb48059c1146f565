#!/usr/bin/env python3
"""Headline benchmark: agent-steps/s of the batched Scotland-Yard engine (BASELINE.json metric).

Workload = BASELINE.json configs[1]: 200-node board, 4 police (+MrX), 4096 parallel envs per GPU,
uniform-random policy, env kernels only.  One bench "step" = ONE fused rollout launch: T env-steps
of all B envs with the trajectory (obs, masks, belief, actions, rewards, flags) recorded to HBM.
E=400 edges and money=20 are SURVEY.md section 8 choices (BASELINE gives neither).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Episodes are independent: ranks own disjoint env shards (weak scaling, B per GPU fixed), there is
no data-path collective; the only cross-rank ops are the timing barrier and a MAX reduction.
Rank 0 prints ONE JSON line with `roofline` (HBM) and `cpu_baseline` (the CPU oracle, a port).
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


SETTLE_LAUNCHES = 12


def cpu_baseline(args, boards, weights, A):
    """The CPU oracle (plain-C port of the reference algorithm, OpenMP over envs) on a bounded sample
    of the same workload: the same boards / sizes, T_cpu fused steps of B envs."""
    from oracle import oracle_lib as ol
    import student_mechanism_design_amd as sy
    cores = usable_cores()
    graphs = [ol.OracleGraph(args.nodes, b.edge_links, b.edges.astype(np.int32)) for b in boards]
    B = args.envs
    per = -(-B // len(graphs))
    eg = np.minimum(np.arange(B) // per, len(graphs) - 1).astype(np.int32)
    orc = ol.OracleBatch(graphs, eg, B, args.police, args.money, node_stride=(args.nodes + 15) // 16 * 16,
                         weights=weights, tables=sy.reward_tables(), reveal_interval=args.reveal,
                         threads=cores)
    orc.reset(seed=1)
    orc.rollout(2, record=False)  # warm
    t_cpu, steps = 0.0, 0
    T = 8
    t0 = time.perf_counter()
    while t_cpu < args.cpu_seconds:
        orc.rollout(T, record=True)
        steps += T
        t_cpu = time.perf_counter() - t0
    rate = steps * B * A / t_cpu
    return {"value": rate, "unit": "agent-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/sy_oracle.c batched engine, B={B} envs x {steps} fused steps, trajectory recorded, "
                      f"{t_cpu:.1f}s wall, OpenMP {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--edges", type=int, default=400)
    ap.add_argument("--police", type=int, default=4)
    ap.add_argument("--money", type=int, default=20)
    ap.add_argument("--graphs", type=int, default=8, help="boards in the pool")
    ap.add_argument("--fused", type=int, default=256, help="env steps per launch (T); 256 ~ the reference's 250-step episode cap")
    ap.add_argument("--reveal", type=int, default=5)
    ap.add_argument("--wpb", type=int, default=0, help="waves (envs) per launch block, 0 = engine default")
    ap.add_argument("--no-record", action="store_true", help="do not write the trajectory (diagnostic only)")
    ap.add_argument("--no-belief", action="store_true", help="diagnostic: engine without the belief filter")
    ap.add_argument("--no-mask-record", action="store_true", help="diagnostic: do not record masks")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--step-api", action="store_true", help="also time the per-step sy_env_step launch path")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import student_mechanism_design_amd as sy

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
    elif args.gpus > 1:
        print("launch with torch.distributed.run for --gpus > 1", file=sys.stderr)
        sys.exit(2)
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)

    N, P, A, B, T = args.nodes, args.police, args.police + 1, args.envs, args.fused
    boards = sy.sample_board_pool(args.graphs, N, args.edges, seed=0)   # same synthetic boards on all ranks
    weights = np.full(11, 0.5)
    env = sy.BatchedScotlandYardEnv(B, boards, P, args.money, weights, seed=1234, reveal_interval=args.reveal,
                                    env_id_offset=rank * B, waves_per_block=args.wpb, device=device,
                                    with_belief=not args.no_belief)
    out = None if args.no_record else env.alloc_rollout(T, record_mask=not args.no_mask_record)

    def one_step():
        env.rollout(T, out=out, record=not args.no_record)

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # Device clocks and power state take about ten launches (~8 ms of load) to settle after an idle start
    # (tools: per-launch times 0.60, 0.60, 0.63, 0.68, 0.68, 0.67, 0.65, 0.64, 0.61, 0.60, 0.59 ms): part of set-up,
    # like allocating the buffers; the W warm-up steps asked for on the command line follow.
    for _ in range(SETTLE_LAUNCHES):
        one_step()
    for _ in range(args.warmup):
        one_step()
    sync_all()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()   # the engine launches on torch's current stream, so these bracket the kernel
        one_step()
        ev[i][1].record()
    sync_all()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    env_steps = args.steps * T * B * world
    value = env_steps * A / elapsed
    R, W = algorithmic_bytes_per_env_step(N, P)
    bytes_per_launch = (R + W) * T * B
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # measured by tools/pmc_traffic.py under rocprofv3
    if os.path.exists(pmc):
        try:
            with open(pmc) as f:
                rec = json.load(f)
            full = not (args.no_record or args.no_belief or args.no_mask_record)   # the counters were taken on the full workload
            if full and rec.get("config") == {"nodes": N, "police": P, "envs": B, "fused": T}:
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "agent-steps/sec, 200-node graph, 4 police, 4096 parallel envs, 1/2/4/8 GPUs",
        "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32 state / u8 masks / f32 belief / f64 reward", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 200-node random board, 4 police + MrX, 4096 envs/GPU, "
                               "uniform-random policy in-kernel, env kernels only",
                   "nodes": N, "edges": boards[0].num_edges, "police": P, "agent_money": args.money,
                   "envs_per_gpu": B, "graphs_in_pool": args.graphs, "fused_env_steps_per_launch": T,
                   "reveal_interval": args.reveal, "trajectory_recorded": not args.no_record,
                   "waves_per_block": env.waves_per_block, "lds_bytes_per_block": env.lds_bytes,
                   "settle_launches_before_warmup": SETTLE_LAUNCHES,
                   "parallelism": f"env-shard x{world} (no data-path collective)"},
        "env_steps_per_s": value / A,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "sy::rollout2_kernel<4,true,4,false> (fused rollout: paired move waves + belief waves)", "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_env_step": R + W, "algorithmic_read_bytes_per_env_step": R,
                     "read_only_frac": (R * T * B / (kern_ms * 1e-3) / 1e9) / HBM_PEAK_GBS},
    }
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
    if rank == 0:
        if not args.no_cpu and world == 1:
            result["cpu_baseline"] = cpu_baseline(args, boards, weights, A)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
