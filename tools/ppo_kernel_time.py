"""Per-call time of sy_mappo_ppo_grad (32 768 rows of the configs[2] shape) with HIP events, and — with a diagnostic build of the
engine (tools/build_variants.sh NAME:"-DSY_PPO_DIAG_TIMES [...]", loaded through SY_ENGINE_LIB, its name must contain "ppot") —
the start / end stamps of every block grouped by role and the phase shares of wave 0 (how the grid shares
and the "LDS pipe, not latency" reading were obtained).  Other diagnostic flags of csrc/sy_ppo.hip: -DSY_PPO_DIAG_NO_ADD,
-DSY_PPO_DIAG_NOLOGIT, -DSY_PPO_DIAG_NOZ, -DSY_PPO_DIAG_ONLY=<role>, -DSY_PPO_DIAG_BUDGET=<blocks>, -DSY_PPO_RPG=2.
NOTE: builds that drop work (ONLY / NO_ADD ...) leave garbage or zeros in the gradient; a run that then takes an Adam step
turns the parameters into NaN and every later launch skips its backward pass — time such builds on the FIRST call only, or
on parameters that were not updated (this script times the gradient call without an optimiser state).
Run through gpurun:  SY_ENGINE_LIB=$PWD/tools/_diag/libsy_ppot256.so python tools/ppo_kernel_time.py"""
import os, sys, time, numpy as np, torch, ctypes as C
sys.path.insert(0, os.getcwd())
import student_mechanism_design_amd as sy
from student_mechanism_design_amd import collector as col, policies as pol
from student_mechanism_design_amd.update import MappoUpdater
dev = torch.device("cuda", 0)
N, P, B, T = 200, 4, 4096, 64
boards = sy.sample_board_pool(8, N, 400, seed=0)
env = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=1, reveal_interval=5, device=dev)
torch.manual_seed(0)
net = pol.MappoPolicy(N, P, hidden_size=64).to(dev)
env.set_policy(pol.DeviceMappoPolicy(net, seed=3))
rec = env.rollout(T)
ret, _ = col.device_returns(rec["reward"], rec["terminated"], 0.99, done_b=rec["truncated"])
up = MappoUpdater(net, env.ell, env.env_graph, minibatch=32768, fused=True)
up.update(rec, ret)
z = up._fused_state()
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())
for mbn in (32768,):
    args = z["_lib"].PpoArgs(ptr(z["image"]), int(z["image_rows"]), 32768, None, mbn, ptr(up.ell), P, N, 64, ptr(z["theta"]), 0.2, 0.5, ptr(z["scratch"]), int(z["scratch"].numel()), ptr(z["grads"]), None, None, None, 3e-4, 0.9, 0.999, 1e-8)
    for _ in range(5):
        z["lib"].sy_mappo_ppo_grad(C.byref(args), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        z["lib"].sy_mappo_ppo_grad(C.byref(args), st)
    e1.record(); torch.cuda.synchronize()
    print("%s: %.1f us per sy_mappo_ppo_grad (%d rows)" % (os.environ.get("SY_ENGINE_LIB", "prod").split("/")[-1], e0.elapsed_time(e1) * 1e3 / 50, mbn))
pos = rec["pos"].reshape(-1, P + 1)
for a in (0, 1):
    c = torch.bincount(pos[:, a].long(), minlength=N).float()
    print("agent", a, "node histogram: max share %.3f, top-5 share %.3f, nodes with >1%% share: %d" % (float(c.max() / c.sum()), float(c.topk(5).values.sum() / c.sum()), int((c / c.sum() > 0.01).sum())))
if "ppot" in os.environ.get("SY_ENGINE_LIB", ""):
    S = z["S"]
    off = 95 * (P + 2) * S
    tw = z["scratch"][off:off + 3 * 256 * 2].view(torch.int64).view(-1, 3).cpu().numpy()
    t0 = tw[:, 0].min()
    import collections
    per = collections.defaultdict(list)
    for b in range(256):
        if tw[b, 1] > 0 and tw[b, 0] >= t0:
            per[int(tw[b, 2]) & 255].append(((tw[b, 0] - t0) / 100.0, (tw[b, 1] - t0) / 100.0, (int(tw[b, 2]) >> 8) / max((tw[b, 1] - tw[b, 0]) / 100.0, 1e-9)))
    for y in sorted(per):
        st = np.array([a for a, _, _ in per[y]]); en = np.array([b for _, b, _ in per[y]]); mhz = np.mean([c for _, _, c in per[y]])
        print("role %2d: %3d blocks, start %.1f-%.1f us, end mean %.1f max %.1f us, run mean %.1f us" % (y, len(st), st.min(), st.max(), en.mean(), en.max(), (en - st).mean()), "| s_memtime ticks per us: %.0f" % mhz)
if "ppot" in os.environ.get("SY_ENGINE_LIB", ""):
    pw = z["scratch"][off + 3 * 256 * 2: off + 3 * 256 * 2 + 8 * 256].view(torch.int32).view(256, 8).cpu().numpy()
    names = ["lookups(ELL+W1t) -> h", "compaction + logits", "softmax + surrogate", "backward + adds", "loop top (prefetch wait)"]
    for y in sorted(per):
        blocks = [b for b in range(256) if tw[b, 1] > 0 and tw[b, 0] >= t0 and (int(tw[b, 2]) & 255) == y]
        if not blocks:
            continue
        m = pw[blocks].mean(0)
        tot = m[:5].sum()
        print("role %2d phases (wave 0, %% of %.0f k ticks): " % (y, tot / 1e3) + ", ".join("%s %.0f%%" % (names[k], 100 * m[k] / max(tot, 1)) for k in range(5)))

if "ppot" in os.environ.get("SY_ENGINE_LIB", ""):
    we = z["scratch"][off + (3 * 256 + 4 * 256) * 2: off + (3 * 256 + 4 * 256 + 16 * 256) * 2].view(torch.int64).view(256, 16).cpu().numpy() / 100.0
    for y in sorted(per):
        blocks = [b for b in range(256) if tw[b, 1] > 0 and tw[b, 0] >= t0 and (int(tw[b, 2]) & 255) == y]
        if blocks:
            m = we[blocks].mean(0)
            print("role %2d: waves leave the row loop after (us, mean over blocks): " % y + " ".join("%.0f" % x for x in m))
