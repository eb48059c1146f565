import numpy as np, torch, sys
sys.path.insert(0, '.')
import student_mechanism_design_amd as sy
boards = sy.sample_board_pool(1, 50, 90, seed=6)
w = np.linspace(0.1, 0.9, 11)
for trial in range(4):
    eager = sy.BatchedScotlandYardEnv(128, boards, 4, 15, w, seed=4, reveal_interval=5)
    graphed = sy.BatchedScotlandYardEnv(128, boards, 4, 15, w, seed=4, reveal_interval=5)
    rec = eager.alloc_rollout(6); rec_g = graphed.alloc_rollout(6)
    act = torch.full((128, 5), -1, dtype=torch.int32, device=eager.device)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            graphed.rollout(6, out=rec_g)
            graphed.step(act)
    torch.cuda.current_stream().wait_stream(side)
    for it in range(3):
        g.replay()
        eager.rollout(6, out=rec)
        eager.step(act)
        torch.cuda.synchronize()
        bad = [n for n in ("pos","budget","t","step_count","_mask","_belief","_visits","reward","_terminated") if not torch.equal(getattr(eager,n), getattr(graphed,n))]
        rb = [k for k in ("record","mask","belief") if not torch.equal(rec[k], rec_g[k])]
        print('trial',trial,'iter',it,'state mismatch',bad,'record mismatch',rb)
        if rb:
            k=rb[0]; d=(rec[k]!=rec_g[k]).nonzero()[:5].tolist(); print('  first diffs',k,d)
# eager vs eager determinism
a = sy.BatchedScotlandYardEnv(128, boards, 4, 15, w, seed=4, reveal_interval=5)
b = sy.BatchedScotlandYardEnv(128, boards, 4, 15, w, seed=4, reveal_interval=5)
ra=a.alloc_rollout(6); rb_=b.alloc_rollout(6)
for it in range(5):
    a.rollout(6,out=ra); a.step(act); b.rollout(6,out=rb_); b.step(act); torch.cuda.synchronize()
    print('eager-eager iter',it,[k for k in ("record","mask","belief") if not torch.equal(ra[k], rb_[k])], torch.equal(a.pos,b.pos))
