# Round-3 variant table (run through gpurun): the bench line over seeds 0..4, pool sizes 1 / 8 / 64, other
# shapes / fused lengths, diagnostics; then the secondary benches (learned policy in the loop, collectors).
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03_variants}; mkdir -p $O
run() { timeout -k 5 150 python bench.py --no-cpu --no-config3 --timed-seconds 0.8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '| G agent-steps/s', round(d['value']/1e9,3), '| kernel_ms', round(d['roofline']['kernel_ms'],4), '| frac', round(d['roofline']['frac'],3), '| verified', d['verified'], '|', d['roofline']['kernel'])"; }
{
for g in 8 1 64; do for s in 0 1 2 3 4; do run --graphs $g --seed $s; done; done
run --fused 128
run --fused 512
run --envs 8192
run --police 6
run --nodes 199 --police 5 --envs 8192
run --no-record --no-verify
run --no-belief --no-verify
run --no-record --no-belief --no-verify
} > $O/variants.log 2>&1
cat $O/variants.log
python - $O/variants.log << 'PY' | tee $O/variants_summary.log
import re, sys, statistics as st
rows = [l for l in open(sys.argv[1]) if l.startswith("--graphs")]
for g in ("8", "1", "64"):
    v = [float(re.search(r"steps/s ([0-9.]+)", l).group(1)) for l in rows if l.startswith("--graphs %s " % g)]
    k = [float(re.search(r"kernel_ms ([0-9.]+)", l).group(1)) for l in rows if l.startswith("--graphs %s " % g)]
    ok = all("verified True" in l for l in rows if l.startswith("--graphs %s " % g))
    print("pool of %2s boards, seeds 0-4: %.2f G agent-steps/s mean (min %.2f, max %.2f), kernel %.4f ms mean, all verified: %s" % (g, st.mean(v), min(v), max(v), st.mean(k), ok))
PY
timeout -k 5 300 python tools/policy_rollout_bench.py > $O/policy_rollout_bench.log 2>&1; tail -8 $O/policy_rollout_bench.log
timeout -k 5 300 python tools/collector_bench.py > $O/collector_bench.log 2>&1; tail -8 $O/collector_bench.log
timeout -k 5 300 python tools/update_bench.py > $O/update_bench.log 2>&1; tail -6 $O/update_bench.log
# per-kernel times of the fused PPO update (rocprofv3 kernel trace of 10 updates of 8 minibatches)
R=$GRAFT_REPO_ROOT; (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/update_prof -- python3 $R/tools/update_bench.py "fused kernel, 8 minibatches, eager" > $R/$O/update_prof.log 2>&1)
cp $(ls $O/update_prof/*/*kernel_stats.csv | head -1) $O/update_kernel_stats.csv && head -8 $O/update_kernel_stats.csv | cut -c1-160
