# Round-2 variant table (run through gpurun): the bench line on other shapes / pools / fused lengths.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_final; mkdir -p $O
run() { timeout -k 5 150 python bench.py --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '| G agent-steps/s', round(d['value']/1e9,3), '| kernel_ms', round(d['roofline']['kernel_ms'],4), '| frac', round(d['roofline']['frac'],3), '| verified', d['verified'], '|', d['roofline']['kernel'].split(' ')[0])"; }
{
run
run --graphs 1
run --graphs 64
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
timeout -k 5 200 python tools/policy_rollout_bench.py > $O/policy_rollout_bench.log 2>&1; tail -3 $O/policy_rollout_bench.log
timeout -k 5 300 python tools/collector_bench.py > $O/collector_bench.log 2>&1; tail -8 $O/collector_bench.log
timeout -k 10 500 python tools/config3_bench.py > $O/config3_bench.log 2>&1; tail -3 $O/config3_bench.log
