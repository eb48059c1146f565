# Round-2 secondary measurements (run through gpurun): learned policy in the loop, configs[2] end to end.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_final; mkdir -p $O
timeout -k 5 200 python tools/policy_rollout_bench.py > $O/policy_rollout_bench.log 2>&1 || { tail -5 $O/policy_rollout_bench.log; exit 1; }
tail -3 $O/policy_rollout_bench.log
timeout -k 5 300 python tools/collector_bench.py > $O/collector_bench.log 2>&1 || { tail -5 $O/collector_bench.log; exit 1; }
tail -6 $O/collector_bench.log
timeout -k 10 500 python tools/config3_bench.py > $O/config3_bench.log 2>&1 || { tail -5 $O/config3_bench.log; exit 1; }
tail -4 $O/config3_bench.log
