# Which regime is the fused rollout in?  Same per-wave work at 4 / 2 / 1 waves per SIMD (run through gpurun).
cd $GRAFT_REPO_ROOT
for cfg in "--envs 4096 --wpb 16" "--envs 2048 --wpb 8" "--envs 1024 --wpb 4" "--envs 512 --wpb 2" "--envs 4096 --wpb 16 --no-belief" "--envs 2048 --wpb 8 --no-belief" "--envs 4096 --wpb 16 --no-record" "--envs 2048 --wpb 8 --no-record"; do
  timeout -k 5 120 python bench.py --no-cpu --no-verify $cfg "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', 'G/s', round(d['value']/1e9,3), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'blocks*wpb', d['config']['waves_per_block'])"
done
