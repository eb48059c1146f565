# Collect SQ instruction / activity counters for the fused rollout (run on the GPU box through gpurun).
# usage: bash tools/collect_sq.sh <tag> [bench flags...]   -> gpurun_out/sq_<tag>_pass{1,2}.csv
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
tag=$1; shift
for pass in 1 2; do
  ctrs=$(sed -n "${pass}p" tools/pmc_sq.txt | sed 's/^pmc: //')
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/sq_${tag}_p${pass} -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > gpurun_out/sq_${tag}_p${pass}.log 2>&1 || exit 1
  f=$(ls gpurun_out/sq_${tag}_p${pass}/*/*_counter_collection.csv | head -1)
  cp $f gpurun_out/sq_${tag}_pass${pass}.csv
done
