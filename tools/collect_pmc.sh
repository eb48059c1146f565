# Collect the counters listed in a pmc file (one "pmc:" line per pass) for the fused rollout (run through gpurun).
# usage: bash tools/collect_pmc.sh <pmc file> <tag> [bench flags...]   -> gpurun_out/<tag>_pass<N>.csv
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
file=$1; tag=$2; shift 2
n=$(grep -c '^pmc:' $file)
for pass in $(seq 1 $n); do
  ctrs=$(grep '^pmc:' $file | sed -n "${pass}p" | sed 's/^pmc: //')
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/${tag}_p${pass} -- python3 bench.py --no-cpu --no-verify --steps 3 --warmup 1 --repeats 1 "$@" > gpurun_out/${tag}_p${pass}.log 2>&1 || { tail -5 gpurun_out/${tag}_p${pass}.log; exit 1; }
  f=$(ls gpurun_out/${tag}_p${pass}/*/*_counter_collection.csv | head -1)
  cp $f gpurun_out/${tag}_pass${pass}.csv
done
