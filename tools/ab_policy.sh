# A/B engine builds on the in-kernel policy rollout (one box): bash tools/ab_policy.sh "prod name1 ..." [reps]
cd $GRAFT_REPO_ROOT
NAMES=$1; REPS=${2:-2}
for rep in $(seq 1 $REPS); do
  for n in $NAMES; do
    if [ "$n" = prod ]; then L=$PWD/student_mechanism_design_amd/libsy_env.so; else L=$PWD/tools/_diag/libsy_$n.so; fi
    SY_ENGINE_LIB=$L timeout -k 5 200 python tools/policy_rollout_bench.py 2>/dev/null | grep MAPPO | sed "s/^/$n: /" | cut -c1-120
  done
done
