# A/B engine builds on ONE GPU box, interleaved: bash tools/ab_quick.sh "prod name1 name2" [reps] [bench flags]   (libs: tools/_diag/libsy_<name>.so; prod = shipped)
cd $GRAFT_REPO_ROOT
NAMES=$1; REPS=${2:-3}; shift; shift
for rep in $(seq 1 $REPS); do
  for n in $NAMES; do
    if [ "$n" = prod ]; then L=$PWD/student_mechanism_design_amd/libsy_env.so; else L=$PWD/tools/_diag/libsy_$n.so; fi
    SY_ENGINE_LIB=$L timeout -k 5 120 python bench.py --no-cpu --no-config3 --timed-seconds 0.4 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', 'G/s', round(d['value']/1e9,3), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'min', round(d['roofline']['kernel_ms_min'],4), 'verified', d['verified'])"
  done
done
