# Time diagnostic engine builds on one GPU box: bash tools/ab_variants.sh "name1 name2 ..." [bench flags]   (libs: tools/_diag/libsy_<name>.so; "prod" = shipped)
cd $GRAFT_REPO_ROOT
NAMES=$1; shift
for cfg in "" "--no-belief"; do
 for n in $NAMES; do
  if [ "$n" = prod ]; then L=$PWD/student_mechanism_design_amd/libsy_env.so; else L=$PWD/tools/_diag/libsy_$n.so; fi
  SY_ENGINE_LIB=$L timeout -k 5 120 python bench.py --no-cpu --no-verify $cfg "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n $cfg', 'G/s', round(d['value']/1e9,3), 'kernel_ms', round(d['roofline']['kernel_ms'],4))"
 done
done
