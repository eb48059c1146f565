# A/B engine builds on the same GPU box: bash tools/ab_libs.sh "libA.so libB.so ..." [bench flags]
cd $GRAFT_REPO_ROOT
LIBS=$1; shift
for rep in 1 2 3; do
  for L in $LIBS; do
    SY_ENGINE_LIB=$PWD/$L timeout -k 5 120 python bench.py --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value']/1e9,3), round(d['roofline']['kernel_ms'],4))"
  done
done
