set -e
cd $GRAFT_REPO_ROOT
for mw in 8 4; do
  SY_HIPCC_FLAGS="-DSY_ROLLOUT_MIN_WAVES=$mw" python -c "from student_mechanism_design_amd.build import build_extension; build_extension(force=True)" > /dev/null 2>&1
  for f in "" "--no-belief" "--no-record"; do timeout -k 5 120 python bench.py --no-cpu $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"minwaves=$mw $f\", round(d[\"value\"]/1e9,3), round(d[\"roofline\"][\"kernel_ms\"],4))"; done
done
python -c "from student_mechanism_design_amd.build import build_extension; build_extension(force=True)" > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
