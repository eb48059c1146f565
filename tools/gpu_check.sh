# GPU parity tests + the default bench line and its no-belief variant (run through gpurun)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
for f in "" "--no-belief"; do
  timeout -k 5 120 python bench.py --no-cpu $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['value']/1e9,3), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done
