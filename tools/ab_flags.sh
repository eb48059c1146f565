# A/B compiler flags / launch parameters on the GPU box (diagnostic)
cd $GRAFT_REPO_ROOT
run() { timeout -k 5 120 python bench.py --no-cpu $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 | $2 |', round(d['value']/1e9,3), round(d['roofline']['kernel_ms'],4))"; }
for fl in "" "-fno-unroll-loops" "-O2" "-mllvm -amdgpu-schedule-metric-bias=0"; do
  SY_HIPCC_FLAGS="$fl" python -c "from student_mechanism_design_amd.build import build_extension; build_extension(force=True)" > /dev/null 2>&1 || { echo "build failed: $fl"; continue; }
  run "flags=[$fl]" ""
done
python -c "from student_mechanism_design_amd.build import build_extension; build_extension(force=True)" > /dev/null 2>&1
for w in 2 4 6 8; do run "default" "--wpb $w"; done
run default "--fused 32"; run default "--fused 128"; run default "--graphs 1"; run default "--police 6"; run default "--police 2"
