# Build timing-only diagnostic variants of the engine: bash tools/build_variants.sh NAME:"-DFLAG ..." ...
# -> tools/_diag/libsy_NAME.so (git-ignored; travels to the GPU box with gpurun).  Outputs of such builds may be WRONG by design.
# Each variant compiles its translation units in parallel (student_mechanism_design_amd/build.py), variants one after the other.
cd "$(dirname "$0")/.."
mkdir -p tools/_diag
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  python - "$name" $flags << 'PY' > tools/_diag/build_$name.log 2>&1 || { tail -20 tools/_diag/build_$name.log; exit 1; }
import sys
from student_mechanism_design_amd.build import build_extension
name, flags = sys.argv[1], sys.argv[2:]
build_extension(force=False, verbose=True, extra_flags=flags, out="tools/_diag/libsy_%s.so" % name, tag="diag_" + name)
PY
done
ls -la tools/_diag/*.so
