# Build timing-only diagnostic variants of the engine in parallel: bash tools/build_variants.sh NAME:"-DFLAG ..." ...
# -> tools/_diag/libsy_NAME.so (git-ignored; travels to the GPU box with gpurun).  Outputs of these builds are WRONG by design.
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-pass-failed -Wno-unused-function $flags \
    student_mechanism_design_amd/csrc/sy_kernels.hip student_mechanism_design_amd/csrc/sy_capi.hip -o tools/_diag/libsy_$name.so > tools/_diag/build_$name.log 2>&1 &
done
wait
ls -la tools/_diag/*.so
