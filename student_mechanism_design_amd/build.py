"""Build the HIP engine library in-tree (hipcc, gfx950 only).

`python -m student_mechanism_design_amd.build` or `build_extension()`; the resulting
`libsy_env.so` sits next to this file so it travels with the source tree.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsy_env.so")
SOURCES = ["sy_kernels.hip", "sy_capi.hip"]
DEPS = SOURCES + ["sy_kernels.h", os.path.join("..", "..", "include", "sy_env.h")]


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_extension(force=False, verbose=False):
    """Compile csrc/*.hip -> libsy_env.so for gfx950.  -ffp-contract=off keeps the float64 reward
    arithmetic in the reference's operation order (no FMA contraction)."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wall", "-Wno-unused-function"] + os.environ.get("SY_HIPCC_FLAGS", "").split() + \
        [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
