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


FAULT_LIB = os.path.join(HERE, "libsy_env_fault.so")
# Fault-injection build (tests only, never loaded by the product path): one episode stops publishing its hand-offs
# after step 2 and the spins are short, so the status word (sy_env_status) can be seen to fire on a real GPU.
FAULT_FLAGS = ["-DSY_INJECT_LOST_HANDOFF", "-DSY_SPIN_MAX=2048"]


def _command(out, extra=()):
    return [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
            "-Wall", "-Wno-unused-function", "-Wno-pass-failed"] + os.environ.get("SY_HIPCC_FLAGS", "").split() + list(extra) + \
        [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]


def build_extension(force=False, verbose=False, with_fault_build=False):
    """Compile csrc/*.hip -> libsy_env.so for gfx950.  -ffp-contract=off keeps the float64 reward
    arithmetic in the reference's operation order (no FMA contraction).  `with_fault_build` also compiles the
    fault-injection variant (libsy_env_fault.so) next to it; the two hipcc runs go in parallel."""
    jobs = []
    if force or is_stale():
        jobs.append(_command(LIB))
    if with_fault_build and (force or not os.path.exists(FAULT_LIB) or
                             any(os.path.getmtime(os.path.join(CSRC, d)) > os.path.getmtime(FAULT_LIB) for d in DEPS)):
        jobs.append(_command(FAULT_LIB, FAULT_FLAGS))
    procs = []
    for cmd in jobs:
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    return LIB


if __name__ == "__main__":
    import sys
    print(build_extension(force=True, verbose=True, with_fault_build="--fault" in sys.argv))
