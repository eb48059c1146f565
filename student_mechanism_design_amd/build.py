"""Build the HIP engine library in-tree (hipcc, gfx950 only).

`python -m student_mechanism_design_amd.build` or `build_extension()`; the resulting `libsy_env.so` sits next to this
file so it travels with the source tree.  The engine is one translation unit per kernel family / instance group
(csrc/*.hip, see csrc/sy_kernels.h): the units compile in parallel into csrc/_obj/<variant>/ and are linked into one
shared library, so an edit to the hot rollout only rebuilds the units that include it.
"""
import hashlib
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsy_env.so")
FAULT_LIB = os.path.join(HERE, "libsy_env_fault.so")
OBJ_ROOT = os.path.join(CSRC, "_obj")
# translation units, the slowest first (the scheduler starts them in this order)
SOURCES = ["sy_rollout3_p.hip", "sy_rollout3_q.hip", "sy_rollout2_c.hip", "sy_rollout1_b.hip", "sy_rollout2_a.hip", "sy_rollout1_a.hip",
           "sy_rollout3_a.hip", "sy_rollout3_b.hip", "sy_rollout3_c.hip", "sy_rollout3_d.hip", "sy_rollout2_b.hip",
           "sy_policy.hip", "sy_aux.hip", "sy_step.hip", "sy_gnn.hip", "sy_ppo.hip", "sy_returns.hip", "sy_dispatch.hip", "sy_capi.hip"]
HEADERS = {
    "sy_kernels.h": ["sy_kernels.h", os.path.join("..", "..", "include", "sy_env.h")],
    "sy_device.hpp": ["sy_device.hpp", "@sy_kernels.h"],
    "sy_pair.hpp": ["sy_pair.hpp", "@sy_device.hpp"],
    "sy_rollout3.hpp": ["sy_rollout3.hpp", "@sy_pair.hpp"],
    "sy_rollout_legacy.hpp": ["sy_rollout_legacy.hpp", "@sy_pair.hpp"],
}
# Fault-injection build (tests only, never loaded by the product path): one episode stops publishing its hand-offs
# after step 2 and the spins are short, so the status word (sy_env_status) can be seen to fire on a real GPU.
FAULT_FLAGS = ["-DSY_INJECT_LOST_HANDOFF", "-DSY_SPIN_MAX=2048"]
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
              "-Wno-pass-failed"]


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _header_deps(name, seen=None):
    seen = set() if seen is None else seen
    for d in HEADERS[name]:
        if d.startswith("@"):
            _header_deps(d[1:], seen)
        else:
            seen.add(os.path.normpath(os.path.join(CSRC, d)))
    return seen


def source_deps(src):
    """Files a translation unit depends on: itself + the headers it (transitively) includes."""
    deps = {os.path.join(CSRC, src)}
    with open(os.path.join(CSRC, src)) as f:
        text = f.read()
    for h in HEADERS:
        if '#include "%s"' % h in text:
            deps |= _header_deps(h)
    return deps


def existing_sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def is_stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for s in existing_sources() for d in source_deps(s))


def source_digest():
    """sha256 over the engine's sources (what `build_id` in the library is derived from; bench.py compares the stamp
    of stored profiles with it)."""
    h = hashlib.sha256()
    files = set()
    for s in existing_sources():
        files |= source_deps(s)
    for f in sorted(files):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _variant(tag, extra, out, force, verbose, jobs):
    """Queue the compile jobs of one library variant; returns (objects, link command, stamps to write)."""
    odir = os.path.join(OBJ_ROOT, tag)
    os.makedirs(odir, exist_ok=True)
    flags = BASE_FLAGS + os.environ.get("SY_HIPCC_FLAGS", "").split() + list(extra)
    stamp = os.path.join(odir, "flags.txt")
    flag_text = " ".join(flags)
    if not os.path.exists(stamp) or open(stamp).read() != flag_text:
        force = True
    # the library carries a digest of its sources (sy_build_id): only sy_capi.hip sees it, and is recompiled when it moves
    digest = source_digest()
    id_stamp = os.path.join(odir, "build_id.txt")
    id_moved = not os.path.exists(id_stamp) or open(id_stamp).read() != digest
    objs = []
    for s in existing_sources():
        o = os.path.join(odir, s.replace(".hip", ".o"))
        objs.append(o)
        own = ['-DSY_BUILD_ID="%s"' % digest] if s == "sy_capi.hip" else []
        if force or (own and id_moved) or not os.path.exists(o) or \
                any(os.path.getmtime(d) > os.path.getmtime(o) for d in source_deps(s)):
            jobs.append(([hipcc_path()] + flags + own + ["-c", os.path.join(CSRC, s), "-o", o], s))
    link = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    return objs, link, ((stamp, flag_text), (id_stamp, digest))


def _run_jobs(jobs, verbose, max_parallel):
    running, t0 = [], time.time()
    jobs = list(jobs)
    while jobs or running:
        while jobs and len(running) < max_parallel:
            cmd, name = jobs.pop(0)
            if verbose:
                print("[%5.1fs] hipcc %s" % (time.time() - t0, " ".join(cmd[-3:])), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        for item in list(running):
            cmd, pr = item
            rc = pr.poll()
            if rc is None:
                continue
            running.remove(item)
            if rc != 0:
                for _, other in running:
                    other.kill()
                raise subprocess.CalledProcessError(rc, cmd)
        time.sleep(0.05)


def build_extension(force=False, verbose=False, with_fault_build=False, extra_flags=(), out=None, tag=None):
    """Compile csrc/*.hip -> libsy_env.so for gfx950.  -ffp-contract=off keeps the float64 reward arithmetic in the
    reference's operation order (no FMA contraction).  `with_fault_build` also compiles the fault-injection variant
    (libsy_env_fault.so); all translation units of all variants share one pool of parallel hipcc processes.
    `extra_flags` + `out` + `tag`: diagnostic variants (tools/build_variants.sh)."""
    max_parallel = int(os.environ.get("SY_BUILD_JOBS", "0")) or max(1, min(16, (os.cpu_count() or 4)))
    jobs, links = [], []
    main_out = out or LIB
    main_tag = tag or ("main" if not extra_flags else "x" + hashlib.sha256(" ".join(extra_flags).encode()).hexdigest()[:8])
    if force or is_stale(main_out) or extra_flags:
        links.append(_variant(main_tag, list(extra_flags), main_out, force, verbose, jobs))
    if with_fault_build and (force or is_stale(FAULT_LIB)):
        links.append(_variant("fault", FAULT_FLAGS, FAULT_LIB, force, verbose, jobs))
    _run_jobs(jobs, verbose, max_parallel)
    for objs, link, stamps in links:
        if verbose:
            print("link", link[-1], flush=True)
        subprocess.check_call(link)
        for path, text in stamps:
            with open(path, "w") as f:
                f.write(text)
    return main_out


if __name__ == "__main__":
    t0 = time.time()
    print(build_extension(force="--force" in sys.argv, verbose=True, with_fault_build="--no-fault" not in sys.argv))
    print("build took %.1f s" % (time.time() - t0))
