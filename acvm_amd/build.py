"""Build libacvm_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m acvm_amd.build          # incremental
    python -m acvm_amd.build --force

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libacvm_amd.so")
SOURCES = ["circuit.cpp", "tuning.cpp", "display.cpp", "plan.cpp", "schedule.cpp", "schedule_check.cpp", "batch.cpp", "batch_schedule.cpp", "batch_exact.cpp", "batch_export.cpp", "probes.cpp", "kernels.hip", "kernels_ops.hip", "kernels_hash.hip", "kernels_grumpkin.hip", "grumpkin_host.cpp", "kernels_brillig.hip", "kernels_ecdsa.hip", "shim.cpp", "node.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-result", "-Wno-unused-value",
         "-ffp-contract=off"] + os.environ.get("ACVM_EXTRA_FLAGS", "").split()


def _deps(depfile):
    """prerequisites recorded by the compiler (-MD) at the object's last build, or None"""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    body = text.split(":", 1)[1] if ":" in text else ""
    return [t for t in body.replace("\\\n", " ").split() if t]


def build(force=False, verbose=False):
    objs = []
    jobs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(HERE, "build", s + ".o")
        dep = obj + ".d"
        stale = force or not os.path.exists(obj)
        if not stale:  # rebuilt when the source or any header the last build of this object read is newer (no depfile: rebuild)
            deps = _deps(dep)
            t = os.path.getmtime(obj)
            stale = deps is None or any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps + [src])
        if stale:
            jobs.append([HIPCC] + FLAGS + ["-MD", "-MF", dep, "-c", src, "-o", obj])
        objs.append(obj)
    if jobs:  # translation units are independent: compile them concurrently
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1, 6)) as ex:
            list(ex.map(run, jobs))
    if not jobs and os.path.exists(LIB) and all(os.path.getmtime(o) <= os.path.getmtime(LIB) for o in objs):
        return LIB  # nothing was recompiled and the library is newer than every object
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built", LIB)
