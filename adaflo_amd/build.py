"""Build the HIP engine (adaflo_amd/lib/libadaflo_hip.so) for gfx950 with hipcc.

In-tree build so the shared object travels with the repo snapshot to the GPU box.
"""
import glob
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libadaflo_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [
        os.path.join(_HERE, "..", "include", "adaflo_hip.h")]
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = ["hipcc", "-c", s, "-o", o] + FLAGS
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed for " + s)
    if force or procs or _stale(LIB, objs):
        cmd = ["hipcc", "-shared", "-o", LIB] + objs + ["--offload-arch=" + ARCH, "-fno-gpu-rdc"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
