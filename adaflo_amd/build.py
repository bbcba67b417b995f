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
# the superseded Q3..Q5 kernels (ns_ho.hip: kernel variant 2, ns_hop.hip: variant 3) are compiled into the library only on
# request; the stamp file makes a change of the setting rebuild the two units
if os.environ.get("ADAFLO_BUILD_VARIANTS") == "1":
    FLAGS.append("-DADAFLO_BUILD_VARIANTS")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


AUDIT = os.path.join(LIBDIR, ".isa_audit.json")


def _audit_listing(path, remarks=""):
    from . import isa_audit
    with open(path, errors="replace") as f:
        rec = isa_audit.summarize(isa_audit.flow_block_copies(f.read()))
    rec.update(isa_audit.scratch_use(remarks))
    return rec


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [
        os.path.join(_HERE, "..", "include", "adaflo_hip.h")]
    objs = []
    procs = []
    audits = []
    stamp, setting = os.path.join(LIBDIR, ".variants_setting"), os.environ.get("ADAFLO_BUILD_VARIANTS", "0")
    changed = not os.path.exists(stamp) or open(stamp).read() != setting
    import json
    report = {}
    if os.path.exists(AUDIT):
        try:
            report = json.load(open(AUDIT))
        except ValueError:
            report = {}
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs) or (changed and os.path.basename(s) in ("ns_ho.hip", "ns_hop.hip", "capi.hip")) or \
                os.path.basename(s)[:-4] not in report:   # (no audit record of the unit: compile it again)
            cmd = ["hipcc", "-c", s, "-o", o] + FLAGS
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
            # the device listing of the same unit, for the audit below (a compile of its own: the product object is built
            # exactly as before)
            listing = o[:-2] + ".gfx950.s"
            # (-Rpass-analysis=kernel-resource-usage: scratch bytes per lane of every kernel, recorded beside the audit)
            audits.append((s, listing, subprocess.Popen(["hipcc", "-S", "--cuda-device-only", s, "-o", listing,
                                                         "-Rpass-analysis=kernel-resource-usage"] + FLAGS,
                                                        stdout=subprocess.DEVNULL, stderr=open(listing + ".remarks", "w"))))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed for " + s)
    # ISA audit (adaflo_amd/isa_audit.py): register-allocator copies under a partial EXEC mask in Flow blocks
    for s, listing, p in audits:
        unit = os.path.basename(s)[:-4]
        if p.wait() != 0 or not os.path.exists(listing):
            raise RuntimeError("hipcc -S failed for " + s)
        with open(listing + ".remarks", errors="replace") as f:
            remarks = f.read()
        report[unit] = _audit_listing(listing, remarks)
        os.remove(listing)
        os.remove(listing + ".remarks")
    if audits:
        with open(AUDIT, "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
    bad = {u: r for u, r in report.items() if r.get("copies", 0) and os.path.exists(os.path.join(CSRC, u + ".hip"))}
    if bad and os.environ.get("ADAFLO_ALLOW_FLOW_COPIES") != "1":
        raise RuntimeError("ISA audit: register-allocator copies in Flow blocks ahead of the EXEC flip (they execute under the "
                           "THEN mask; DESIGN.md section 8): %s -- restructure the branch, or ADAFLO_ALLOW_FLOW_COPIES=1" % bad)
    with open(stamp, "w") as f:
        f.write(setting)
    if force or procs or _stale(LIB, objs):
        cmd = ["hipcc", "-shared", "-o", LIB] + objs + ["--offload-arch=" + ARCH, "-fno-gpu-rdc"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


# Test infrastructure (tests/test_lb_differential_gpu.py): the kernels that ship at one workgroup per CU (512 registers) built
# once more for two (256 registers) -- the same source, the same floating-point operations, another register allocation.  The
# product never loads these libraries.
VARIANTS = {"q2_lb2": ("ns_q2", ["-DQ2_RES_LB=2", "-DQ2_RCP_LB=2", "-DQ2_EXT_LB=2"]),
            "hox_lb2": ("ns_hox", ["-DHOX_RES_LB=2", "-DHOX_EXT_LB=2"])}


def build_variants(force=False, verbose=False):
    build(verbose=verbose)
    vdir = os.path.join(LIBDIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(_HERE, "..", "include", "adaflo_hip.h")]
    jobs, libs = [], {}
    for tag, (unit, flags) in VARIANTS.items():
        lib, obj, src = os.path.join(vdir, "lib_%s.so" % tag), os.path.join(vdir, "%s_%s.o" % (unit, tag)), os.path.join(CSRC, unit + ".hip")
        libs[tag] = lib
        if force or _stale(lib, [src, LIB] + hdrs):
            cmd = ["hipcc", "-c", src, "-o", obj] + FLAGS + flags
            if verbose:
                print(" ".join(cmd))
            jobs.append((tag, unit, obj, lib, subprocess.Popen(cmd)))
    for tag, unit, obj, lib, p in jobs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed for variant " + tag)
        objs = [os.path.join(LIBDIR, f) for f in sorted(os.listdir(LIBDIR)) if f.endswith(".o") and f != unit + ".o"]
        subprocess.check_call(["hipcc", "-shared", "-o", lib] + objs + [obj, "--offload-arch=" + ARCH, "-fno-gpu-rdc"])
        os.remove(obj)
    return libs


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--variants" in sys.argv:
        print(build_variants(force="--force" in sys.argv, verbose=True))
