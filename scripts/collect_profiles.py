#!/usr/bin/env python3
"""Copy the summaries of one `scripts/measure_round.sh <tag>` run from gpurun_out/<tag>/ into profiles/ (tracked)
and rebuild profiles/pmc_traffic.json, which bench.py reads for `roofline.traffic`.
usage: collect_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def program_lines(path):
    """the program's own output: rocprofv3's timer / warning chatter is dropped"""
    keep = []
    for line in open(path, errors="replace"):
        if line[:1] in "WEI" and "] " in line[:120] and ".cpp:" in line[:120]:
            continue
        keep.append(line)
    return "".join(keep)


def counters(directory, pattern):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pattern in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[r["Kernel_Name"]] = dict(vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"], lds=r["LDS_Block_Size"],
                                              scratch=r["Scratch_Size"], grid=r["Grid_Size"], wg=r["Workgroup_Size"])
    return acc, meta


for name in sorted(os.listdir(src)):
    p = os.path.join(src, name)
    if os.path.isdir(p):
        for f in glob.glob(p + "/**/*_kernel_stats.csv", recursive=True):
            shutil.copy(f, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, name)))
        acc, meta = counters(p, "")
        if acc:
            with open(os.path.join(dst, "%s_%s.txt" % (tag, name)), "w") as out:
                out.write("# rocprofv3 --pmc pass '%s' (scripts/measure_round.sh): mean counter value per launch\n" % name)
                for k, ctr in acc.items():
                    if not any(s in k for s in ("ns_q2", "ns_ho", "seam_fixup", "ho_fixup", "hox_")):
                        continue
                    out.write("%s\n   %s\n" % (k[:140], " ".join("%s=%s" % kv for kv in meta[k].items())))
                    for c, v in sorted(ctr.items()):
                        out.write("   %-26s launches %3d  mean %.5e\n" % (c, len(v), sum(v) / len(v)))
    elif name.endswith(".log"):
        with open(os.path.join(dst, "%s_%s" % (tag, name)), "w") as out:
            out.write(program_lines(p))

# HBM bytes per launch of the dominant kernels; gfx950: FETCH_SIZE counts 64 B per 128-B request of wide (16 B per
# lane) coalesced reads, which is how the Q2/Q1 kernel issues all its loads (MI355X_MICROARCH.md, HBM section)
try:   # (entries of kernels the run did not profile are kept)
    with open(os.path.join(dst, "pmc_traffic.json")) as _f:
        _old = json.load(_f)
except (OSError, ValueError):
    _old = {}
traffic = {"_comment": "scripts/collect_profiles.py from scripts/measure_round.sh %s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                       "separate passes of bench.py --steps 5 --warmup 2; KB per launch averaged.  ns_q2_kernel: hbm_bytes = "
                       "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction for 16-B-per-lane streaming reads), the same for "
                       "ns_hox_kernel (state by 16-B-per-lane LDS-DMA, node lines 8 B per lane contiguous: both count 0.5 "
                       "per scripts/dev/fetch_probe.hip, fetch_size_calibration); the raw figure is given beside it." % tag}


def mean_kb(pass_name, kernel, counter):
    acc, _ = counters(os.path.join(src, pass_name), kernel)
    v = [x for ctr in acc.values() for x in ctr.get(counter, [])]
    return sum(v) / len(v) if v else None


# (the residual mode <..., true, false, false> runs once as set-up; round 5: variant 1 = the recompute-state mode
# <..., false, false, true, false>, variant 4 = the streaming kernel <..., false, false, false, false>; the last parameter: EXT)
for key, passes, kernel in (("128x128x128 k=2 variant=1", ("pmc_q2_fetch", "pmc_q2_write"), "ns_q2_kernel<0, true, true, false, false, false, true, false>"),
                            ("128x128x128 k=2 variant=4", ("pmc_q2s_fetch", "pmc_q2s_write"), "ns_q2_kernel<0, true, true, false, false, false, false, false>")):
    f, w = mean_kb(passes[0], kernel, "FETCH_SIZE"), mean_kb(passes[1], kernel, "WRITE_SIZE")
    if f and w:
        traffic[key] = {"ns_q2_kernel": {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes": int((2 * f + w) * 1024)}}
# calibration of FETCH_SIZE for the access shape of ns_ho_kernel<4> (8 B per lane in runs of 25 lanes):
# scripts/dev/fetch_probe.hip reads a known number of bytes in that shape (and 8 / 16 B per lane contiguously)
PROBE_BYTES = {"read16_contiguous": 2147483648, "read8_contiguous": 2147483648, "read8_runs_of_25": 2147472000}
ratio = {}
for name, nbytes in PROBE_BYTES.items():
    kb = mean_kb("pmc_fetch_probe", name, "FETCH_SIZE")
    if kb:
        ratio[name] = kb * 1024 / nbytes
if ratio:
    traffic["fetch_size_calibration"] = {"FETCH_SIZE_bytes_per_byte_read": ratio,
                                         "note": "gfx950: 0.5 for contiguous 8- and 16-B-per-lane reads, ~0.8 for the "
                                                 "runs of 25 x 8 B of the Q4/Q3 state reads"}
# round 4: the x-marching kernel (ns_hox_kernel) streams its state with 16-B-per-lane LDS-DMA copies like ns_q2_kernel
# (FETCH_SIZE counts half of those bytes: calibrated with the no-state build of the same kernel, pass
# pmc_q4_fetch_nostate -- (FETCH - FETCH_nostate) / known state bytes = 0.51); its node-line reads (8 B per lane, 120-B
# runs in different rows) count in full
f, w = mean_kb("pmc_q4_fetch", "ns_hox_kernel", "FETCH_SIZE"), mean_kb("pmc_q4_write", "ns_hox_kernel", "WRITE_SIZE")
f0 = mean_kb("pmc_q4_fetch_nostate", "ns_hox_kernel", "FETCH_SIZE")
if not f0:   # (the no-state build is a development build: keep the calibration of the run that had it)
    f0 = _old.get("64x64x64 k=4 variant=1", {}).get("ns_hox_kernel", {}).get("FETCH_SIZE_KB_without_state_stream")
if f and w:
    entry = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "raw_bytes": int((f + w) * 1024)}
    if f0:
        state_bytes = 64 ** 3 * 125 * 12 * 8
        entry.update({"FETCH_SIZE_KB_without_state_stream": f0, "state_bytes_algorithmic": state_bytes,
                      "FETCH_SIZE_bytes_per_state_byte": (f - f0) * 1024 / state_bytes,
                      "hbm_bytes": int((2 * (f - f0) + f0 + w) * 1024)})
    else:
        entry["hbm_bytes"] = int((2 * f + w) * 1024)
    traffic["64x64x64 k=4 variant=1"] = {"ns_hox_kernel": entry}
for _k, _v in _old.items():
    traffic.setdefault(_k, _v)
with open(os.path.join(dst, "pmc_traffic.json"), "w") as out:
    json.dump(traffic, out, indent=1)
print(json.dumps(traffic, indent=1))
