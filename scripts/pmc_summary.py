#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc counter_collection.csv below a directory.
usage: pmc_summary.py <dir> [kernel-name substring]"""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if pat not in name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[name] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                      r["Grid_Size"], r["Workgroup_Size"])
for name, ctr in acc.items():
    print("%s\n   vgpr %s agpr %s sgpr %s lds %s scratch %s grid %s wg %s" % ((name[:110],) + meta[name]))
    for c, v in sorted(ctr.items()):
        print("   %-28s n=%3d  mean %.4e" % (c, len(v), sum(v) / len(v)))
