#!/bin/bash
# HBM write traffic of the Q2/Q1 residual kernel with the state laid out (eager) and with the stores sent to the one-block sink (lazy)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_res
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
rocprofv3 --pmc $c --output-format csv -d $O/$c -o $c -- python3 $R/scripts/dev/res_lazy_bench.py > $O/$c.log 2>&1
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r06_res/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "ns_q2_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc[r["Kernel_Name"][r["Kernel_Name"].index("<"):r["Kernel_Name"].index(">") + 1]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in acc.items():
        v.sort()
        vals = [x for _, x in v]
        half = len(vals) // 2          # res_lazy_bench.py: lazy first, then eager
        print(c, k, "launches", len(vals), "lazy KB/launch %.0f" % (sum(vals[:half]) / max(half, 1)), "eager KB/launch %.0f" % (sum(vals[half:]) / max(len(vals) - half, 1)))
PY
