#!/bin/bash
# instruction-cache counters of the dominant kernels (development): bash scripts/dev/pmc_icache.sh <tag> [bench args...]
tag=${1:-icache}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/pmc" -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > "$O/pmc.log" 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(s in r["Kernel_Name"] for s in ("ns_q2_kernel", "ns_hox_kernel", "ns_hop_kernel")):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-28s launches %3d mean %.4e" % (n, len(v), sum(v) / len(v)))
PY
