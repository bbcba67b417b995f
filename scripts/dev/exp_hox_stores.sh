#!/bin/bash
# store-granularity experiment on the Q4/Q3 x-marching kernel: time + WRITE_SIZE / FETCH_SIZE per build
# usage (GPU box): bash scripts/dev/exp_hox_stores.sh "<defines>" "<defines>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp_hox_stores
mkdir -p $O
i=0
for e in "$@"; do
  i=$((i+1))
  cd $R
  bash scripts/exp_ho.sh "$e" 2>&1 | grep "^exp\|^{" | tee -a $O/times.log
  cd /tmp && export TMPDIR=/tmp
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${i}_$c -o p -- python3 $R/scripts/bench_ho.py 4 1 0 > $O/pmc_${i}_$c.log 2>&1
    python3 - <<PY | tee -a $O/times.log
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("$O/pmc_${i}_$c/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "ns_hox_kernel" in r["Kernel_Name"]]
print("   [$e] $c ns_hox_kernel: launches %d mean %.1f KB"%(len(v), sum(v)/max(len(v),1)))
PY
  done
done
