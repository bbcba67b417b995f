#!/bin/bash
# kernel trace of one vmult through the forced three-phase schedule (bench.py --through-comm), last timed steps
# usage (GPU box): bash scripts/dev/trace_through_comm.sh <tag>
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tc_q2" -o tc_q2 -- python3 $R/bench.py --through-comm --no-cpu-baseline --steps 20 --warmup 5 > "$O/tc_q2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tc_q4" -o tc_q4 -- python3 $R/bench.py --through-comm --config cavity --cells 32 --no-cpu-baseline --steps 20 --warmup 5 > "$O/tc_q4.log" 2>&1
cd $R
for n in tc_q2 tc_q4; do
python3 - "$O/$n" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last vmult: from the last-but-one sweep-kernel group to the end
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "subtract_scaled" in n or "constrained" in n.lower()]
tail = rows[-40:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    print("%9.1f us  +%7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:90]))
PY
done
