#!/usr/bin/env python3
"""GPU idle gaps in a rocprofv3 kernel trace: which kernel precedes the idle time"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in rows)
t_from = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0       # fraction of the span to skip (setup)
t0 = ev[0][0] + t_from * (ev[-1][1] - ev[0][0])
ev = [e for e in ev if e[0] >= t0]
span, busy = ev[-1][1] - ev[0][0], sum(e - s for s, e, _ in ev)
print("span %.3f s  busy %.3f s  kernels %d" % (span / 1e9, busy / 1e9, len(ev)))
gaps = collections.defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    g = s1 - e0
    if g > 0:
        gaps[n0][0] += g
        gaps[n0][1] += 1
for n, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print("%-72s gap total %8.1f ms  count %6d  avg %7.1f us" % (n, g / 1e6, c, g / c / 1e3))
