#!/usr/bin/env python3
"""print the top rows of a rocprofv3 kernel_stats.csv found below a directory"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f" % (tot / 1e6))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%-80s %7s %9.2f ms %8.1f us %5.1f%%" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                 float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
