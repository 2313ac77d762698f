#!/usr/bin/env python
"""Developer tool: per-kernel means (mi:: kernels only) of a rocprofv3 --kernel-trace --stats CSV.   python tools/kstats.py <dir-or-csv> [min_calls]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
for r in csv.DictReader(open(path)):
    if "mi::" in r["Name"] and int(r["Calls"]) >= min_calls:
        n = r["Name"].split("(")[0].replace("void ", "")
        print("%-58s calls %4s avg %9.1f us  min %9.1f" % (n[:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
