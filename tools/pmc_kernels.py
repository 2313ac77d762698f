#!/usr/bin/env python
"""Developer tool: per-kernel means of rocprofv3 --pmc passes (one sub-directory per pass) for ANY workload.

    python tools/pmc_kernels.py <dir with p*/ sub-directories> [min share of dispatches]
Prints one JSON object per kernel name: launches, and the mean per launch of every counter collected (summed over
XCDs / instances), plus derived l2_hit, fetch_GB_x2 (FETCH_SIZE is tallied at 64 B per 128-B request on gfx950 for
wide loads: doubled, MI355X_MICROARCH.md), write_GB."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
for f in sorted(glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"])[:120]
        acc[k][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
for k, ctrs in sorted(acc.items()):
    row = {"kernel": k}
    for name, d in ctrs.items():
        row[name] = sum(d.values()) / len(d)
        row["launches"] = len(d)
    if "TCC_HIT_sum" in row and row["TCC_HIT_sum"] + row.get("TCC_MISS_sum", 0) > 0:
        row["l2_hit"] = round(row["TCC_HIT_sum"] / (row["TCC_HIT_sum"] + row["TCC_MISS_sum"]), 4)
    if "FETCH_SIZE" in row:
        row["fetch_GB_x2"] = round(row["FETCH_SIZE"] * 2048 / 1e9, 4)
    if "WRITE_SIZE" in row:
        row["write_GB"] = round(row["WRITE_SIZE"] * 1024 / 1e9, 4)
    if "SQ_WAVE_CYCLES" in row and row["SQ_WAVE_CYCLES"] > 0:
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
            if c in row:
                row[c + "_frac"] = round(row[c] / row["SQ_WAVE_CYCLES"], 4)
    print(json.dumps(row))
