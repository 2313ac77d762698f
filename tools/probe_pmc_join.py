#!/usr/bin/env python
"""Join a spmm_gather_probe log (dispatch ranges per variant) with rocprofv3 --pmc counter CSVs of the same run.
    python tools/r04/probe_pmc_join.py <probe log> <dir with *counter_collection.csv>"""
import csv, glob, os, re, sys
from collections import defaultdict
log, src = sys.argv[1], sys.argv[2]
variants = []
for line in open(log):
    m = re.match(r"(.*?)\s+min ([\d.]+) avg ([\d.]+) ms\s+\[dispatches (\d+)\.\.(\d+), (\d+) per rep\]", line)
    if m:
        variants.append((m.group(1).strip(), float(m.group(2)), int(m.group(4)), int(m.group(5)), int(m.group(6))))
vals = defaultdict(lambda: defaultdict(float))  # dispatch id -> counter -> value
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gather" not in r["Kernel_Name"]:
            continue
        vals[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(vals)
for name, tmin, a, b, per in variants:
    sel = ids[a + per:b + 1]  # skip the warm-up rep
    if not sel:
        continue
    tot = defaultdict(float)
    for i in sel:
        for k, v in vals[i].items():
            tot[k] += v
    reps = len(sel) / per
    out = {k: v / reps for k, v in tot.items()}
    hit = out.get("TCC_HIT_sum", 0.0); miss = out.get("TCC_MISS_sum", 0.0)
    extra = ""
    if hit + miss > 0:
        extra += " l2_hit %.3f req %.1fM" % (hit / (hit + miss), (hit + miss) / 1e6)
    if "TCC_EA0_RDREQ_sum" in out:
        extra += " ea_rd %.2f GB(128B)" % (out["TCC_EA0_RDREQ_sum"] * 128 / 1e9)
    print("%-64s min %.3f ms%s" % (name, tmin, extra))
