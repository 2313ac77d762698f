#!/usr/bin/env python
"""Developer tool: attribute the per-dispatch counters of `rocprofv3 --pmc ... -- python tools/spmm_sweep.py`
to the sweep's variants (dispatch order = variant order, launches_per_variant dispatches each; the first of
each group is the warm-up and is dropped) and print one table row per variant.

    python tools/pmc_sweep_summary.py <sweep stdout log> <pmc dir with p*/ sub-directories> [kernel substring]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

log, src = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else "k_spmm<"
rows = [json.loads(l) for l in open(log) if l.startswith("{")]
head, variants = rows[0], rows[1:]
per = head["launches_per_variant"]
table = [dict(v) for v in variants]
for f in sorted(glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(float))  # counter -> dispatch id -> value summed over XCDs / instances
    for r in csv.DictReader(open(f)):
        if want not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for name, d in acc.items():
        ids = sorted(d)
        if len(ids) != per * len(variants):
            print("# %s: %d dispatches, expected %d -- skipped" % (name, len(ids), per * len(variants)))
            continue
        for i, v in enumerate(table):
            grp = ids[i * per + 1:(i + 1) * per]
            v[name] = sum(d[k] for k in grp) / len(grp)
for v in table:
    if "TCC_HIT_sum" in v:
        v["l2_hit"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
    if "FETCH_SIZE" in v:
        v["fetch_GB_x2"] = round(v["FETCH_SIZE"] * 2048 / 1e9, 3)
    if "WRITE_SIZE" in v:
        v["write_GB"] = round(v["WRITE_SIZE"] * 1024 / 1e9, 3)
    if "fetch_GB_x2" in v and "write_GB" in v:
        v["traffic_GB"] = round(v["fetch_GB_x2"] + v["write_GB"], 3)
        v["traffic_over_algorithmic"] = round(v["traffic_GB"] * 1e9 / head["algorithmic_bytes"], 2)
    print(json.dumps(v))
