"""Developer tool: per-launch means of the PMC passes collected by tools/prof_pmc.sh for the headline
SpMM kernel -> profiles/r01_spmm_pmc_raw_tagged.json and profiles/spmm_traffic.json (read by bench.py).
    python tools/pmc_summary.py [gpurun_out/pmc] [kernel-name-substring]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
want = sys.argv[2] if len(sys.argv) > 2 else "k_spmm<"

means = {}
for f in sorted(glob.glob(os.path.join(src, "p*", "*counter_collection.csv"))):
    acc = defaultdict(lambda: defaultdict(float))  # counter -> dispatch -> value (summed over dimensions / XCDs)
    for r in csv.DictReader(open(f)):
        if want not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for name, per in acc.items():
        means[name] = sum(per.values()) / len(per)
        means[name + "__launches"] = len(per)
if not means:
    sys.exit("no rows for kernel %r under %s" % (want, src))
fetch_kb, write_kb = means["FETCH_SIZE"], means["WRITE_SIZE"]
hit, miss = means["TCC_HIT_sum"], means["TCC_MISS_sum"]
out = {
    "kernel": "k_spmm<float,4,32,4,true> (hot/cold tagged gather, 8 MiB hot budget)",
    "workload": "bench.py default (R-MAT scale 20, N=128 fp32)",
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "fetch_bytes_raw": fetch_kb * 1024, "fetch_bytes_corrected_x2": fetch_kb * 2048, "write_bytes": write_kb * 1024,
    "hbm_bytes_per_launch": fetch_kb * 2048 + write_kb * 1024,
    "TCC_HIT": hit, "TCC_MISS": miss, "l2_hit_rate": hit / (hit + miss), "cross_check_TCC_MISS_x128B": miss * 128,
    "note": "FETCH_SIZE = TCC_EA0_RDREQ x 64 B on gfx950 and reads exactly half of a wide (16 B/lane) coalesced stream "
            "(MI355X_MICROARCH.md, HBM section): doubled. WRITE_SIZE matches the expected C + carry bytes uncorrected. "
            "Separate rocprofv3 --pmc passes (tools/prof_pmc.sh), kernel-trace only; summarised by tools/pmc_summary.py.",
}
old = os.path.join(ROOT, "profiles", "spmm_traffic.json")
if os.path.exists(old):
    prev = json.load(open(old))
    if "untagged_reference" in prev:
        out["untagged_reference"] = prev["untagged_reference"]
json.dump(means, open(os.path.join(ROOT, "profiles", "r01_spmm_pmc_raw_tagged.json"), "w"), indent=1, sort_keys=True)
json.dump(out, open(old, "w"), indent=1)
print(json.dumps(out, indent=1))
