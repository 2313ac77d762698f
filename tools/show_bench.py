#!/usr/bin/env python
"""Developer tool: the fields of a bench.py line that matter at a glance.   python tools/show_bench.py <log>"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline", "plan", "cpu_baseline") if k in d})[:3500])
for k, v in d.get("secondary", {}).items():
    rf = v.get("roofline") or {}
    print(k, {kk: v.get(kk) for kk in ("ms", "ms_per_step", "value", "error", "first_call_ms") if kk in v},
          {kk: rf.get(kk) for kk in ("frac", "traffic", "traffic_over_algorithmic")})
