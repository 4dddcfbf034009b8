#!/usr/bin/env python3
"""One line per configuration of a bench.py JSON line: rate, roofline fraction, CPU baseline."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline: %s = %.4g %s, %.3f ms per step, roofline frac %.3f, cpu %s" % (
    d["metric"][:40], d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value")))
for k, v in d.get("other_configs", {}).items():
    r = v.get("roofline") or {}
    print("%-78s %10s sweeps/s  frac %-7s cpu %s" % (k[:78], v.get("sweeps_per_s"), r.get("frac"), (v.get("cpu_baseline") or {}).get("value")))
