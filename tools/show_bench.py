#!/usr/bin/env python3
"""Print kernel_ms / ms_per_step / roofline frac of bench.py JSON lines (files given as arguments)."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s kernel_ms %.4f ms_per_step %.4f frac %.3f value %.3e" % (f, r["kernel_ms"], d["ms_per_step"], r["frac"], d["value"]))
