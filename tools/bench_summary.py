#!/usr/bin/env python
"""Compact view of a bench.py JSON line (stdin or file): headline, roofline, single-query block, per-kernel times."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
line = [ln for ln in src if ln.startswith("{")][-1]
j = json.loads(line)
print("value %.1f %s  ms/step %.3f  hit %.4f" % (j["value"], j["unit"], j["ms_per_step"], j.get("top1_hit_rate", -1)))
for key in ("roofline", "single_query_scan_roofline", "cpu_baseline", "oracle_decision_parity", "alt_modes", "builder",
            "pcie_inclusive", "seq_score_seam"):
    if j.get(key) is not None:
        print(key, json.dumps(j[key]))
for k, v in j["kernels"].items():
    print("  %-28s %8.3f ms/step  x%-5g %9.1f us  %s" % (k, v["ms_per_step"], v["launches_per_step"], v["avg_us"],
                                                        ("frac %.3f" % v["frac_of_peak"]) if "frac_of_peak" in v else ""))
