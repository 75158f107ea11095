#!/usr/bin/env python
"""Per-layer table of the fused conv GEMM from a `PFANN_PROF_LAYERS=1 python bench.py ...` JSON line
(the library then tags every launch with its layer shape):
    python tools/per_layer_table.py gpurun_out/per_layer.json > profiles/r2/per_layer.txt
Columns: tile, rows per sample, M (= B * rows), N, K_live, us per launch, TFLOP/s, fraction of the fp32 MFMA peak,
share of the summed GEMM time."""
import json
import re
import sys

PEAK = 157.3


def main(path):
    line = [ln for ln in open(path) if ln.startswith("{")][-1]
    js = json.loads(line)
    B = js["config"]["segments_per_step"] // js["n_gpus"]
    rows = []
    for tag, kv in js["kernels"].items():
        m = re.match(r"(conv_gemm_ln_\d+(?: w22)?) rows=(\d+) K=(\d+) N=(\d+)( first)?", tag)
        if not m:
            continue
        rps, K, N = int(m.group(2)), int(m.group(3)), int(m.group(4))
        rows.append((m.group(1) + (m.group(5) or ""), rps, B * rps, N, K, kv["launches_per_step"], kv["avg_us"],
                     kv["work_per_launch"]))
    tot = sum(r[5] * r[6] for r in rows)
    print("# %s\n# B = %d segments per launch; peak = %.1f TFLOP/s (fp32 MFMA)" % (js["metric"], B, PEAK))
    print("%-24s %6s %10s %6s %6s %3s %10s %8s %6s %6s" % ("kernel", "rows", "M", "N", "K_live", "n", "us", "TFLOP/s", "frac", "share"))
    for r in sorted(rows, key=lambda r: -r[5] * r[6]):
        tf = r[7] / (r[6] * 1e-6) / 1e12
        print("%-24s %6d %10d %6d %6d %3g %10.1f %8.1f %6.3f %6.3f" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], tf, tf / PEAK,
                                                                      r[5] * r[6] / tot))
    work = sum(r[5] * r[7] for r in rows)
    print("# all GEMM launches: %.1f us per step, %.1f TFLOP/s = %.3f of peak" % (tot, work / (tot * 1e-6) / 1e12,
                                                                                 work / (tot * 1e-6) / 1e12 / PEAK))


if __name__ == "__main__":
    main(sys.argv[1])
