"""GPU timeline of ONE 10 s query (tuning aid): from a rocprofv3 --kernel-trace database of tools/ubench/one_query.py,
the kernels of one whole-query call in launch order with their start offsets, durations and the idle gap before each.
    rocprofv3 --kernel-trace -d DIR -o t -- python tools/ubench/one_query.py ; python tools/ubench/one_query_timeline.py DIR/.../t_results.db"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    # a whole-query call starts with pcm_to_mono_kernel and ends with the last kernel before the next one
    firsts = [i for i, r in enumerate(rows) if "pcm_to_mono_kernel" in r[0]]
    calls = [(firsts[j], firsts[j + 1]) for j in range(len(firsts) - 1)]
    # keep the calls that contain a search and a match (the "whole query" loops), last 40
    whole = [(a, b) for a, b in calls if any("match" in rows[i][0] for i in range(a, b)) and
             any("scan" in rows[i][0] for i in range(a, b))][-40:]
    if not whole:
        print("no whole-query call found")
        return
    spans, sums, counts = [], [], []
    for a, b in whole:
        ks = rows[a:b]
        spans.append((ks[-1][2] - ks[0][1]) / 1e3)
        sums.append(sum(k[2] - k[1] for k in ks) / 1e3)
        counts.append(len(ks))
    spans.sort(); sums.sort()
    print("# %d whole-query calls: kernels per call %d, GPU span median %.1f us, sum of kernel durations median %.1f us" %
          (len(whole), counts[0], spans[len(spans) // 2], sums[len(sums) // 2]))
    a, b = whole[len(whole) // 2]
    t0 = rows[a][1]
    prev = None
    print("%9s %8s %8s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
    for name, s, e in rows[a:b]:
        print("%9.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, name[:100]))
        prev = e


if __name__ == "__main__":
    main(sys.argv[1])
