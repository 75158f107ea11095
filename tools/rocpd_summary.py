#!/usr/bin/env python
"""Per-kernel summary (calls, total, avg, min, max, %) of a rocprofv3 rocpd sqlite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db on ROCm 7.2), plus
PMC counter sums per kernel when the run collected any.  Used to produce profiles/*.txt."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    # bench.py brackets its timed region with two pfann_bench_region_marker launches
    marks = [r[0] for r in c.execute("select start from kernels where name like '%pfann_bench_region_marker%' "
                                     "order by start").fetchall()]
    where = ""
    if len(marks) >= 2:
        where = "where start > %d and start < %d" % (marks[0], marks[-1])
        print("# restricted to bench.py's timed region (between the two pfann_bench_region_marker launches): "
              "%.3f ms of GPU timeline" % ((marks[-1] - marks[0]) / 1e6))
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels %s group by name "
                     "order by sum(duration) desc" % where).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-86s %7s %12s %11s %11s %11s %6s %5s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us",
                                                          "max_us", "%", "vgpr", "lds"))
    for name, n, s, a, mn, mx, vg, av, lds in rows:
        print("%-86s %7d %12.1f %11.1f %11.1f %11.1f %6.2f %5d %7d" %
              (name[:86], n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot, (vg or 0) + (av or 0), lds or 0))
    try:
        w2 = where.replace("start", "k.start")
        pm = c.execute("select k.name, p.name, count(distinct k.id), sum(e.value) from rocpd_pmc_event e "
                       "join rocpd_info_pmc p on e.pmc_id = p.id "
                       "join kernels k on k.id = e.event_id %s group by k.name, p.name order by k.name" % w2).fetchall()
    except sqlite3.Error as x:
        print("pmc query failed:", x)
        pm = []
    if pm:
        print("\nPMC counters, summed over the counter's hardware instances; per-dispatch average over the "
              "dispatches in the region:")
        for kn, pn, n, v in pm:
            print("%-70s %-26s dispatches=%-5d sum=%.6g per_dispatch=%.6g" % (kn[:70], pn, n, v, v / max(n, 1)))


if __name__ == "__main__":
    main(sys.argv[1])
