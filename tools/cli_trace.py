#!/usr/bin/env python
"""GPU timeline of `matcher.py <query list> <db> <result>` under rocprofv3 --kernel-trace (tuning aid / evidence that the
drop-in CLI keeps the GPU busy): generates tools/cli_bench.py's files (kept), runs the matcher as a subprocess under
rocprofv3, and prints the query phase's span, its kernel-busy share, the idle gaps above 1 ms and the kernel table.
    python tools/cli_trace.py [songs] [queries] > profiles/rN/cli_matcher_trace.txt"""
import glob
import os
import shutil
import sqlite3
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tools import cli_bench                                     # noqa: E402


def main():
    n_songs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    res = cli_bench.run(n_songs=n_songs, n_queries=n_q, keep=True, log=lambda *a: print(*a, file=sys.stderr))
    work = res["workdir"]
    try:
        print("# un-profiled: matcher.py %d segments, `total query time` %.3f s (%.0f segments/s), stages %s" %
              (res["matcher"]["segments"], res["matcher"]["total_query_time_s"], res["matcher"]["segments_per_s"], res["matcher"]["stages_s"]))
        out = os.path.join(work, "trace")
        env = dict(os.environ, PYTHONPATH=REPO, TMPDIR="/tmp", PFANN_FAST_EXIT="0")     # (the profiler finalises at exit)
        r = subprocess.run(["rocprofv3", "--kernel-trace", "-d", out, "-o", "t", "--", sys.executable, os.path.join(REPO, "matcher.py"),
                            os.path.join(work, "queries.txt"), os.path.join(work, "db"), os.path.join(work, "result2.txt")],
                           capture_output=True, text=True, env=env, cwd="/tmp")
        tot = [ln for ln in r.stdout.splitlines() if ln.startswith("total query time")]
        print("# under rocprofv3: %s" % (tot[0] if tot else "(no total line; rc %d)" % r.returncode))
        dbs = glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)
        if not dbs:
            print("no trace database written:", r.stderr[-500:])
            return
        c = sqlite3.connect(dbs[0])
        ks = c.execute("select name, start, end from kernels order by start").fetchall()
        # the query phase starts with the first LARGE conversion launch (warm-ups use tiny inputs)
        big = [k for k in ks if "pcm_to_mono" in k[0] and k[2] - k[1] > 20000]
        t0, t1 = big[0][1], ks[-1][2]
        phase = [k for k in ks if k[1] >= t0]
        busy = sum(k[2] - k[1] for k in phase)
        print("# query phase on the GPU: span %.1f ms, kernels %.1f ms = %.1f %% busy, %d launches" %
              ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len(phase)))
        prev = None
        for k in phase:
            if prev is not None and k[1] - prev > 1e6:
                print("#   idle %.2f ms before %s at +%.1f ms" % ((k[1] - prev) / 1e6, k[0][:60], (k[1] - t0) / 1e6))
            prev = max(prev or 0, k[2])
        agg = {}
        for name, s, e in phase:
            a = agg.setdefault(name, [0, 0])
            a[0] += 1
            a[1] += e - s
        print("%-90s %7s %12s %7s" % ("kernel", "calls", "total_us", "%"))
        for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print("%-90s %7d %12.1f %7.2f" % (name[:90], n, t / 1e3, 100.0 * t / busy))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
