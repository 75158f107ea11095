#!/usr/bin/env python
"""profiles/<round>/pmc_fetch.txt + pmc_write.txt -> profiles/traffic.json: HBM-side bytes per
launch of each kernel, corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950:
FETCH_SIZE (KB) under-counts wide (16 B/lane) streaming reads by exactly 2x -> doubled;
WRITE_SIZE (KB) taken as is.  bench.py copies the dominant kernel's figure into
roofline.traffic."""
import json
import re
import sys


def parse(path, counter):
    out = {}
    for ln in open(path):
        m = re.match(r"(.*?)\s+%s\s+dispatches=(\d+)\s+sum=(\S+)\s+per_dispatch=(\S+)" % counter, ln)
        if m:
            out[m.group(1).strip()] = (float(m.group(4)), int(m.group(2)))
    return out


def main(d):
    fe, wr = parse(d + "/pmc_fetch.txt", "FETCH_SIZE"), parse(d + "/pmc_write.txt", "WRITE_SIZE")
    res = {}
    for k in fe:
        f, nd = fe[k]
        w = wr.get(k, (0.0, 0))[0]
        res[k] = {"fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w, "dispatches": nd,
                  "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                  "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane loads), WRITE_SIZE x1", "source": d}
    # template instantiations of one kernel that bench.py times under one tag (e.g. the fused GEMM with
    # and without the folded first conv): launch-weighted average
    groups = {}
    for k, r in res.items():
        m = re.match(r"(?:void )?(pfann::\w+<\d+(?:, \d+)*)", k)
        if m:
            groups.setdefault(m.group(1), []).append(r)
    for g, rs in groups.items():
        if len(rs) > 1:
            nd = sum(r["dispatches"] for r in rs)
            res[g + ", ...> (all instantiations)"] = {
                "dispatches": nd, "source": d,
                "fetch_size_kb_per_launch": sum(r["fetch_size_kb_per_launch"] * r["dispatches"] for r in rs) / nd,
                "write_size_kb_per_launch": sum(r["write_size_kb_per_launch"] * r["dispatches"] for r in rs) / nd,
                "hbm_bytes_per_launch": sum(r["hbm_bytes_per_launch"] * r["dispatches"] for r in rs) / nd,
                "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane loads), WRITE_SIZE x1; launch-weighted over instantiations"}
    # bench.py's profiling tag "conv_gemm_ln_128" = every 128x128-tile conv GEMM launch: the five-block kernel
    # (conv_gemm_ln_w22_kernel, both variants) and what is left on conv_gemm_ln_kernel<128, ...>
    rs = [r for k, r in res.items() if re.match(r"(?:void )?pfann::conv_gemm_ln_(w22_kernel|kernel<128)", k) and "all instantiations" not in k]
    if rs:
        nd = sum(r["dispatches"] for r in rs)
        res["tag:conv_gemm_ln_128"] = {
            "dispatches": nd, "source": d,
            "fetch_size_kb_per_launch": sum(r["fetch_size_kb_per_launch"] * r["dispatches"] for r in rs) / nd,
            "write_size_kb_per_launch": sum(r["write_size_kb_per_launch"] * r["dispatches"] for r in rs) / nd,
            "hbm_bytes_per_launch": sum(r["hbm_bytes_per_launch"] * r["dispatches"] for r in rs) / nd,
            "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane loads), WRITE_SIZE x1; launch-weighted over "
                          "conv_gemm_ln_w22_kernel<*> and conv_gemm_ln_kernel<128,*>"}
    # which build the counters were collected on: bench.py copies this next to roofline.traffic, so that the line says
    # the figure is a committed observation of THAT commit, not a measurement of the run that prints it
    import hashlib
    import subprocess
    try:
        head = subprocess.check_output(["git", "rev-parse", "HEAD"], text=True).strip()
        dirty = bool(subprocess.check_output(["git", "status", "--porcelain", "--", "pfann_amd", "bench.py"], text=True).strip())
    except (OSError, subprocess.CalledProcessError):
        head, dirty = None, None
    src = hashlib.sha256()
    import glob
    for f in sorted(glob.glob("pfann_amd/csrc/*.hip") + glob.glob("pfann_amd/csrc/*.h")):
        src.update(open(f, "rb").read())
    res["_meta"] = {"profiled_commit": head, "tree_dirty_when_written": dirty, "csrc_sha16": src.hexdigest()[:16], "profile_dir": d,
                    "command": "tools/profile_bench.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes)"}
    json.dump(res, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1)[:1500])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r1")
