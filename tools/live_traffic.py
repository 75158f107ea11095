#!/usr/bin/env python
"""HBM-side traffic per launch of the encoder's 128-tile conv GEMMs, MEASURED in the run that asks for it
(bench.py's `roofline.traffic`; VERDICT r4 weak 13: the figure used to be a committed constant).

measure() starts this file as a CHILD under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under
`--pmc WRITE_SIZE` (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md's HBM section prescribes; no trace domain
beside the kernel trace).  The child pushes one launch group of 16-bit PCM -- the bench's group: `windows` one-second
windows at a 0.5 s hop out of ten-second clips -- through the library's front end and encoder twice; the counters of every
dispatch of conv_gemm_ln_w22_kernel<*> and conv_gemm_ln_kernel<128, ...> are read from the rocpd database and averaged
per launch, with the guide's gfx950 correction: FETCH_SIZE (KB) counts 16 B/lane streaming reads at half their size -> x2;
WRITE_SIZE (KB) as is.  The encoder's traffic depends on shapes only, so the child needs no database and no real audio.

Nothing here is imported by the product; bench.py calls measure() on rank 0 at N = 1 after its timed loop."""
import glob
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEMM_RE = re.compile(r"(?:void )?pfann::conv_gemm_ln_(w22_kernel|kernel<128)")


def child(windows, device):
    sys.path.insert(0, REPO)
    import torch
    from pfann_amd import synth
    from pfann_amd.engine import Engine
    from pfann_amd.utils import read_config
    params = read_config(os.path.join(REPO, "configs", "default.json"))
    eng = Engine(params, device, max_batch=windows)
    eng.load_state_dict(synth.make_state_dict_calibrated(params, seed=123))
    dev = eng.device
    per_clip, clip = 19, 80000
    n_clips = (windows + per_clip - 1) // per_clip
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    pcm = (torch.randn((n_clips, clip), device=dev, generator=g) * 3000.0).clamp_(-32768, 32767).to(torch.int16)
    starts = (torch.arange(n_clips, device=dev)[:, None] * clip + torch.arange(per_clip, device=dev)[None, :] * 4000).reshape(-1)[:windows]
    for _ in range(2):
        emb = eng.embed_windows(eng.pcm16_to_mono(pcm.reshape(-1)), starts.contiguous())
    torch.cuda.synchronize()
    assert emb.shape[0] == windows and bool(torch.isfinite(emb).all())


def read_counter(db_path, counter):
    """-> {kernel name: (per-dispatch average of the counter summed over its hardware instances, dispatches)}"""
    c = sqlite3.connect(db_path)
    rows = c.execute("select k.name, count(distinct k.id), sum(e.value) from rocpd_pmc_event e "
                     "join rocpd_info_pmc p on e.pmc_id = p.id join kernels k on k.id = e.event_id "
                     "where p.name = ? group by k.name", (counter,)).fetchall()
    return {name: (v / max(n, 1), n) for name, n, v in rows}


def one_pass(counter, windows, device, out_dir, timeout_s):
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PFANN_GPUS"):
        env.pop(k, None)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out_dir, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--child", str(windows), str(device)]
    # its own process group: a pass that has to be given up takes the profiler AND the child under it along
    proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        _, err = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        import signal
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.communicate()
        raise RuntimeError("rocprofv3 --pmc %s: no result within %d s" % (counter, timeout_s))
    dbs = glob.glob(os.path.join(out_dir, "**", "*_results.db"), recursive=True)
    if proc.returncode != 0 or not dbs:
        raise RuntimeError("rocprofv3 --pmc %s: rc %d, %d result files: %s" % (counter, proc.returncode, len(dbs), (err or "")[-400:]))
    return read_counter(dbs[0], counter)


def measure(windows=9728, device=0, timeout_s=90, keep_dir=None):
    """-> dict(hbm_bytes_per_launch, fetch_size_kb_per_launch, write_size_kb_per_launch, dispatches, kernels, seconds, ...).
    Raises on any failure (no rocprofv3, a pass that times out, no matching dispatch): the caller keeps its fallback."""
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 is not on PATH")
    t0 = time.time()
    top = keep_dir or tempfile.mkdtemp(prefix="pfann_live_traffic_", dir="/tmp")
    try:
        fe = one_pass("FETCH_SIZE", windows, device, os.path.join(top, "fetch"), timeout_s)
        wr = one_pass("WRITE_SIZE", windows, device, os.path.join(top, "write"), timeout_s)
    finally:
        if keep_dir is None:
            shutil.rmtree(top, ignore_errors=True)
    names = [n for n in fe if GEMM_RE.match(n)]
    nd = sum(fe[n][1] for n in names)
    if not names or nd == 0 or any(wr.get(n, (0, 0))[1] != fe[n][1] for n in names):
        raise RuntimeError("no conv GEMM dispatches in the PMC passes (fetch %r, write %r)" % (sorted(fe)[:4], sorted(wr)[:4]))
    f_kb = sum(fe[n][0] * fe[n][1] for n in names) / nd
    w_kb = sum(wr[n][0] * wr[n][1] for n in names) / nd
    return {"hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0, "fetch_size_kb_per_launch": f_kb, "write_size_kb_per_launch": w_kb,
            "dispatches": nd, "launches_per_group": nd // 2, "windows_per_group": windows,
            "kernels": {n[:70]: {"dispatches": fe[n][1], "hbm_bytes_per_launch": (2.0 * fe[n][0] + wr[n][0]) * 1024.0} for n in names},
            "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane loads), WRITE_SIZE x1; launch-weighted over conv_gemm_ln_w22_kernel<*> "
                          "and conv_gemm_ln_kernel<128,*>",
            "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- tools/live_traffic.py --child %d" % windows,
            "seconds": round(time.time() - t0, 1)}


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    else:
        print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 9728), indent=1))
