#!/usr/bin/env python
"""The drop-in CLIs measured end to end from WAV files on disk (BASELINE config 2's shape: 10 k songs -> ~590 k x 128
db, 2000 ten-second SNR-0 queries):

    python builder.py <music list> <db dir> <model dir>          (reference builder.py:30-44)
    python matcher.py <query list> <db dir> <result file>        (reference matcher.py:34-42)

run as SUBPROCESSES, exactly as a user of the reference would start them.  Songs and queries come from the seeded
generators of pfann_amd/synth.py (written as 8 kHz mono 16-bit WAVs to a tmpfs directory), the weights are the seeded
calibrated state_dict.  Reports, per CLI: wall time of the whole process (interpreter start, torch import, model/db
load included), the tool's own `total build time` / `total query time` line (what the reference prints: work after
the model and database are loaded), segments/s by both clocks, and the stage split under the reference's stage names
(tools/stat.py:17) parsed from the tool's stdout.  bench.py --cli and tests/test_gpu_cli.py call run().
"""
import argparse
import json
import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
SEG_PER_SONG, QUERY_SEGS = 59, 19


def _wav_header(n_samples, sr=8000, ch=1):
    data = n_samples * ch * 2
    return (b"RIFF" + struct.pack("<I", 36 + data) + b"WAVEfmt " +
            struct.pack("<IHHIIHH", 16, 1, ch, sr, sr * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", data))


def write_wavs(paths, pcm_rows):
    """pcm_rows: int16 numpy [len(paths), n] mono @ 8 kHz."""
    hdr = _wav_header(pcm_rows.shape[1])
    for p, row in zip(paths, pcm_rows):
        with open(p, "wb") as f:
            f.write(hdr)
            f.write(row.tobytes())


def pick_tmp(need_bytes):
    """a tmpfs directory with room for the WAVs (falls back to the system temp dir)"""
    for base in ("/dev/shm", tempfile.gettempdir()):
        try:
            if shutil.disk_usage(base).free > need_bytes * 1.3 + (1 << 30):
                return tempfile.mkdtemp(prefix="pfann_cli_", dir=base)
        except OSError:
            pass
    return None


def parse_startup(text):
    """'startup <what> <seconds>s' lines (PFANN_STARTUP_TIMING=1) -> dict"""
    out = {}
    for ln in text.splitlines():
        m = re.match(r"^startup (.+) ([0-9.]+)s$", ln.strip())
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


def parse_stdout(text):
    """'<stage> <seconds>s' lines + the total line -> dict"""
    stages, total = {}, None
    for ln in text.splitlines():
        if ln.startswith("startup "):
            continue
        m = re.match(r"^(total (?:build|query) time) ([0-9.]+)s$", ln.strip())
        if m:
            total = float(m.group(2))
            continue
        m = re.match(r"^([a-z][a-z ]+?) ([0-9.]+)s$", ln.strip())
        if m:
            stages[m.group(1)] = round(float(m.group(2)), 4)
    return stages, total


def run(n_songs=10000, n_queries=2000, snr=0.0, device=0, keep=False, workdir=None, log=print, gpus=1, tool_timeout_s=3600, config="default"):
    """gpus > 1: both tools are started with PFANN_GPUS=gpus and launch their own ranks, one per GPU (the database is
    then built by, and sharded over, all of them)."""
    import torch
    from pfann_amd import synth
    from pfann_amd.utils import read_config
    dev = torch.device("cuda", device)
    params = read_config(os.path.join(REPO, "configs", config + ".json"))      # (default / seg / n640d64: same segment geometry)
    d_emb = params["model"]["d"]
    need = n_songs * SEG_PER_SONG * 4000 * 2 * 1.02 + n_queries * 160000 + n_songs * SEG_PER_SONG * 128 * 4 * 2.2 + \
        n_queries * n_songs * 8
    work = workdir or pick_tmp(need)
    if work is None:
        return {"skipped": "no temp directory with %.1f GB free" % (need / 1e9)}
    os.makedirs(work, exist_ok=True)
    t_gen = time.time()
    try:
        # ---- model dir (builder.py:38-44 form: a directory holding configs.json + model.pt)
        mdir = os.path.join(work, "model")
        os.makedirs(mdir, exist_ok=True)
        sd = synth.make_state_dict_calibrated(params, seed=123)
        torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, os.path.join(mdir, "model.pt"))
        cfg = dict(params)
        cfg["indexer"] = dict(params["indexer"], index_factory="Flat")        # the exact index the build serves (seg.json:47)
        json.dump(cfg, open(os.path.join(mdir, "configs.json"), "w"))
        # ---- songs
        sdir = os.path.join(work, "songs")
        os.makedirs(sdir, exist_ok=True)
        music = [os.path.join(sdir, "s%06d.wav" % i) for i in range(n_songs)]
        for c0 in range(0, n_songs, 256):
            ids = list(range(c0, min(c0 + 256, n_songs)))
            write_wavs(music[c0:c0 + len(ids)], synth.make_songs_torch(ids, 30.0, device=dev).cpu().numpy())
        mlist = os.path.join(work, "music.txt")
        open(mlist, "w").write("".join(p + "\n" for p in music))
        # ---- queries (genquery.py protocol: random 10 s crop + noise at `snr`, peak-normalised, 16 bit)
        qdir = os.path.join(work, "queries")
        os.makedirs(qdir, exist_ok=True)
        q_song = [int((j * 7919 + 13) % n_songs) for j in range(n_queries)]
        queries = [os.path.join(qdir, "q%06d.wav" % j) for j in range(n_queries)]
        q_off = []
        for c0 in range(0, n_queries, 256):
            c1 = min(c0 + 256, n_queries)
            qp, qo = synth.make_queries_torch(synth.make_songs_torch(q_song[c0:c1], 30.0, device=dev),
                                              list(range(c0, c1)), 10.0, snr)
            write_wavs(queries[c0:c1], qp.cpu().numpy())
            q_off.append(qo.cpu().numpy())
        q_off = np.concatenate(q_off) if q_off else np.zeros(0)
        qlist = os.path.join(work, "queries.txt")
        open(qlist, "w").write("".join(p + "\n" for p in queries))
        torch.cuda.synchronize()
        t_gen = time.time() - t_gen
        log("cli_bench: wrote %d songs + %d queries (%.1f GB) to %s in %.1f s" %
            (n_songs, n_queries, (n_songs * SEG_PER_SONG * 8000 + n_queries * 160000) / 1e9, work, t_gen))

        env = dict(os.environ, PYTHONPATH=REPO)
        under_launcher = "TORCHELASTIC_RUN_ID" in env
        for k in list(env):                   # nothing of a launcher this process may itself be running under
            if k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
                     "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_PORT") or k.startswith("TORCHELASTIC_") or \
                    k.startswith("TORCH_NCCL_ASYNC") or k == "NCCL_ASYNC_ERROR_HANDLING" or \
                    (k == "OMP_NUM_THREADS" and under_launcher):         # (the launcher's own default of 1)
                env.pop(k, None)
        if gpus > 1:
            env["PFANN_GPUS"] = str(gpus)
        db = os.path.join(work, "db")
        out = {"config": config, "songs": n_songs, "queries": n_queries, "snr_db": snr, "dir": os.path.dirname(work), "gpus": gpus,
               "decode_workers": int(os.environ.get("PFANN_DECODE_WORKERS", "8")),
               "wav_bytes": int(n_songs * SEG_PER_SONG * 8000 + n_queries * 160000)}
        env["PFANN_STARTUP_TIMING"] = "1"
        # ---- builder
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(REPO, "builder.py"), mlist, db, mdir], capture_output=True,
                           text=True, env=dict(env, PFANN_T0=repr(t0)), cwd=work, timeout=tool_timeout_s)
        wall = time.time() - t0
        if r.returncode != 0:
            raise RuntimeError("builder.py failed:\n" + r.stdout[-2000:] + r.stderr[-3000:])
        stages, total = parse_stdout(r.stdout)
        n_seg = n_songs * SEG_PER_SONG
        key = np.fromfile(os.path.join(db, "landmarkKey"), dtype=np.int32)
        assert int(key.sum()) == n_seg and os.path.getsize(os.path.join(db, "embeddings")) == n_seg * d_emb * 4
        out["builder"] = {"segments": n_seg, "process_wall_s": round(wall, 3), "total_build_time_s": total,
                          "segments_per_s_process": round(n_seg / wall, 1),
                          "segments_per_s": round(n_seg / total, 1) if total else None, "stages_s": stages,
                          "startup_s": round(wall - total, 3) if total else None, "startup_split_s": parse_startup(r.stdout)}
        log("cli_bench: builder.py %d segments: process %.2f s, `total build time` %.2f s, stages %s" % (n_seg, wall, total or -1, stages))
        # ---- matcher
        result = os.path.join(work, "result.txt")
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(REPO, "matcher.py"), qlist, db, result], capture_output=True,
                           text=True, env=dict(env, PFANN_T0=repr(t0)), cwd=work, timeout=tool_timeout_s)
        wall = time.time() - t0
        if r.returncode != 0:
            raise RuntimeError("matcher.py failed:\n" + r.stdout[-2000:] + r.stderr[-3000:])
        stages, total = parse_stdout(r.stdout)
        n_qseg = n_queries * QUERY_SEGS
        rows = [ln.rstrip("\n").split("\t") for ln in open(result, encoding="utf8")]
        assert len(rows) == n_queries
        hit = sum(1 for j, (qn, ans) in enumerate(rows) if ans == music[q_song[j]])
        assert os.path.getsize(result + ".bin") == n_queries * n_songs * 8
        out["matcher"] = {"segments": n_qseg, "process_wall_s": round(wall, 3), "total_query_time_s": total,
                          "segments_per_s_process": round(n_qseg / wall, 1),
                          "segments_per_s": round(n_qseg / total, 1) if total else None, "stages_s": stages,
                          "startup_s": round(wall - total, 3) if total else None, "startup_split_s": parse_startup(r.stdout),
                          "top1_hit_rate": round(hit / max(n_queries, 1), 4),
                          "bin_bytes": n_queries * n_songs * 8}
        log("cli_bench: matcher.py %d segments: process %.2f s, `total query time` %.2f s, hit-rate %.4f, stages %s" %
            (n_qseg, wall, total or -1, hit / max(n_queries, 1), stages))
        out["generate_s"] = round(t_gen, 2)
        if keep:
            out["workdir"] = work
        return out
    finally:
        if not keep:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=10000)
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--snr", type=float, default=0.0)
    ap.add_argument("--gpus", type=int, default=1, help="> 1: the tools start their own ranks (PFANN_GPUS)")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--config", default="default", help="configs/<name>.json: default, seg, n640d64")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = run(a.songs, a.queries, a.snr, keep=a.keep, log=lambda *x: print(*x, file=sys.stderr, flush=True), gpus=a.gpus, config=a.config)
    print(json.dumps(res))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)
