#!/usr/bin/env python
"""The CPU oracle on many host cores at once: W worker PROCESSES (each PFANN_ORACLE_THREADS = 8 torch / BLAS threads -- the
oracle's encoder anti-scales beyond that inside one process), fed through .npy files in a tmpfs directory.  Every worker
runs the whole reference path for its share of the queries -- segmenter -> log-mel -> encoder (torch CPU fp32) -> exact
top-k (BLAS sgemm + argpartition) -> the reference's Python-path sequence matcher -- and reports decisions, the numbers a
flip is classified with, and its stage times.  TEST / BASELINE INFRASTRUCTURE: used by tools/decision_parity.py, the
-m gpu parity test and bench.py's cpu_baseline leg; nothing here is imported by the product.

    python tools/oracle_pool.py --worker <dir> <w> <W>        (internal)
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
QSEG, HOP = 19, 4000


def worker(work, w, W):
    import torch
    torch.set_num_threads(int(os.environ.get("PFANN_ORACLE_THREADS", "8")))
    from oracle import encoder as oe
    from oracle import melspec as om
    from oracle import search as osr
    from oracle import segmenter as osg
    from oracle import seqscore as osq
    meta = json.load(open(os.path.join(work, "meta.json")))
    params, k = meta["params"], meta["k"]
    sd = dict(np.load(os.path.join(work, "weights.npz")))
    db = np.load(os.path.join(work, "db.npy"), mmap_mode="r")          # shared page cache: no per-process copy (3 GB at config 4)
    song_pos = np.load(os.path.join(work, "song_pos.npy"))
    pcm = np.load(os.path.join(work, "q_pcm.npy"), mmap_mode="r")
    emb_gpu = np.load(os.path.join(work, "q_emb_gpu.npy"), mmap_mode="r") if os.path.exists(os.path.join(work, "q_emb_gpu.npy")) else None
    js = list(range(w, pcm.shape[0], W))
    out = {"j": [], "song": [], "sec": [], "score": [], "emb_err": [], "kth": [], "next": [], "runner_up": [], "labels": []}
    st = [0.0, 0.0, 0.0]
    t_begin = time.time()
    for j in js:
        t0 = time.perf_counter()
        segs = osg.segment(osg.pcm_to_mono(np.asarray(pcm[j])[:, None]), 8000, HOP)
        e = oe.encode(om.melspec(segs, params), sd, params)
        t1 = time.perf_counter()
        D, I = osr.flat_ip_topk_blas(e, db, k + 1)
        t2 = time.perf_counter()
        sc, (song, sec), ss = osq.query_embeddings_base(e, I[:, :k], db, song_pos, meta["hop_s"], 1)
        t3 = time.perf_counter()
        st[0] += t1 - t0
        st[1] += t2 - t1
        st[2] += t3 - t2
        two = np.sort(ss[:, 0])[-2:]                       # best per-song scores: winner and the runner-up SONG
        out["emb_err"].append(float(np.abs(e - emb_gpu[j * QSEG:(j + 1) * QSEG]).max()) if emb_gpu is not None else 0.0)
        out["j"].append(j), out["song"].append(song), out["sec"].append(sec), out["score"].append(sc)
        out["kth"].append(D[:, k - 1].copy()), out["next"].append(D[:, k].copy()), out["runner_up"].append(float(two[0]))
        out["labels"].append(I[:, :k].copy())
    np.savez(os.path.join(work, "res_%d.npz" % w), j=np.asarray(out["j"]), song=np.asarray(out["song"]), sec=np.asarray(out["sec"]),
             score=np.asarray(out["score"]), emb_err=np.asarray(out["emb_err"]), kth=np.asarray(out["kth"]),
             next=np.asarray(out["next"]), runner_up=np.asarray(out["runner_up"]), labels=np.asarray(out["labels"]),
             stages=np.asarray(st), span=np.asarray([t_begin, time.time()]))


def run(params, sd, db, song_pos, q_pcm, k, workers=16, q_emb_gpu=None, keep=False):
    """db float32 [N, d], q_pcm int16 [nq, samples] (numpy) -> dict of per-query arrays in query order + timing:
    'compute_s' = first worker's start of work to last worker's end (process start-up and the loading of the database by
    every worker excluded), 'wall_s' = everything, 'stages_s' = summed over the workers."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="pfann_oracle_", dir=base)
    try:
        np.save(os.path.join(work, "db.npy"), np.ascontiguousarray(db, np.float32))
        np.save(os.path.join(work, "song_pos.npy"), np.asarray(song_pos, np.int64))
        np.save(os.path.join(work, "q_pcm.npy"), np.ascontiguousarray(q_pcm, np.int16))
        if q_emb_gpu is not None:
            np.save(os.path.join(work, "q_emb_gpu.npy"), np.ascontiguousarray(q_emb_gpu, np.float32))
        np.savez(os.path.join(work, "weights.npz"), **{n: np.asarray(v) for n, v in sd.items()})
        json.dump({"params": params, "k": k, "hop_s": params["hop_size"]}, open(os.path.join(work, "meta.json"), "w"))
        workers = max(1, min(workers, q_pcm.shape[0]))
        t1 = time.time()
        env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS=os.environ.get("PFANN_ORACLE_THREADS", "8"))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", work, str(w), str(workers)], env=env)
                 for w in range(workers)]
        rcs = [p.wait() for p in procs]
        if any(rcs):
            raise RuntimeError("oracle workers failed: %r" % rcs)
        wall = time.time() - t1
        parts = [np.load(os.path.join(work, "res_%d.npz" % w)) for w in range(workers)]
    finally:
        if not keep:
            shutil.rmtree(work, ignore_errors=True)
    order = np.argsort(np.concatenate([p["j"] for p in parts]))
    out = {name: np.concatenate([p[name] for p in parts])[order]
           for name in ("song", "sec", "score", "emb_err", "kth", "next", "runner_up", "labels")}
    spans = np.stack([p["span"] for p in parts])
    stages = np.sum([p["stages"] for p in parts], axis=0)
    out.update(wall_s=wall, compute_s=float(spans[:, 1].max() - spans[:, 0].min()), workers=workers,
               threads_per_worker=int(os.environ.get("PFANN_ORACLE_THREADS", "8")),
               stages_s={"compute embedding": float(stages[0]), "search": float(stages[1]), "rerank": float(stages[2])})
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
