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
    files = json.load(open(os.path.join(work, "q_files.json"))) if os.path.exists(os.path.join(work, "q_files.json")) else None
    pcm = np.load(os.path.join(work, "q_pcm.npy"), mmap_mode="r") if files is None else None
    n_q = len(files) if files is not None else pcm.shape[0]
    keep_ss = bool(meta.get("keep_song_scores"))
    bank = np.load(os.path.join(work, "bank.npy")) if os.path.exists(os.path.join(work, "bank.npy")) else None
    emb_gpu = np.load(os.path.join(work, "q_emb_gpu.npy"), mmap_mode="r") if os.path.exists(os.path.join(work, "q_emb_gpu.npy")) else None
    js = list(range(w, n_q, W))
    out = {"j": [], "song": [], "sec": [], "score": [], "emb_err": [], "kth": [], "next": [], "runner_up": [], "labels": [], "ss": []}
    st = [0.0, 0.0, 0.0]
    t_begin = time.time()
    # batch_queries > 1 (the parity tools; bench.py's cpu_baseline keeps 1, the reference's one-query-at-a-time loop):
    # several queries share one encoder call and one sgemm over the database -- the host is the scarce resource of every
    # parity run, and a 19-row sgemm streams the whole database for 19 rows' worth of work.  Rows per call are bounded by
    # the score matrix it makes (rows x N floats <= 1 GB)
    bq = max(1, int(meta.get("batch_queries", 1)))
    bq = max(1, min(bq, (1 << 28) // max(1, db.shape[0] * QSEG)))
    ready = []                                            # (j, e, D, I) of the current batch, and the stage time each owes
    for b0 in range(0, len(js), bq):
        t0 = time.perf_counter()
        seg_list = []
        for j in js[b0:b0 + bq]:
            if files is not None:                         # the oracle reads the WAV file itself (oracle/segmenter.py)
                seg_list.append(osg.load_segments(files[j], params))
            else:
                seg_list.append(osg.segment(osg.pcm_to_mono(np.asarray(pcm[j])[:, None]), 8000, HOP))
        cuts = np.cumsum([0] + [x.shape[0] for x in seg_list])
        e_all = oe.encode(om.melspec(np.concatenate(seg_list), params, bank=bank), sd, params)
        t1 = time.perf_counter()
        D_all, I_all = osr.flat_ip_topk_blas(e_all, db, k + 1)
        t2 = time.perf_counter()
        st[0] += t1 - t0
        st[1] += t2 - t1
        ready += [(j, e_all[cuts[i]:cuts[i + 1]], D_all[cuts[i]:cuts[i + 1]], I_all[cuts[i]:cuts[i + 1]])
                  for i, j in enumerate(js[b0:b0 + bq])]
    for j, e, D, I in ready:
        t2 = time.perf_counter()
        sc, (song, sec), ss = osq.query_embeddings_base(e, I[:, :k], db, song_pos, meta["hop_s"], 1)
        t3 = time.perf_counter()
        st[2] += t3 - t2
        two = np.sort(ss[:, 0])[-2:]                       # best per-song scores: winner and the runner-up SONG
        out["emb_err"].append(float(np.abs(e - emb_gpu[j * QSEG:(j + 1) * QSEG]).max()) if emb_gpu is not None else 0.0)
        out["j"].append(j), out["song"].append(song), out["sec"].append(sec), out["score"].append(sc)
        out["kth"].append(D[:, k - 1].copy()), out["next"].append(D[:, k].copy()), out["runner_up"].append(float(two[0]))
        out["labels"].append(I[:, :k].copy())
        if keep_ss:
            out["ss"].append(ss)
    if keep_ss:
        np.save(os.path.join(work, "ss_%d.npy" % w), np.asarray(out["ss"], np.float32).reshape(len(js), -1, 2))
    np.savez(os.path.join(work, "res_%d.npz" % w), j=np.asarray(out["j"]), song=np.asarray(out["song"]), sec=np.asarray(out["sec"]),
             score=np.asarray(out["score"]), emb_err=np.asarray(out["emb_err"]), kth=np.asarray(out["kth"]),
             next=np.asarray(out["next"]), runner_up=np.asarray(out["runner_up"]), labels=np.asarray(out["labels"]),
             stages=np.asarray(st), span=np.asarray([t_begin, time.time()]))


def embed_worker(work, w, W):
    """Fingerprints only, three ways (tools/embedding_error_budget.py): the fp32 oracle; the same op sequence in float64
    from the float64 log-mel (both with the oracle's own float64-built mel bank); and float64 again with the mel bank the
    PRODUCT hands its kernel (bank.npy: built the way torchaudio builds it, in fp32 torch ops) -- the two banks differ by
    up to 3.8e-5 of a unit-peak weight, and that alone moves fingerprints by more than all fp32 rounding together."""
    import torch
    torch.set_num_threads(int(os.environ.get("PFANN_ORACLE_THREADS", "8")))
    from oracle import encoder as oe
    from oracle import melspec as om
    from oracle import segmenter as osg
    meta = json.load(open(os.path.join(work, "meta.json")))
    params = meta["params"]
    sd = dict(np.load(os.path.join(work, "weights.npz")))
    pcm = np.load(os.path.join(work, "q_pcm.npy"), mmap_mode="r")
    js = list(range(w, pcm.shape[0], W))
    bank = np.load(os.path.join(work, "bank.npy")) if os.path.exists(os.path.join(work, "bank.npy")) else None
    e32, e64, e64b = [], [], []
    for c0 in range(0, len(js), 8):                                     # 8 queries = 152 windows per call
        segs = np.concatenate([osg.segment(osg.pcm_to_mono(np.asarray(pcm[j])[:, None]), 8000, HOP) for j in js[c0:c0 + 8]])
        e32.append(oe.encode(om.melspec(segs, params), sd, params))
        if meta.get("f64", True):
            e64.append(oe.encode(om.melspec_f64(segs, params), sd, params, dtype=np.float64))
        if bank is not None:
            e64b.append(oe.encode(om.melspec_f64(segs, params, bank=bank), sd, params, dtype=np.float64))
    d = params["model"]["d"]
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros((0, d), dt)
    np.savez(os.path.join(work, "emb_%d.npz" % w), j=np.asarray(js, np.int64), emb32=cat(e32, np.float32), emb64=cat(e64, np.float64),
             emb64_bank=cat(e64b, np.float64))


def run_embed(params, sd, q_pcm, workers=32, bank=None, f64=True):
    """q_pcm int16 [nq, samples] -> {'emb32' f32, 'emb64' f64[, 'emb64_bank' f64: float64 with the mel bank `bank`
    float32 [n_freqs, n_mels]]} [nq * windows, d] in window order."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="pfann_oracle_", dir=base)
    try:
        np.save(os.path.join(work, "q_pcm.npy"), np.ascontiguousarray(q_pcm, np.int16))
        if bank is not None:
            np.save(os.path.join(work, "bank.npy"), np.ascontiguousarray(bank, np.float32))
        np.savez(os.path.join(work, "weights.npz"), **{n: np.asarray(v) for n, v in sd.items()})
        json.dump({"params": params, "f64": bool(f64)}, open(os.path.join(work, "meta.json"), "w"))
        workers = max(1, min(workers, q_pcm.shape[0]))
        t1 = time.time()
        env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS=os.environ.get("PFANN_ORACLE_THREADS", "8"))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--embed-worker", work, str(w), str(workers)], env=env)
                 for w in range(workers)]
        rcs = [p.wait() for p in procs]
        if any(rcs):
            raise RuntimeError("oracle workers failed: %r" % rcs)
        wall = time.time() - t1
        parts = [np.load(os.path.join(work, "emb_%d.npz" % w)) for w in range(workers)]
    finally:
        shutil.rmtree(work, ignore_errors=True)
    nq = q_pcm.shape[0]
    nwin = sum(p["emb32"].shape[0] for p in parts) // nq
    out = {}
    for name in ("emb32",) + (("emb64",) if f64 else ()) + (("emb64_bank",) if bank is not None else ()):
        full = np.empty((nq, nwin, parts[0][name].shape[1]), parts[0][name].dtype)
        for p in parts:
            full[p["j"]] = p[name].reshape(len(p["j"]), nwin, -1)
        out[name] = full.reshape(nq * nwin, -1)
    out.update(wall_s=wall, workers=workers)
    return out


def run_taps(params, sd, q_pcm, seg, bank=None):
    """The 16 sub-layer activations (fp32 and float64), both log-mels and fingerprints of the windows `seg` (global
    window numbers query * QSEG + t), in this process."""
    import torch
    torch.set_num_threads(int(os.environ.get("PFANN_ORACLE_THREADS", "8")))
    from oracle import encoder as oe
    from oracle import melspec as om
    from oracle import segmenter as osg
    segs = np.stack([osg.segment(osg.pcm_to_mono(np.asarray(q_pcm[s // QSEG])[:, None]), 8000, HOP)[s % QSEG] for s in seg])
    m32, m64 = om.melspec(segs, params), om.melspec_f64(segs, params)
    t32, t64 = [], []
    out = {"mel32": m32, "mel64": m64, "emb32": oe.encode(m32, sd, params, taps=t32),
           "emb64": oe.encode(m64, sd, params, taps=t64, dtype=np.float64),
           "emb64_mel32": oe.encode(m32, sd, params, dtype=np.float64)}
    for i in range(16):
        out["tap32_%d" % i], out["tap64_%d" % i] = t32[i], t64[i]
    if bank is not None:
        tb = []
        out["mel64_bank"] = om.melspec_f64(segs, params, bank=bank)
        out["emb64_bank"] = oe.encode(out["mel64_bank"], sd, params, taps=tb, dtype=np.float64)
        for i in range(16):
            out["tap64_bank_%d" % i] = tb[i]
    return out


def song_worker(work, w, W):
    """The oracle as the BUILDER (reference builder.py:88-100): its share of the music list, file by file through the
    oracle's own WAV reader, segmenter, log-mel and encoder; unreadable files give 0 rows (musicdata.py:95-101)."""
    import torch
    torch.set_num_threads(int(os.environ.get("PFANN_ORACLE_THREADS", "8")))
    from oracle import encoder as oe
    from oracle import melspec as om
    from oracle import segmenter as osg
    meta = json.load(open(os.path.join(work, "meta.json")))
    params = meta["params"]
    sd = dict(np.load(os.path.join(work, "weights.npz")))
    music = json.load(open(os.path.join(work, "music.json")))
    bank = np.load(os.path.join(work, "bank.npy")) if os.path.exists(os.path.join(work, "bank.npy")) else None
    ids = list(range(w, len(music), W))
    counts, rows = [], []
    for c0 in range(0, len(ids), 4):                                  # 4 songs = 236 windows per call
        segs = [osg.load_segments(music[i], params) for i in ids[c0:c0 + 4]]
        counts += [x.shape[0] for x in segs]
        cat = np.concatenate(segs)
        if cat.shape[0]:
            rows.append(oe.encode(om.melspec(cat, params, bank=bank), sd, params))
    d = params["model"]["d"]
    np.savez(os.path.join(work, "songs_%d.npz" % w), ids=np.asarray(ids, np.int64), counts=np.asarray(counts, np.int64),
             rows=np.concatenate(rows) if rows else np.zeros((0, d), np.float32))


def _spawn(mode, work, workers):
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS=os.environ.get("PFANN_ORACLE_THREADS", "8"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), mode, work, str(w), str(workers)], env=env)
             for w in range(workers)]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise RuntimeError("oracle workers (%s) failed: %r" % (mode, rcs))


def run_files(params, sd, music, queries, k, workers=32, keep_song_scores=False, bank=None):
    """The whole reference pipeline from FILES, nothing shared with the product but the WAVs and the weights: the oracle
    builds its own database from `music` (list of WAV paths, list order = song ids), then answers `queries` (WAV paths)
    against it.  -> dict: 'db' f32 [N, d], 'key' int64 [n_songs] (rows per song), per-query arrays as run(), and
    'ss' f32 [n_queries, n_songs, 2] when keep_song_scores."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="pfann_oracle_", dir=base)
    try:
        np.savez(os.path.join(work, "weights.npz"), **{n: np.asarray(v) for n, v in sd.items()})
        json.dump({"params": params, "k": k, "hop_s": params["hop_size"], "keep_song_scores": bool(keep_song_scores), "batch_queries": 8},
                  open(os.path.join(work, "meta.json"), "w"))
        json.dump(list(music), open(os.path.join(work, "music.json"), "w"))
        if bank is not None:                  # the mel bank the oracle's front-end uses instead of its float64-built default
            np.save(os.path.join(work, "bank.npy"), np.ascontiguousarray(bank, np.float32))
        json.dump(list(queries), open(os.path.join(work, "q_files.json"), "w"))
        t0 = time.time()
        W = max(1, min(workers, len(music)))
        _spawn("--song-worker", work, W)
        d = params["model"]["d"]
        key = np.zeros(len(music), np.int64)
        parts = [np.load(os.path.join(work, "songs_%d.npz" % w)) for w in range(W)]
        for p in parts:
            key[p["ids"]] = p["counts"]
        song_pos = np.pad(np.cumsum(key), (1, 0))
        db = np.empty((int(song_pos[-1]), d), np.float32)
        for p in parts:
            r0 = 0
            for i, c in zip(p["ids"], p["counts"]):
                db[song_pos[i]:song_pos[i] + c] = p["rows"][r0:r0 + c]
                r0 += c
        t_build = time.time() - t0
        np.save(os.path.join(work, "db.npy"), db)
        np.save(os.path.join(work, "song_pos.npy"), song_pos)
        t1 = time.time()
        W = max(1, min(workers, len(queries)))
        _spawn("--worker", work, W)
        t_query = time.time() - t1
        parts = [np.load(os.path.join(work, "res_%d.npz" % w)) for w in range(W)]
        order = np.argsort(np.concatenate([p["j"] for p in parts]))
        # (all queries of a population have the same length here, so the per-row arrays stack)
        out = {name: np.concatenate([p[name] for p in parts])[order]
               for name in ("song", "sec", "score", "runner_up", "kth", "next", "labels")}
        if keep_song_scores:
            ss = np.concatenate([np.load(os.path.join(work, "ss_%d.npy" % w)) for w in range(W)])
            out["ss"] = ss[order]
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out.update(db=db, key=key, song_pos=song_pos, build_s=t_build, query_s=t_query, workers=workers)
    return out


def run(params, sd, db, song_pos, q_pcm, k, workers=16, q_emb_gpu=None, keep=False, batch_queries=1):
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
        json.dump({"params": params, "k": k, "hop_s": params["hop_size"], "batch_queries": int(batch_queries)},
                  open(os.path.join(work, "meta.json"), "w"))
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
    elif len(sys.argv) > 1 and sys.argv[1] == "--song-worker":
        song_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    elif len(sys.argv) > 1 and sys.argv[1] == "--embed-worker":
        embed_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
