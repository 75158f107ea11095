#!/usr/bin/env python
"""Full-population decision parity (north_star: "top-1 match / segment-offset decisions exactly on the
same queries"): EVERY query of a BASELINE config goes through the CPU oracle -- segmenter -> log-mel -> encoder (torch
CPU fp32) -> exact top-k (BLAS) -> the reference's Python-path sequence matcher -- next to the GPU product path, and
every query is compared: fingerprints within 1e-4, (song, offset) identical, score within 1e-5.  A flip is classified
from the oracle's own numbers:
    boundary tie  : the two top-k label sets differ only by rows whose scores lie within 1e-5 of the oracle's k-th score
                    (5e-6 fingerprint differences move which of two equal-scoring rows is the k-th)
    alignment tie : same candidates, and the two decisions' sequence scores lie within 1e-6 of each other in the oracle
    bug           : anything else
The oracle runs in W worker PROCESSES (tools/oracle_pool.py: each 8 torch threads; the oracle anti-scales beyond that), fed
through .npy files in a tmpfs directory; nothing under oracle/ is imported by the product.

    python tools/decision_parity.py --songs 10000 --queries 2000 --snr 0 [--workers 16] [--out profiles/r4/x.json]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (REPO, os.path.join(REPO, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
SEG, QSEG, HOP = 59, 19, 4000


def run(n_songs, n_queries, snr, workers=16, config="default", plan=9728, max_batch=9728, log=print, keep=False, shards=1,
        prebuilt=None, state=None, oracle_queries=None):
    """prebuilt: {"eng": Engine (plan pinned by the caller), "sd": weights, "shard": device float32 [>= n_songs*59, d]} of a
    database some other test of the session has already embedded (the GPU suite builds each one once); state: a dict that
    receives the GPU-side arrays (q_pcm, emb, labels, res, index, shard) for further checks on the same workload;
    oracle_queries: the CPU oracle answers only the first so many queries (the GPU answers all n_queries) and the
    comparison is made on those."""
    import torch
    from pfann_amd import synth
    from pfann_amd.builder import embed_files
    from pfann_amd.database import DeviceIndex
    from pfann_amd.engine import Engine
    params = json.load(open(os.path.join(REPO, "configs", config + ".json")))
    d, k = params["model"]["d"], params["indexer"]["top_k"]
    if prebuilt is not None:
        eng, sd, shard = prebuilt["eng"], prebuilt["sd"], prebuilt["shard"][: n_songs * SEG]
        dev = eng.device
        t0 = time.time()
    else:
        try:                    # calibrated output bias (an untrained network's fingerprints then spread over the sphere); configs
            sd = synth.make_state_dict_calibrated(params, seed=123)      # without constants fall back to the raw seeded weights,
        except KeyError:                                                 # whose fingerprints all but coincide (a degenerate db)
            sd = synth.make_state_dict(params, seed=123)
        eng = Engine(params, 0, max_batch=max_batch)
        eng.load_state_dict(sd)
        eng.set_plan_batch(plan)
        dev = eng.device
        t0 = time.time()

        class Pcm:
            def __init__(self, ids, pcm):
                self.files, self.pcm = ["song %d" % i for i in ids], pcm

            def load_pcm(self, i):
                return self.pcm[i]

            def __len__(self):
                return len(self.files)
        shard = torch.empty((n_songs * SEG, d), device=dev, dtype=torch.float32)
        for c0 in range(0, n_songs, 256):
            ids = list(range(c0, min(c0 + 256, n_songs)))
            pcm = synth.make_songs_torch(ids, 30.0, device=dev)
            for i, n_seg, e in embed_files(eng, Pcm(ids, pcm), HOP, batch_windows=max_batch):
                shard[ids[i] * SEG:(ids[i] + 1) * SEG] = e
    song_pos = np.arange(n_songs + 1, dtype=np.int64) * SEG
    q_song = [int((j * 7919 + 13) % n_songs) for j in range(n_queries)]
    pcms, embs = [], []
    for c0 in range(0, n_queries, 256):
        ids = q_song[c0:c0 + 256]
        qp, _ = synth.make_queries_torch(synth.make_songs_torch(ids, 30.0, device=dev), list(range(c0, c0 + len(ids))), 10.0, snr)
        pcms.append(qp)
    q_pcm = torch.cat(pcms)
    # the GPU decisions the way the matcher CLI makes them: launch groups of max_batch windows, plan pinned
    # shards > 1: the SHARDED protocol of pfann_amd/dist.py with all N song shards held as N handles on this one GPU and
    # the collectives replaced by stacking (two-phase bounded search per shard, merge of the shard lists, owner-side
    # matcher on every shard, 128-bit winner keys, device pick) -- BASELINE config 4's "sharded 8 ways", decision by decision
    from pfann_amd.dist import shard_songs
    handles = []
    for lo, hi in (shard_songs(song_pos, shards) if shards > 1 else [(0, n_songs)]):
        ix = DeviceIndex(d, 0)
        ix.load(shard[int(song_pos[lo]):int(song_pos[hi])], song_pos, int(song_pos[lo]), song_range=(lo, hi))
        handles.append(ix)
    index = handles[0]
    per = max(1, max_batch // QSEG)
    res, labels = [], []
    for c0 in range(0, n_queries, per):
        qp = q_pcm[c0:c0 + per]
        nq = qp.shape[0]
        starts = (torch.arange(nq, device=dev)[:, None] * qp.shape[1] + torch.arange(QSEG, device=dev)[None, :] * HOP).reshape(-1)
        e = eng.embed_windows(eng.pcm16_to_mono(qp.reshape(-1)), starts)
        qs, ql = np.arange(nq, dtype=np.int64) * QSEG, np.full(nq, QSEG, np.int32)
        if shards > 1:
            m = min(k, 2 * k // shards + 8)
            cands = torch.stack([ix.search_bound(e, k, m) for ix in handles])                 # "all-gather"
            lists = [ix.search_bounded(e, k, ix.reduce_bound(cands, k)) for ix in handles]
            D, I = index.merge_lists(torch.stack([a for a, _ in lists]), torch.stack([b for _, b in lists]), k)
            keys = torch.stack([ix.pack_winner_keys(ix.match(e, I, qs, ql, 1, 0.0, 0, True, False, to_host=False)[0]) for ix in handles])
            r = index.pick_winner(keys)
        else:
            D, I = index.search(e, k)
            r, _ = index.match(e, I, qs, ql)
        embs.append(e), res.append(r), labels.append(I.cpu().numpy())
    res = np.concatenate(res)
    labels = np.concatenate(labels).reshape(n_queries, QSEG, k)
    emb_gpu = torch.cat(embs).cpu().numpy()
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    log("decision_parity: GPU side (db %d rows + %d queries) %.1f s" % (n_songs * SEG, n_queries, t_gpu))

    import oracle_pool
    n_all, res_all = n_queries, res
    if oracle_queries is not None and oracle_queries < n_queries:
        n_queries = int(oracle_queries)
        res, labels, emb_gpu = res[:n_queries], labels[:n_queries], emb_gpu[:n_queries * QSEG]
    pool = oracle_pool.run(params, sd, shard.cpu().numpy(), song_pos, q_pcm[:n_queries].cpu().numpy(), k, workers=workers,
                           q_emb_gpu=emb_gpu, keep=keep, batch_queries=8)
    if state is not None:
        state.update(q_pcm=q_pcm, emb=torch.cat(embs), labels=labels, res=res_all, index=index, shard=shard, song_pos=song_pos,
                     q_song=np.asarray(q_song), params=params, sd=sd, eng=eng, oracle=pool)
    t_cpu = pool["wall_s"]
    o_song, o_sec, o_score, emb_err = pool["song"], pool["sec"], pool["score"], pool["emb_err"]
    kth, nxt, runner, o_lab = pool["kth"], pool["next"], pool["runner_up"], pool["labels"]
    log("decision_parity: oracle side %.1f s on %d processes" % (t_cpu, workers))
    g_song, g_sec, g_score = res["song"].astype(np.int64), res["offset"] * params["hop_size"], res["score"]
    same = (g_song == o_song) & (g_sec == o_sec)
    flips = []
    for j in np.nonzero(~same)[0]:
        sets_equal = all(set(labels[j, t].tolist()) == set(o_lab[j, t].tolist()) for t in range(QSEG))
        # rows that are in one list and not the other must score within 1e-5 of the oracle's k-th score
        gap_k = float(np.min(kth[j] - nxt[j]))
        align_gap = float(o_score[j] - runner[j])
        if not sets_equal and gap_k <= 1e-5:
            kind = "boundary tie"
        elif abs(float(g_score[j]) - float(o_score[j])) <= 1e-6 or align_gap <= 1e-6:
            kind = "alignment tie"
        else:
            kind = "bug"
        flips.append({"query": int(j), "gpu": [int(g_song[j]), float(g_sec[j]), float(g_score[j])],
                      "oracle": [int(o_song[j]), float(o_sec[j]), float(o_score[j])], "topk_sets_equal": bool(sets_equal),
                      "min_gap_kth_vs_next": gap_k, "oracle_best_minus_runner_up_song": align_gap, "class": kind})
    sets_differ = int(sum(1 for j in range(n_queries)
                          if any(set(labels[j, t].tolist()) != set(o_lab[j, t].tolist()) for t in range(QSEG))))
    hit = float(np.mean(g_song == np.asarray(q_song[:n_queries])))
    out = {"config": config, "db_songs": n_songs, "db_rows": n_songs * SEG, "queries": n_queries, "gpu_queries": n_all,
           "top1_hit_rate_gpu_all_queries": round(float(np.mean(res_all["song"].astype(np.int64) == np.asarray(q_song))), 4),
           "snr_db": snr, "top_k": k,
           "plan_batch": plan, "shards": shards, "identical_song_and_offset": int(same.sum()), "flips": flips,
           "bugs": int(sum(1 for f in flips if f["class"] == "bug")),
           "max_embedding_abs_diff": float(emb_err.max()), "embedding_tolerance": 1e-4,
           "max_score_abs_diff_where_decisions_agree": float(np.abs(g_score - o_score)[same].max()) if same.any() else None,
           "score_tolerance": 1e-5,
           "queries_whose_topk_label_sets_differ": sets_differ,
           "smallest_kth_minus_next_gap_over_all_rows": float(np.min(kth - nxt)),
           "top1_hit_rate_gpu": round(hit, 4), "top1_hit_rate_oracle": round(float(np.mean(o_song == np.asarray(q_song[:n_queries]))), 4),
           "gpu_side_s": round(t_gpu, 1), "oracle_side_s": round(t_cpu, 1), "oracle_processes": workers,
           "oracle_threads_per_process": int(os.environ.get("PFANN_ORACLE_THREADS", "8")), "host_cpus": os.cpu_count(),
           "oracle": "oracle/{segmenter,melspec,encoder,search,seqscore}.py (python path, database.py:117-166) against the same "
                     "GPU-built db"}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=10000)
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--snr", type=float, default=0.0)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--plan", type=int, default=9728)
    ap.add_argument("--config", default="default", help="configs/<name>.json (default, seg, n640d64)")
    ap.add_argument("--shards", type=int, default=1, help="> 1: the sharded protocol over N handles on this GPU (no collectives)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = run(a.songs, a.queries, a.snr, a.workers, config=a.config, plan=a.plan, shards=a.shards, log=lambda *x: print(*x, file=sys.stderr, flush=True))
    print(json.dumps({k: v for k, v in r.items() if k != "flips"}), "flips:", json.dumps(r["flips"])[:3000])
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(r, open(a.out, "w"), indent=1)
