#!/usr/bin/env python
"""Fully independent end-to-end parity at scale: the reference's matcher searches a database its OWN
builder wrote (builder.py:88-100,113-139 -> matcher.py:110-163), so here the oracle does too.

    product side : `python builder.py <music list> <db> <model dir>` then `python matcher.py <query list> <db> <result>`
                   as SUBPROCESSES on WAV files (tools/cli_bench.py)
    oracle side  : tools/oracle_pool.run_files on the SAME WAV files and weights, nothing else shared: the oracle reads
                   the files with its own reader, embeds every song (its own database), and answers every query against
                   that database through the reference's Python-path matcher

and the product's FILES are compared with it: `landmarkKey` exact; `embeddings` within 1e-4; every TSV answer and
`_detail.csv` (answer, time) identical, score within 1e-5; the `.bin` per-song block cell by cell (reported).  Unlike
tools/decision_parity.py (which hands the oracle the GPU-built rows, so db-side differences cancel), a fingerprint
difference on the database side is visible here.  A flip is classified from the oracle's own numbers, on the oracle's db:
    alignment tie : the oracle's sequence score of the PRODUCT's (song, offset) lies within 1e-6 of its own best
    boundary tie  : the product's alignment is not among the oracle's candidates, and one of its rows scores within 1e-5
                    of the oracle's k-th score in that query row (which of two equal rows is the 100th)
    bug           : anything else

    python tools/decision_parity_oracle_db.py --songs 2000 --queries 2000 --out profiles/r5/decision_parity_oracle_db.json
"""
import argparse
import csv
import json
import os
import shutil
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (REPO, os.path.join(REPO, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def classify(j, g_song, g_off, pool, params, sd, queries, k):
    """Re-run the oracle for query j in this process and place the product's alignment in ITS numbers."""
    from oracle import encoder as oe
    from oracle import melspec as om
    from oracle import segmenter as osg
    db, song_pos = pool["db"], pool["song_pos"]
    e = oe.encode(om.melspec(osg.load_segments(queries[j], params), params), sd, params)
    nq = e.shape[0]
    start, slen = int(song_pos[g_song]), int(song_pos[g_song + 1] - song_pos[g_song])
    vec = np.zeros_like(e)
    rows = []
    for i in range(nq):
        if 0 <= g_off + i < slen:
            vec[i] = db[start + g_off + i]
            rows.append((i, start + g_off + i))
    sg = np.dot(vec.flatten(), e.flatten()).item() / nq                      # database.py:157
    o_score = float(pool["score"][j])
    labels = pool["labels"][j]
    in_cands = any(lab in set(labels[i].tolist()) for i, lab in rows)
    near_kth = min((abs(float(e[i] @ db[lab]) - float(pool["kth"][j][i])) for i, lab in rows), default=np.inf)
    if abs(sg - o_score) <= 1e-6:
        kind = "alignment tie"
    elif not in_cands and near_kth <= 1e-5:
        kind = "boundary tie"
    else:
        kind = "bug"
    return {"class": kind, "oracle_score_of_product_alignment": sg, "oracle_best": o_score,
            "product_alignment_among_oracle_candidates": bool(in_cands), "closest_row_to_kth": float(near_kth)}


def run(n_songs=2000, n_queries=2000, snr=0.0, workers=32, log=print, bank="torchaudio", config="default"):
    """bank: which statement of the (unpinned) mel filter bank the oracle's front-end uses -- "torchaudio": its restatement of
    torchaudio's own float32 construction (oracle/melspec.mel_filterbank_torchaudio: what a reference installation computes);
    "float64": the bank written from the definition in float64 (oracle/melspec.mel_filterbank).  The two are <= 3.8e-5 apart
    per weight; on clean songs that alone moves fingerprints by up to 1.9e-4 (profiles/r5/decision_parity_oracle_db_f64_bank.json)."""
    import cli_bench
    import oracle_pool
    from pfann_amd import synth
    from pfann_amd.utils import read_config
    params = read_config(os.path.join(REPO, "configs", config + ".json"))
    params["indexer"] = dict(params["indexer"], index_factory="Flat")
    k, d, hop_s = params["indexer"]["top_k"], params["model"]["d"], params["hop_size"]
    sd = synth.make_state_dict_calibrated(params, seed=123)
    t0 = time.time()
    cb = cli_bench.run(n_songs, n_queries, snr, keep=True, log=log, config=config)
    if "skipped" in cb:
        return cb
    work = cb["workdir"]
    try:
        music = open(os.path.join(work, "music.txt")).read().split("\n")[:-1]
        queries = open(os.path.join(work, "queries.txt")).read().split("\n")[:-1]
        dbdir, result = os.path.join(work, "db"), os.path.join(work, "result.txt")
        t_cli = time.time() - t0
        keep_ss = n_songs * n_queries * 8 <= (1 << 30)
        from oracle import melspec as om
        bank_arr = None if bank == "float64" else om.mel_filterbank_torchaudio(
            params["sample_rate"], params["stft_n"], params["n_mels"], params["f_min"], params["f_max"], params.get("naf_mode", False)).numpy()
        pool = oracle_pool.run_files(params, sd, music, queries, k, workers=workers, keep_song_scores=keep_ss, bank=bank_arr)
        log("decision_parity_oracle_db: oracle built %d rows in %.1f s, answered %d queries in %.1f s (%d processes)" %
            (pool["db"].shape[0], pool["build_s"], n_queries, pool["query_s"], workers))
        # ---- the database files
        key = np.fromfile(os.path.join(dbdir, "landmarkKey"), dtype=np.int32)
        emb = np.fromfile(os.path.join(dbdir, "embeddings"), dtype=np.float32).reshape(-1, d)
        key_equal = bool(np.array_equal(key.astype(np.int64), pool["key"]))
        emb_diff = np.abs(emb - pool["db"]).max(axis=1) if emb.shape == pool["db"].shape else None
        # ---- the result files
        tsv = [ln.rstrip("\n").split("\t") for ln in open(result, encoding="utf8")]
        detail = list(csv.reader(open(os.path.splitext(result)[0] + "_detail.csv", newline="")))[1:]
        name_to_id = {p: i for i, p in enumerate(music)}
        g_song = np.asarray([name_to_id.get(r[1], -1) for r in tsv], np.int64)
        g_sec = np.asarray([float(r[3]) for r in detail])
        g_score = np.asarray([float(r[2]) for r in detail])
        tsv_detail_agree = all(a[0] == b[0] == q and a[1] == b[1] for a, b, q in zip(tsv, detail, queries))
        o_song, o_sec, o_score = pool["song"], pool["sec"], pool["score"]
        same = (g_song == o_song) & (g_sec == o_sec)
        flips = []
        for j in np.nonzero(~same)[0]:
            c = classify(int(j), int(g_song[j]), int(round(g_sec[j] / hop_s)), pool, params, sd, queries, k)
            c.update(query=int(j), product=[int(g_song[j]), float(g_sec[j]), float(g_score[j])],
                     oracle=[int(o_song[j]), float(o_sec[j]), float(o_score[j])])
            flips.append(c)
        out = {"config": config + " (index_factory Flat)", "db_songs": n_songs, "db_rows": int(pool["db"].shape[0]), "queries": n_queries,
               "snr_db": snr, "top_k": k,
               "product": "builder.py + matcher.py as subprocesses on WAV files", "oracle": "oracle_pool.run_files: own reader, own "
               "database (every song embedded by oracle/encoder.py on the host), python-path matcher (database.py:117-166)",
               "oracle_mel_bank": "torchaudio's float32 construction, restated (oracle/melspec.mel_filterbank_torchaudio)" if bank != "float64"
               else "float64 from the definition (oracle/melspec.mel_filterbank)",
               "landmarkKey_equal": key_equal,
               "embeddings_rows": int(emb.shape[0]),
               "embeddings_max_abs_diff": float(emb_diff.max()) if emb_diff is not None else None,
               "embeddings_p99.9_row_max": float(np.quantile(emb_diff, 0.999)) if emb_diff is not None else None,
               "embedding_tolerance": 1e-4,
               "tsv_and_detail_csv_agree": bool(tsv_detail_agree),
               "identical_song_and_time": int(same.sum()), "flips": flips, "bugs": int(sum(1 for f in flips if f["class"] == "bug")),
               "max_score_abs_diff_where_decisions_agree": float(np.abs(g_score - o_score)[same].max()) if same.any() else None,
               "score_tolerance": 1e-5,
               "top1_hit_rate_product": cb["matcher"]["top1_hit_rate"],
               "top1_hit_rate_oracle": round(float(np.mean(o_song == np.asarray([int((j * 7919 + 13) % n_songs) for j in range(n_queries)]))), 4),
               "cli": {"builder": cb["builder"], "matcher": cb["matcher"]},
               "oracle_build_s": round(pool["build_s"], 1), "oracle_query_s": round(pool["query_s"], 1), "oracle_processes": workers,
               "oracle_threads_per_process": int(os.environ.get("PFANN_ORACLE_THREADS", "8")), "host_cpus": os.cpu_count(),
               "product_cli_s": round(t_cli, 1)}
        if keep_ss:
            # the per-song block: [n_queries, n_songs, 2] (best score of each song, its time) -- cells where the two sides
            # disagree come from rows at the boundary of a top-k list (a candidate one side has and the other has not)
            b = np.fromfile(result + ".bin", dtype=np.float32).reshape(n_queries, n_songs, 2)
            ss = pool["ss"]
            both = (b[..., 0] != 0) & (ss[..., 0] != 0)               # a cell nobody wrote is (0, 0); one only one side wrote holds
            t_diff = (b[..., 1] != ss[..., 1]) | (both != ((b[..., 0] != 0) | (ss[..., 0] != 0)))   # a candidate the other list lacked
            s_diff = np.abs(b[..., 0] - ss[..., 0]) > 2e-5
            out["bin_cells"] = int(b.shape[0] * b.shape[1])
            out["bin_cells_time_differs_or_written_by_one_side_only"] = int(t_diff.sum())
            out["bin_cells_score_differs_2e-5"] = int((s_diff & ~t_diff).sum())
            out["bin_max_score_diff_where_time_agrees"] = float(np.abs(b[..., 0] - ss[..., 0])[~t_diff].max())
        out["wall_s"] = round(time.time() - t0, 1)
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=2000)
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--snr", type=float, default=0.0)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--bank", default="torchaudio", choices=["torchaudio", "float64"])
    ap.add_argument("--config", default="default", help="configs/<name>.json: default, seg, n640d64")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = run(a.songs, a.queries, a.snr, a.workers, log=lambda *x: print(*x, file=sys.stderr, flush=True), bank=a.bank, config=a.config)
    print(json.dumps({k: v for k, v in r.items() if k not in ("flips", "cli")}, indent=1), "\nflips:", json.dumps(r.get("flips", []))[:3000])
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(r, open(a.out, "w"), indent=1)
