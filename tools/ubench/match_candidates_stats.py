"""What the sequence matcher's candidate lists look like on the bench workload (real 1 M-row db of embedded synthetic
songs, 512 ten-second SNR-0 queries): candidates per query, distinct songs, and how many db rows a query's candidates
touch when every song's offset runs are read ONCE (a run of alignments off, off+1, ... shares all but one row between
neighbours) against the rows the one-wave-per-candidate gather reads (19 per candidate).  Sizes VERDICT r4 item 5."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from pfann_amd import synth                      # noqa: E402
from pfann_amd.builder import embed_files        # noqa: E402
from pfann_amd.database import DeviceIndex       # noqa: E402
from pfann_amd.engine import Engine              # noqa: E402

SEG, QSEG, HOP = 59, 19, 4000
n_songs = int(os.environ.get("SONGS", "16950"))
nq = int(os.environ.get("QUERIES", "512"))
params = json.load(open(os.path.join(REPO, "configs", "default.json")))
d, k = 128, params["indexer"]["top_k"]
sd = synth.make_state_dict_calibrated(params, seed=123)
eng = Engine(params, 0, max_batch=9728)
eng.load_state_dict(sd)
dev = eng.device


class Pcm:
    def __init__(self, ids, pcm):
        self.files, self.pcm = ["song %d" % i for i in ids], pcm

    def load_pcm(self, i):
        return self.pcm[i]

    def __len__(self):
        return len(self.files)


db = torch.empty((n_songs * SEG, d), device=dev)
for c0 in range(0, n_songs, 256):
    ids = list(range(c0, min(c0 + 256, n_songs)))
    for i, n_seg, e in embed_files(eng, Pcm(ids, synth.make_songs_torch(ids, 30.0, device=dev)), HOP, batch_windows=9728):
        db[ids[i] * SEG:(ids[i] + 1) * SEG] = e
pos = np.arange(n_songs + 1, dtype=np.int64) * SEG
q_song = [int((j * 7919 + 13) % n_songs) for j in range(nq)]
qp, _ = synth.make_queries_torch(synth.make_songs_torch(q_song, 30.0, device=dev), list(range(nq)), 10.0, 0.0)
starts = (torch.arange(nq, device=dev)[:, None] * qp.shape[1] + torch.arange(QSEG, device=dev)[None, :] * HOP).reshape(-1)
emb = eng.embed_windows(eng.pcm16_to_mono(qp.reshape(-1)), starts)
ix = DeviceIndex(d, 0)
ix.load(db, pos, 0)
D, I = ix.search(emb, k)
qs, ql = np.arange(nq, dtype=np.int64) * QSEG, np.full(nq, QSEG, np.int32)
for _ in range(3):
    res, _ = ix.match(emb, I, qs, ql)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    res, _ = ix.match(emb, I, qs, ql, to_host=False)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / 10 * 1e3
lab = I.cpu().numpy().reshape(nq, QSEG, k)
tt = np.arange(QSEG)[None, :, None]
song = lab // SEG
off = lab - song * SEG - tt
key = (song.astype(np.int64) << 20) + (off + 100)
n_c, n_s, rows_now, rows_run, rows_song, runs, longest = [], [], [], [], [], [], []
for j in range(nq):
    u = np.unique(key[j].reshape(-1))
    s, o = u >> 20, (u & 0xFFFFF) - 100
    n_c.append(len(u))
    n_s.append(len(np.unique(s)))
    rows_now.append(len(u) * QSEG)
    brk = np.nonzero((s[1:] != s[:-1]) | (o[1:] - o[:-1] >= QSEG))[0] + 1        # runs: same song, windows overlapping
    first = np.concatenate([[0], brk])
    last = np.concatenate([brk - 1, [len(u) - 1]])
    rows_run.append(int(np.sum(np.minimum(o[last] - o[first] + QSEG, SEG + QSEG))))
    runs.append(len(first))
    longest.append(int((last - first + 1).max()))
    rows_song.append(n_s[-1] * SEG)                                           # whole songs staged
out = {"db_rows": n_songs * SEG, "queries": nq, "match_ms_per_launch": round(ms, 3),
       "candidates_per_query_mean": float(np.mean(n_c)), "candidates_per_query_max": int(np.max(n_c)),
       "distinct_songs_per_query_mean": float(np.mean(n_s)),
       "runs_per_query_mean": float(np.mean(runs)), "longest_run_mean": float(np.mean(longest)),
       "db_rows_read_per_query_now": float(np.mean(rows_now)),
       "db_rows_per_query_runs_read_once": float(np.mean(rows_run)),
       "db_rows_per_query_whole_songs": float(np.mean(rows_song)),
       "bytes_per_launch_now_GB": float(np.sum(rows_now)) * 512 / 1e9,
       "bytes_per_launch_runs_GB": float(np.sum(rows_run)) * 512 / 1e9}
print(json.dumps(out, indent=1))
