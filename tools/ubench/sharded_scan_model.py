"""Scan throughput of the song-sharded search at N ranks, measured on ONE GPU (tuning aid; no collectives, so an upper
bound): the real 1 M-segment database and the bench's 4096 real queries (77,824 rows); all N shards are held as N
handles, and RANK 0's share of the real protocol is timed kernel by kernel (HIP events around every launch):
    sampled pass + bound candidates (pfann_search_bound, per 16,384-row chunk) -> [all ranks' candidates] -> k-th largest of
    the union (pfann_topk_merge) -> bounded full pass + selects (pfann_search_topk_bounded) -> merge of the N shard lists
    for this rank's query slice (pfann_topk_merge on [Q/N, N*k])
North star: ">= 6x scan throughput at 8 GPUs" = t(1) / t(N) with t = scan-kernel time per step of the slowest rank (the job is the same at every N).
    python tools/ubench/sharded_scan_model.py [queries=4096]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from pfann_amd import lib as plib                 # noqa: E402
from pfann_amd import synth                       # noqa: E402
from pfann_amd.builder import embed_files          # noqa: E402
from pfann_amd.database import DeviceIndex         # noqa: E402
from pfann_amd.dist import shard_songs             # noqa: E402
from pfann_amd.engine import Engine                # noqa: E402


def main():
    nq_q = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    params = json.load(open(os.path.join(REPO, "configs", "default.json")))
    eng = Engine(params, 0, max_batch=9728)
    eng.load_state_dict(synth.make_state_dict_calibrated(params, seed=123))
    dev, d, k, n_songs = eng.device, 128, 100, 16950
    lib = plib.load()

    class Pcm:
        def __init__(self, ids, pcm):
            self.files, self.pcm = ["s%d" % i for i in ids], pcm

        def load_pcm(self, i):
            return self.pcm[i]

        def __len__(self):
            return len(self.files)
    db = torch.empty((n_songs * 59, d), device=dev)
    for c0 in range(0, n_songs, 656):
        ids = list(range(c0, min(c0 + 656, n_songs)))
        pcm = synth.make_songs_torch(ids, 30.0, device=dev)
        for i, n_seg, e in embed_files(eng, Pcm(ids, pcm), 4000, batch_windows=9728):
            db[ids[i] * 59:(ids[i] + 1) * 59] = e
    song_pos = np.arange(n_songs + 1, dtype=np.int64) * 59
    embs = []
    for c0 in range(0, nq_q, 512):
        js = list(range(c0, min(c0 + 512, nq_q)))
        qp, _ = synth.make_queries_torch(synth.make_songs_torch([int((j * 7919 + 13) % n_songs) for j in js], 30.0, device=dev), js, 10.0, 0.0)
        starts = (torch.arange(len(js), device=dev)[:, None] * qp.shape[1] + torch.arange(19, device=dev)[None, :] * 4000).reshape(-1)
        embs.append(eng.embed_windows(eng.pcm16_to_mono(qp.reshape(-1)), starts))
    q = torch.cat(embs).contiguous()
    Q = q.shape[0]
    del eng
    torch.cuda.empty_cache()

    def tags():
        buf = ctypes.create_string_buffer(4096)
        lib.pfann_prof_tags(buf, 4096)
        out = {}
        for t in buf.value.decode().split(","):
            if t:
                c = ctypes.c_int64(0)
                out[t] = lib.pfann_prof_elapsed_ms(t.encode(), ctypes.byref(c))
        return out

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        lib.pfann_prof_reset()
        lib.pfann_prof_enable(1)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        lib.pfann_prof_enable(0)
        return {t: v / reps for t, v in tags().items()}
    rows = []
    t1 = None
    # the exact global labels every rank ends up with (for the owner-side sequence matcher below)
    full = DeviceIndex(d, 0)
    full.load(db, song_pos, 0)
    I_glob = torch.cat([full.search(q[c0:c0 + 16384].contiguous(), k)[1] for c0 in range(0, Q, 16384)])
    del full
    qstart = np.arange(nq_q, dtype=np.int64) * 19
    qlen = np.full(nq_q, 19, np.int32)
    for N in (1, 2, 4, 8):
        shards = []
        for lo, hi in shard_songs(song_pos, N):
            ix = DeviceIndex(d, 0)
            ix.load(db[song_pos[lo]:song_pos[hi]].contiguous(), song_pos, int(song_pos[lo]))
            shards.append(ix)
        s0 = shards[0]
        if N == 1:
            prof = timed(lambda: s0.search(q, k))
        else:
            m = min(k, 2 * k // N + 8)
            CH = s0.BOUND_CHUNK
            cands = [torch.stack([ix.search_bound(q[c0:c0 + CH], k, m) for ix in shards]) for c0 in range(0, Q, CH)]
            Qs = (Q + N - 1) // N
            # what the all-to-all hands rank 0: every shard's list for ITS query slice (labels ascending by shard)
            lists = []
            for ix in shards:
                Dl, Il = [], []
                for ci, c0 in enumerate(range(0, Q, CH)):
                    lb = ix.reduce_bound(cands[ci], k)
                    Dc, Ic = ix.search_bounded(q[c0:c0 + CH], k, lb)        # (q chunk must be the one search_bound saw: re-run it)
                    Dl.append(Dc), Il.append(Ic)
                lists.append((torch.cat(Dl)[:Qs], torch.cat(Il)[:Qs]))
            S = torch.stack([D for D, _ in lists]).contiguous()          # [N, Qs, k]: what the all-to-all delivers
            L = torch.stack([I for _, I in lists]).contiguous()

            def rank0():
                for ci, c0 in enumerate(range(0, Q, CH)):
                    qc = q[c0:c0 + CH]
                    s0.search_bound(qc, k, m)
                    lb = s0.reduce_bound(cands[ci], k)
                    s0.search_bounded(qc, k, lb)
                s0.merge_lists(S, L, k)
            prof = timed(rank0)
            # exactness of the protocol on this data: merged lists == the unsplit search
            if N == 8:
                full = DeviceIndex(d, 0)
                full.load(db, song_pos, 0)
                Dm, Im = s0.merge_lists(S, L, k)
                Df, If = full.search(q[:Qs].contiguous(), k)
                assert bool((Dm == Df).all()) and bool((torch.sort(Im, 1).values == torch.sort(If, 1).values).all())
                del full
        # owner-side sequence matcher of rank 0: all queries' global lists, candidates of its own songs only
        mprof = timed(lambda: s0.match(q, I_glob, qstart, qlen, 1, 0.0, 0, N > 1, False, to_host=False))
        t_match = mprof.get("seq_match", 0.0)
        scan = {t: v for t, v in prof.items() if t.startswith("scan_topk") or t.startswith("topk_")}
        tot = sum(scan.values())
        if N == 1:
            t1 = tot
        rows.append("N=%d  scan kernels of rank 0: %.2f ms per %d-row step  -> scan throughput x%.2f (ideal x%d); seq_match %.2f ms   %s"
                    % (N, tot, Q, t1 / tot, N, t_match, ", ".join("%s %.2f" % (t, v) for t, v in sorted(scan.items(), key=lambda kv: -kv[1]))))
        print(rows[-1], flush=True)
        del shards, s0
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(REPO, "gpurun_out", "r5"), exist_ok=True)
    open(os.path.join(REPO, "gpurun_out", "r5", os.environ.get("MODEL_OUT", "sharded_scan_model.txt")), "w").write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
