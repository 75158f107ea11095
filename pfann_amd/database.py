"""Host mirror of the reference's retrieval interface (database.py:74-195):
`Database(dir_for_db, indexer_params, hop_size)` and
`Database.query_embeddings(query) -> (score, (song_id, time_s), song_score[n_songs,2])`.

The fingerprints live in HBM; search (exact flat inner-product top-k) and the sequence
matcher run as HIP kernels (csrc/search.hip, csrc/rerank.hip).  `query_batch` exposes the
batched form the CLIs and bench use; `query_embeddings` keeps the reference's per-query
contract, python-path semantics (cpp_accelerate=False, database.py:12).
"""
import contextlib
import ctypes
import json
import os
import struct
import time

import numpy as np
import torch

from . import faissio
from . import lib as _l
from .utils import read_file_list


def song_pos_from_key(landmark_key):
    return np.pad(np.cumsum(np.asarray(landmark_key), dtype=np.int64), (1, 0))      # database.py:86


class DeviceIndex:
    """One shard of fingerprints on one GPU + its search / match kernels."""

    def __init__(self, d, device=0, storage="f32"):
        """storage "f32" (default; exact fp32 results) or "f16": only fp16 rows are kept and searched on the fp16
        matrix cores without fp32 re-scoring (half the footprint and scan bytes; approximate like faiss'
        useFloat16, database.py:101-104)."""
        _l.require_gpu()
        self.lib = _l.load()
        self.d = d
        self.device = torch.device("cuda", device)
        self.handle = self.lib.pfann_db_create(d, device)
        if not self.handle:
            raise _l.PfannError("pfann_db_create failed: " + _l.last_error())
        self.storage = storage
        _l.check(self.lib.pfann_db_set_storage(self.handle, {"f32": 0, "f16": 1}[storage]), "pfann_db_set_storage")
        self.ntotal = 0
        self.label_base = 0
        self.n_songs = 0
        self._small_args = {}
        self._host_res = None

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.pfann_db_destroy(h)

    def load(self, emb, song_pos, label_base=0, song_range=None):
        """emb: float32 [n, d] numpy (host) or torch cuda tensor; song_pos: GLOBAL int64 prefix sums.  song_range: the
        caller's own cut (song_lo, song_hi) of the song list when it made one (dist.shard_songs) -- songs without rows at
        a shard boundary belong to the side the cut says; by default the library derives the songs from the rows."""
        song_pos = np.ascontiguousarray(song_pos, dtype=np.int64)
        self.song_pos = song_pos
        self.n_songs = song_pos.shape[0] - 1
        if isinstance(emb, torch.Tensor) and emb.is_cuda:
            e = emb.to(torch.float32).contiguous()
            ptr, is_dev, n = e.data_ptr(), 1, e.shape[0]
        else:
            e = np.ascontiguousarray(emb.cpu().numpy() if isinstance(emb, torch.Tensor) else emb, np.float32)
            e = e.reshape(-1, self.d)
            ptr, is_dev, n = e.ctypes.data, 0, e.shape[0]
        _l.check(self.lib.pfann_db_load(self.handle, ptr, is_dev, n,
                                        song_pos.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                        self.n_songs, label_base), "pfann_db_load")
        self.ntotal = n
        self.label_base = label_base
        if song_range is not None:
            _l.check(self.lib.pfann_db_set_owned_songs(self.handle, int(song_range[0]), int(song_range[1])),
                     "pfann_db_set_owned_songs")

    def set_prefilter(self, on=True):
        """fp16 pre-filter of the batched scan (exact result either way) -> True if in use."""
        return bool(self.lib.pfann_db_set_prefilter(self.handle, 1 if on else 0))

    def _stream(self):
        return _l.current_stream_ptr(self.device)

    def search(self, q, k):
        """index.search(q, k): q torch cuda [nq, d] -> (D [nq,k] f32 desc, I [nq,k] int64) on device."""
        q = q.to(self.device, torch.float32).contiguous()
        nq = q.shape[0]
        D = torch.empty((nq, k), device=self.device, dtype=torch.float32)
        I = torch.empty((nq, k), device=self.device, dtype=torch.int64)
        if nq:
            _l.check(self.lib.pfann_search_topk(self.handle, q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(),
                                                self._stream()), "pfann_search_topk")
        return D, I

    BOUND_CHUNK = 16384        # query rows per pfann_search_bound / pfann_search_topk_bounded call

    def search_bound(self, q, k, m=1):
        """First half of a sharded search (<= BOUND_CHUNK rows): -> [nq, m] f32 on the device: per query row the m best
        sampled scores of THIS shard (m different real rows), each lowered to a bound of its exact score; -inf padded (all
        -inf where no sampled threshold exists).  reduce_bound over the ranks' tensors gives what search_bounded wants."""
        nq = q.shape[0]
        lb = torch.empty((nq, m), device=self.device, dtype=torch.float32)
        if nq:
            _l.check(self.lib.pfann_search_bound(self.handle, q.data_ptr(), nq, k, m, lb.data_ptr(), self._stream()),
                     "pfann_search_bound")
        return lb

    def reduce_bound(self, cands, k):
        """cands [n_ranks, nq, m] (all-gathered search_bound outputs) -> [nq]: the k-th largest of every row's union, a lower
        bound of the row's k-th best score over all shards (>= k different real rows reach it); -FLT_MAX when the union
        holds fewer than k finite values."""
        G, nq, m = cands.shape
        if G * m <= 1024 and nq:
            # one wavefront per row straight from the gathered layout (pfann_bound_reduce): a selection, not a sort
            c = cands.to(self.device, torch.float32).contiguous()
            lb = torch.empty((nq,), device=self.device, dtype=torch.float32)
            _l.check(self.lib.pfann_bound_reduce(self.handle, c.data_ptr(), G, nq, m, k, lb.data_ptr(), self._stream()),
                     "pfann_bound_reduce")
            return lb
        vals = cands.to(self.device).permute(1, 0, 2).reshape(nq, G * m).contiguous()
        if G * m < k:
            return torch.full((nq,), -3.4028234663852886e38, device=self.device, dtype=torch.float32)
        labels = torch.where(torch.isfinite(vals), torch.arange(G * m, device=self.device, dtype=torch.int64).expand(nq, -1),
                             torch.full_like(vals, -1, dtype=torch.int64)).contiguous()
        D, _ = self.merge_topk(vals, labels, k)
        return D[:, k - 1].contiguous()

    def search_bounded(self, q, k, lb):
        """Second half: (D, I) of this shard restricted to rows that can be in the global top-k (padded with
        -FLT_MAX / -1); q must be the very tensor search_bound was given."""
        nq = q.shape[0]
        D = torch.empty((nq, k), device=self.device, dtype=torch.float32)
        I = torch.empty((nq, k), device=self.device, dtype=torch.int64)
        if nq:
            _l.check(self.lib.pfann_search_topk_bounded(self.handle, q.data_ptr(), nq, k, lb.contiguous().data_ptr(),
                                                        D.data_ptr(), I.data_ptr(), self._stream()),
                     "pfann_search_topk_bounded")
        return D, I

    def merge_lists(self, Dl, Il, k):
        """Dl / Il [G, nq, k] (every shard's list for these query rows, as the all-to-all delivers them) -> exact top-k of
        the union (D [nq, k] descending, I), ties to the lower shard: the merge of ShardedIndex.search_global."""
        G, nq, kk = Dl.shape
        if kk == k and k <= 128 and G * k <= 1024:
            Dl, Il = Dl.to(self.device, torch.float32).contiguous(), Il.to(self.device, torch.int64).contiguous()
            D = torch.empty((nq, k), device=self.device, dtype=torch.float32)
            I = torch.empty((nq, k), device=self.device, dtype=torch.int64)
            if nq:
                _l.check(self.lib.pfann_topk_merge_lists(self.handle, Dl.data_ptr(), Il.data_ptr(), G, nq, k, D.data_ptr(),
                                                         I.data_ptr(), self._stream()), "pfann_topk_merge_lists")
            return D, I
        S = Dl.permute(1, 0, 2).reshape(nq, G * kk).contiguous()            # shard-major: ascending labels on ties
        L = Il.permute(1, 0, 2).reshape(nq, G * kk).contiguous()
        return self.merge_topk(S, L, k)

    def merge_topk(self, S, L, k):
        nq, m = S.shape
        D = torch.empty((nq, k), device=self.device, dtype=torch.float32)
        I = torch.empty((nq, k), device=self.device, dtype=torch.int64)
        if nq:
            _l.check(self.lib.pfann_topk_merge(self.handle, S.contiguous().data_ptr(), L.contiguous().data_ptr(),
                                               nq, m, k, D.data_ptr(), I.data_ptr(), self._stream()),
                     "pfann_topk_merge")
        return D, I

    RESULT_DTYPE = np.dtype([("song", "<i4"), ("offset", "<i4"), ("shift", "<i4"), ("n_cand", "<i4"), ("score", "<f8")])

    def results_to_host(self, res_dev):
        """device results (uint8 [nQ, 24]) -> structured numpy array: the ONE device-to-host copy of a step."""
        out = np.frombuffer(res_dev.cpu().numpy().tobytes(), dtype=self.RESULT_DTYPE)
        if (out["song"] == -2).any():
            raise _l.PfannError("matcher refused a query (candidate buffer sizing error)")
        return out

    def pack_winner_keys(self, res_dev):
        """-> int64 [nQ, 2] device tensor of 128-bit keys (bit patterns of two uint64), see pfann_match_pack."""
        nQ = res_dev.shape[0]
        keys = torch.empty((nQ, 2), device=self.device, dtype=torch.int64)
        _l.check(self.lib.pfann_match_pack(self.handle, res_dev.data_ptr(), nQ, keys.data_ptr(), self._stream()), "pfann_match_pack")
        return keys

    def pick_winner(self, all_keys, to_host=True):
        """all_keys int64 [G, nQ, 2] (all-gathered) -> structured array of the winners (one D2H), or with to_host=False
        the device tensor of results (results_to_host turns it into the array later)."""
        all_keys = all_keys.to(self.device).contiguous()
        G, nQ = all_keys.shape[0], all_keys.shape[1]
        out = torch.empty((nQ, ctypes.sizeof(_l.MatchResult)), device=self.device, dtype=torch.uint8)
        _l.check(self.lib.pfann_match_pick(self.handle, all_keys.data_ptr(), G, nQ, out.data_ptr(), self._stream()), "pfann_match_pick")
        return self.results_to_host(out) if to_host else out

    def owned_songs(self):
        """-> (song_lo, song_hi): the songs whose rows all live in this shard"""
        lo, hi = ctypes.c_int(0), ctypes.c_int(0)
        self.lib.pfann_db_owned_songs(self.handle, ctypes.byref(lo), ctypes.byref(hi))
        return lo.value, hi.value

    def song_scores_to_seconds(self, ss, fsm, hop_size, native_path=False):
        """in place: the alignment slot of every (score, alignment) pair of a song_scores block, fine frames -> seconds"""
        if ss is not None and ss.numel():
            _l.check(self.lib.pfann_song_scores_to_seconds(self.handle, ss.data_ptr(), ss.numel() // 2, int(fsm), float(hop_size),
                                                           1 if native_path else 0, self._stream()),
                     "pfann_song_scores_to_seconds")
        return ss

    def match(self, q, labels, qstart, qlen, fsm=1, alpha=0.0, mode=0, only_owned=False, want_song_scores=False,
              to_host=True, owned_block=False):
        """Sequence matcher for nQ queries.  Returns (results structured array -- or, with to_host=False, the device
        tensor of results --, song_scores or None).  owned_block (with only_owned): song_scores is [nQ, owned songs, 2],
        this shard's columns of the score matrix, instead of [nQ, n_songs, 2]."""
        q = q.to(self.device, torch.float32).contiguous()
        labels = labels.to(self.device, torch.int64).contiguous()
        qs_np = np.ascontiguousarray(qstart, dtype=np.int64)
        ql_np = np.ascontiguousarray(qlen, dtype=np.int32)
        nQ = int(ql_np.shape[0])
        if nQ <= 4:
            # the one-query regime calls with the same tiny (qstart, qlen) over and over: keep their device copies
            key = (qs_np.tobytes(), ql_np.tobytes())
            hit = self._small_args.get(key)
            if hit is None:
                if len(self._small_args) > 64:
                    self._small_args.clear()
                hit = (_l.upload_async(qs_np, self.device, np.int64), _l.upload_async(ql_np, self.device, np.int32))
                self._small_args[key] = hit
            qs, ql = hit
        else:
            qs = _l.upload_async(qs_np, self.device, np.int64)
            ql = _l.upload_async(ql_np, self.device, np.int32)
        k = labels.shape[1]
        rsz = ctypes.sizeof(_l.MatchResult)
        host_res = None
        if to_host and 0 < nQ <= 64:
            # few queries: the kernel writes its 24-byte results straight into pinned (device-mapped) host memory, so the
            # answer is on the host when the stream has drained -- no device-to-host copy call on the latency path
            if self._host_res is None:
                self._host_res = torch.empty((64, rsz), dtype=torch.uint8, pin_memory=True)
            host_res = self._host_res[:nQ]
            res = host_res
        else:
            res = torch.empty((nQ, rsz), device=self.device, dtype=torch.uint8)
        ss = None
        if want_song_scores:
            lo, hi = self.owned_songs() if owned_block else (0, self.n_songs)
            ss = torch.zeros((nQ, hi - lo, 2), device=self.device, dtype=torch.float32)
        if nQ:
            _l.check(self.lib.pfann_match(self.handle, q.data_ptr(), labels.data_ptr(), k, qs.data_ptr(),
                                          ql.data_ptr(), nQ, int(ql_np.max()), fsm, float(alpha), mode,
                                          (1 if only_owned else 0) | (2 if owned_block else 0), res.data_ptr(),
                                          ss.data_ptr() if ss is not None else None, self._stream()),
                     "pfann_match")
        if not to_host:
            return res, ss
        if host_res is not None:
            torch.cuda.current_stream(self.device).synchronize()
            out = np.frombuffer(host_res.numpy().tobytes(), dtype=self.RESULT_DTYPE)
            if (out["song"] == -2).any():
                raise _l.PfannError("matcher refused a query (candidate buffer sizing error)")
            return out, ss
        return self.results_to_host(res), ss


def _fine_to_time(fine, fsm, hop_size):
    """fine = t*fsm - shift  ->  (t - shift/fsm) * hop_size, as database.py:148 computes it."""
    fine = np.asarray(fine, dtype=np.int64)
    shift = (-fine) % fsm
    t = (fine + shift) // fsm
    return (t - shift / fsm) * hop_size


# The reference's module-level switch (database.py:12): False = the Python-path semantics of query_embeddings_base (the
# default), True = those of query_embeddings_cpp / cpp/seqscore.cpp (fp32 divide, score_alpha honoured, score and time read
# back from the per-song block).  PFANN_CPP_ACCELERATE=1 sets it for the CLIs.  Either way the work runs in the HIP kernels.
cpp_accelerate = os.environ.get("PFANN_CPP_ACCELERATE", "0") not in ("0", "")


class LazyLaunches:
    """[(j0, j1)] cuts + launch(j0, j1) -> iterable of (j0, j1, launch result) with the first launch made at once and
    every later one made when its predecessor is handed out (so: one ahead of what the consumer is reading back)."""

    def __init__(self, cuts, launch):
        self._cuts, self._launch, self._next, self._ready = list(cuts), launch, 0, []
        self.max_in_flight = 0                       # (for the tests: launched and not yet handed out + the one handed out)
        if self._cuts:
            self._start_one()

    def _start_one(self):
        j0, j1 = self._cuts[self._next]
        self._next += 1
        self._ready.append((j0, j1, self._launch(j0, j1)))

    def __len__(self):
        return len(self._cuts)

    def __iter__(self):
        while self._ready or self._next < len(self._cuts):
            if not self._ready:
                self._start_one()
            cur = self._ready.pop(0)
            if self._next < len(self._cuts):
                self._start_one()                    # the successor is in flight before `cur` is read back
            self.max_in_flight = max(self.max_in_flight, 1 + len(self._ready))
            yield cur


class Database:
    def __init__(self, dir_for_db, indexer_params, hop_size, device=0, d=None, storage=None, ranks=None):
        """ranks: a pfann_amd.dist.Ranks (one process per GPU).  With more than one rank (or PFANN_FORCE_SHARDED=1) the
        database is sharded by whole songs: this process reads and holds only its contiguous song range, every query
        method is then COLLECTIVE (all ranks call it with the same arguments and get the same answers), and the
        batched form hands back this shard's columns of the per-song score matrix (song_range)."""
        self.ranks = ranks if (ranks is not None and ranks.sharded) else None
        if ranks is not None:
            device = ranks.device
        self.dir_for_db = dir_for_db
        self.params = indexer_params
        self.top_k = self.params["top_k"]
        self.frame_shift_mul = self.params.get("frame_shift_mul", 1)
        self.hop_size = hop_size
        self.score_alpha = self.params.get("score_alpha", 0)
        self.timer = None          # a utils.StageTimer: query_batch then reports 'search' and 'rerank' separately

        self.songList = read_file_list(os.path.join(dir_for_db, "songList.txt"))
        key = np.fromfile(os.path.join(dir_for_db, "landmarkKey"), dtype=np.int32)
        assert len(self.songList) == key.shape[0]
        self.song_pos = song_pos_from_key(key)

        n_songs, n_rows = len(self.songList), int(self.song_pos[-1])
        self.song_range = (0, n_songs)                           # songs whose score columns this process holds
        r_lo, r_hi = 0, n_rows
        if self.ranks is not None:
            from .dist import shard_songs
            ranges = shard_songs(self.song_pos, self.ranks.world)
            self.song_range = ranges[self.ranks.rank]
            # (query_launch_chunks must cut alike on every rank -- its launches are collective: the widest shard decides)
            self._widest_shard = max(hi - lo for lo, hi in ranges)
            r_lo, r_hi = int(self.song_pos[self.song_range[0]]), int(self.song_pos[self.song_range[1]])
        emb = None
        lv = os.path.join(dir_for_db, "landmarkValue")
        if os.path.exists(lv):
            try:
                emb, _, n_all = faissio.read_index_flat(lv, rows=(r_lo, r_hi))
                assert n_all == n_rows, "landmarkValue rows != sum(landmarkKey)"
            except (ValueError, struct.error, OSError, IndexError) as x:   # not a flat index / truncated file
                print("landmarkValue unusable (%s): falling back to the raw embeddings file" % x)
                emb = None
        if emb is None:                                         # database.py:96-97 fallback
            if d is None:
                cfg = os.path.join(dir_for_db, "configs.json")
                d = json.load(open(cfg))["model"]["d"]
            path = os.path.join(dir_for_db, "embeddings")
            assert os.path.getsize(path) == n_rows * d * 4, "embeddings rows != sum(landmarkKey)"
            emb = np.fromfile(path, dtype=np.float32, count=(r_hi - r_lo) * d, offset=r_lo * d * 4).reshape(-1, d)
        self.d = emb.shape[1] if emb.ndim == 2 and emb.shape[0] else (d or emb.shape[-1])
        assert emb.shape[0] == r_hi - r_lo, "embeddings rows != sum(landmarkKey)"
        # "use_float16" in the indexer params (or PFANN_DB_STORAGE=f16) selects fp16-only storage: the knob the
        # reference hard-codes as co.useFloat16 = True for its GPU index (database.py:101-104)
        if storage is None:
            storage = os.environ.get("PFANN_DB_STORAGE") or ("f16" if self.params.get("use_float16", False) else "f32")
        self.index = DeviceIndex(self.d, device, storage)
        # ONE statement of which songs this process owns: the cut made above (0-row songs at a shard boundary cannot be
        # told from the row range), handed to the library and read back
        self.index.load(emb, self.song_pos, r_lo, song_range=self.song_range if self.ranks is not None else None)
        if self.ranks is not None:
            assert self.index.owned_songs() == tuple(self.song_range), (self.index.owned_songs(), self.song_range)
        self.sharded = None
        if self.ranks is not None:
            from .dist import ShardedIndex
            self.sharded = ShardedIndex(self.index, self.song_pos, self.top_k, self.frame_shift_mul, self.score_alpha,
                                        group=self.ranks.group, always_exchange=self.ranks.world == 1)
        # per-song score blocks of one launch: at most this many (score, alignment) pairs in HBM (and as many in the
        # pinned landing buffer); the CLIs split a launch group's queries accordingly (query_launch_chunks)
        self.max_score_pairs = int(float(os.environ.get("PFANN_SCORE_BLOCK_MB", "1024")) * (1 << 20)) // 8

    def attach_engine(self, engine):
        """The tools call this with the Engine that embeds their queries: with the exchange stream on, the engine's front
        end then starts behind the shard scan of the exchange in flight (dist.ShardedIndex.hold_front_end)."""
        if self.sharded is not None and self.sharded.xs is not None:
            engine.before_front_end = self.sharded.hold_front_end

    def warmup(self, rows=19 * 64):
        """throw-away queries through search + match: kernel code objects and scratch buffers exist afterwards.  rows: the
        largest number of query rows one launch group will bring (the matcher: PFANN_MAX_BATCH) -- the search workspace is
        sized by it, and growing it later means a hipFree, which waits for everything in flight: the first full group of
        a matcher run used to stall 30 ms behind its own encoder there (profiles/r3/NOTES.md)."""
        if int(self.song_pos[-1]):                   # (the WHOLE database: under ranks every rank must come along)
            q = torch.zeros((19, self.d), device=self.index.device)
            q[:, 0] = 1.0
            self.query_finish(self.query_launch(q, [0], [19], want_song_scores=True))
            nq = max(1, int(rows) // 19)
            big = torch.cat([q] * nq)                # (also loads torch's concatenation kernel, which the CLIs use per group)
            self.query_finish(self.query_launch(big, np.arange(nq) * 19, [19] * nq, want_song_scores=True), reuse_buffers=True)

    # ---- batched form ------------------------------------------------------------------
    def query_launch(self, emb, qstart, qlen, want_song_scores=False, mode=None):
        """First half of query_batch: search + sequence match launched asynchronously, nothing read back.  The CLIs launch
        group g+1 before they finish group g, so the GPU never idles while the host formats and writes results."""
        dev = self.index.device
        if mode is None:
            mode = 1 if cpp_accelerate else 0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] + [torch.cuda.Event()]
        # song-sharded, PFANN_EXCHANGE_STREAM=1: search, collectives, matcher and winner pick run on the exchange stream
        # (dist.ShardedIndex.exchange), so the next group's encoder does not queue behind the collectives
        with (self.sharded.exchange(emb) if self.sharded is not None else contextlib.nullcontext()):
            ev[0].record()
            if self.sharded is not None:
                # two-phase shard search + all-to-all merge (`search`), owner-side match + 128-bit key all-gather +
                # device pick (`rerank`); `res` are the winners over ALL shards, `ss` this shard's columns
                D, I = self.sharded.search_global(emb)
                ev[1].record()
                res, ss = self.sharded.match_global(emb, I, qstart, qlen, want_song_scores, mode)
            else:
                D, I = self.index.search(emb, self.top_k)
                ev[1].record()
                res, ss = self.index.match(emb, I, qstart, qlen, self.frame_shift_mul, self.score_alpha, mode,
                                           False, want_song_scores, to_host=False)
            ev[2].record()
            # frames -> seconds where the block lives (database.py:148,193 do it on the host): (t - shift/fsm) * hop_size
            # with fine = t*fsm - shift, in double like the reference's Python floats, stored as float32
            self.index.song_scores_to_seconds(ss, self.frame_shift_mul, self.hop_size, native_path=mode == 1)
            ev[3].record()              # (the block is complete HERE, not at ev[2]: query_finish copies it after this one)
        return {"res": res, "ss": ss, "ev": ev, "nq": len(qlen), "keep": (emb, I), "dev": dev, "mode": mode}

    def query_launch_chunks(self, emb, qstart, qlen, want_song_scores=False, mode=None):
        """query_launch over as many sub-launches as the score-block budget asks for (PFANN_SCORE_BLOCK_MB, default
        1024: one-segment queries against a 100 k-song database would otherwise want 7.8 GB of HBM and as much pinned
        host memory per launch group).  -> an iterable of (first query, one past the last, launch), LAZY beyond its first
        element: sub-launch 0 is in flight when this returns (the CLIs launch group g+1 before they read group g back),
        sub-launch i+1 is launched when the consumer asks for sub-launch i, i.e. just before it reads i back.  At most
        three score blocks of max_score_pairs pairs therefore exist at any time -- the one being read back, its
        successor, and the first one of the next launch group -- whatever the number of sub-launches (all of them at
        once until round 5: the budget then bounded only the pinned landing buffer)."""
        nq = len(qlen)
        width = max(getattr(self, "_widest_shard", self.song_range[1] - self.song_range[0]), 1)
        step = max(1, nq) if not want_song_scores else max(1, min(nq, self.max_score_pairs // width))
        qstart = np.asarray(qstart, dtype=np.int64)

        def launch(j0, j1):
            r0 = int(qstart[j0])
            r1 = int(qstart[j1 - 1]) + int(qlen[j1 - 1])
            sub = emb if (j0 == 0 and j1 == nq) else emb[r0:r1]
            return self.query_launch(sub, qstart[j0:j1] - r0, qlen[j0:j1], want_song_scores, mode)
        return LazyLaunches([(j0, min(j0 + step, nq)) for j0 in range(0, nq, step)], launch)

    def _pinned(self, shape, dtype):
        """one reusable pinned landing buffer per result kind (a pinned allocation costs milliseconds)"""
        n = int(np.prod(shape))
        key = str(dtype)
        buf = self._pin.get(key)
        if buf is None or buf.numel() < n:
            buf = torch.empty(max(n, 1), dtype=dtype, pin_memory=True)
            self._pin[key] = buf
        return buf[:n].view(shape)

    def query_finish(self, p, reuse_buffers=False):
        """Second half: wait for that group only (a side stream copies its results; later groups keep running) and
        return the list of (score, (song, time), song_score|None).  reuse_buffers: the song_score blocks are views of a
        pinned buffer that the NEXT query_finish overwrites (the CLIs write them out at once)."""
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(p["dev"])
            self._pin = {}
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(p["ev"][-1])
            ss_np = None
            if p["ss"] is not None:
                land = self._pinned(p["ss"].shape, torch.float32)
                land.copy_(p["ss"], non_blocking=True)
            res = self.index.results_to_host(p["res"])    # synchronises the copy stream: `land` is complete too
            if p["ss"] is not None:
                self._copy_stream.synchronize()
                ss_np = land.numpy() if reuse_buffers else land.numpy().copy()
        if self.timer is not None:                        # stage split as database.py:165 logs it, from events
            self.timer.mark_gpu("search", p["ev"][0], p["ev"][1])
            self.timer.mark_gpu("rerank", p["ev"][1], p["ev"][2])
            self.timer.resolve()
        out = []
        fsm = self.frame_shift_mul
        for j in range(p["nq"]):
            r = res[j]
            song_score = ss_np[j] if ss_np is not None else None
            if p.get("mode", 0) == 1:
                # query_embeddings_cpp (database.py:166-195) reads score and time back from the per-song block, which only
                # ever records scores > 0 (no candidate: song_score[-1], the last song's untouched row)
                sc32 = float(np.float32(r["score"])) if r["song"] >= 0 else 0.0
                if r["song"] < 0 or not sc32 > 0.0:
                    out.append((0.0, (int(r["song"]) if r["song"] >= 0 else -1, 0.0), song_score))
                else:
                    fine = float(np.float32(int(r["offset"]) * fsm - int(r["shift"])))
                    out.append((sc32, (int(r["song"]), fine * self.hop_size / fsm), song_score))
                continue
            if self.index.ntotal == 0 or r["song"] < 0:
                out.append((-1e999, (-1, 0), song_score))
                continue
            real_time = (int(r["offset"]) - int(r["shift"]) / fsm) * self.hop_size
            out.append((float(r["score"]), (int(r["song"]), real_time), song_score))
        return out

    def query_batch(self, emb, qstart, qlen, want_song_scores=False, mode=None):
        """emb: torch cuda [sum(qlen), d] unit-norm rows; -> list of (score, (song, time), song_score|None)."""
        return self.query_finish(self.query_launch(emb, qstart, qlen, want_song_scores, mode))

    # ---- the reference's per-query contract ---------------------------------------------
    def query_embeddings(self, query):
        q = torch.as_tensor(np.ascontiguousarray(query, dtype=np.float32)) if not isinstance(query, torch.Tensor) else query
        q = q.to(self.index.device)
        if self.sharded is not None:
            # the reference's tuple on every rank: the shards' score columns all-gathered into the [n_songs, 2] block
            from .dist import all_gather_ragged, shard_songs
            p = self.query_launch(q, [0], [q.shape[0]], want_song_scores=True)
            counts = [hi - lo for lo, hi in shard_songs(self.song_pos, self.ranks.world)]
            with self.sharded.on_exchange_stream():         # (the block was produced there when the exchange stream is on)
                full = all_gather_ragged(p["ss"][0], counts, self.ranks.group).cpu().numpy()
            p["ss"] = None
            (score, best_song_t, _), = self.query_finish(p)
            return score, best_song_t, full
        (score, best_song_t, song_score), = self.query_batch(q, [0], [q.shape[0]], want_song_scores=True)
        return score, best_song_t, song_score

    query_embeddings_base = query_embeddings
