"""Song-sharded retrieval across the GPUs of one node (new capability, the
reference is single-device).  One process per GPU, `torch.distributed` backend "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

Per query batch there is exactly one exchange step, small collectives only:
  1. every rank scans ITS shard for all query rows -> all-to-all of the per-shard top-k
     (score f32, global label i64)[Q,k] by query slice -> rank r merges slice r to the GLOBAL top-k
     -> all_gather of the merged slices (needed for parity: the reference's candidates come from
     the global list);
  2. every rank sequence-scores the candidates whose song it owns -> all_gather of the
     per-rank best (score, song, offset, shift) -> lexicographic argmax in the reference's
     candidate order (ties -> smallest (shift, song, offset)).
Shards are contiguous song ranges balanced by row count and never split a song, because a
sequence score only touches consecutive rows of one song.

The compute backend is any object with the DeviceIndex interface
(search / merge_topk / match); the product uses pfann_amd.database.DeviceIndex (HIP).
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist


class Ranks:
    """The process group a product entry point (builder.py / matcher.py / extractemb.py / matchemb.py, `Database`) runs
    in: one process per GPU.  `group` carries the data collectives (RCCL; gloo when PFANN_DIST_BACKEND=gloo), `meta` is
    a gloo group for small host-side metadata (segment counts of a launch group, barriers around file creation) that
    must not queue behind GPU work."""

    def __init__(self, rank, world, device, backend, meta=None, force_sharded=False):
        self.rank, self.world, self.device, self.backend = rank, world, device, backend
        self.group = None                      # the default group
        self.meta = meta
        self.sharded = world > 1 or force_sharded

    def barrier(self):
        dist.barrier(group=self.meta)

    def sum_host(self, arr):
        """element-wise SUM of a small int64 host array over the ranks (gloo, never touches a HIP stream)"""
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64).copy())
        if self.world > 1:
            dist.all_reduce(t, group=self.meta)
        return t.numpy()

    def max_host(self, value):
        t = torch.tensor([float(value)], dtype=torch.float64)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.meta)
        return float(t.item())


class quiet_stdout:
    """File descriptor 1 points at stderr inside the block: gloo announces every connection on stdout ("[Gloo] Rank 0 is
    connected to 7 peer ranks ..."), and the tools' / bench.py's stdout is a contract (stage lines, ONE JSON line)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def self_launch_if_asked(argv):
    """See pfann_amd/launch.py (the torch-free launcher the root shims call before they import anything heavy)."""
    from .launch import self_launch_if_asked as _go
    return _go(argv)


def init_ranks():
    """-> Ranks when this process was started as one rank of a job (WORLD_SIZE in the environment: torch.distributed.run,
    or self_launch_if_asked), else None.  One rank per GPU (LOCAL_RANK; PFANN_FORCE_DEVICE pins every rank to one device:
    a debugging aid for one-GPU boxes together with PFANN_DIST_BACKEND=gloo).  PFANN_FORCE_SHARDED=1 runs the sharded
    protocol even at world 1, so every collective really goes through the backend."""
    if "WORLD_SIZE" not in os.environ:
        return None
    backend = os.environ.get("PFANN_DIST_BACKEND", "nccl")
    local = int(os.environ.get("PFANN_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if torch.cuda.is_available():
        if local >= torch.cuda.device_count():
            raise RuntimeError("rank %s wants HIP device %d but only %d are visible" %
                               (os.environ.get("RANK"), local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    with quiet_stdout():
        if not dist.is_initialized():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
        meta = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else None
        dist.barrier(group=meta)                   # (gloo connects lazily: make it talk now)
    r = Ranks(dist.get_rank(), dist.get_world_size(), local, dist.get_backend(), meta,
              force_sharded=os.environ.get("PFANN_FORCE_SHARDED", "0") != "0")
    if backend == "nccl" and r.world > 1:
        seen = [None] * r.world
        dist.all_gather_object(seen, local, group=meta)
        if len(set(seen)) != r.world:
            raise RuntimeError("%d ranks share HIP devices %r: RCCL needs one device per rank" % (r.world, seen))
    return r


def finish_ranks(ranks):
    if ranks is not None and dist.is_initialized():
        dist.barrier(group=ranks.meta)
        dist.destroy_process_group()


def shard_songs(song_pos, world):
    """-> list of (song_lo, song_hi) per rank: contiguous, balanced by rows, whole songs."""
    song_pos = np.asarray(song_pos, dtype=np.int64)
    n_songs = song_pos.shape[0] - 1
    total = int(song_pos[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        s = int(np.searchsorted(song_pos, target, side="left"))
        s = min(max(s, cuts[-1]), n_songs)
        cuts.append(s)
    cuts.append(n_songs)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def split_even(n, world):
    """-> list of (lo, hi) splitting range(n) into `world` near-equal contiguous parts."""
    base, rem = divmod(n, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def all_gather_rows(x, group=None):
    """all_gather of equally shaped tensors -> [world, *x.shape]."""
    world = dist.get_world_size(group)
    if x.is_cuda and dist.get_backend(group) == "gloo":       # debugging on one GPU: stage through host
        return all_gather_rows(x.cpu(), group).to(x.device)
    out = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out.view(-1), x.contiguous().view(-1), group=group)
    return out


def all_to_all_rows(x, group=None):
    """x [world, ...]: slice r goes to rank r -> out [world, ...] with out[r] = what rank r sent here."""
    if x.is_cuda and dist.get_backend(group) == "gloo":       # debugging on one GPU: stage through host
        return all_to_all_rows(x.cpu(), group).to(x.device)
    x = x.contiguous()
    out = torch.empty_like(x)
    dist.all_to_all_single(out.view(-1), x.view(-1), group=group)
    return out


def all_gather_ragged(x, counts, group=None):
    """Rows split unevenly over ranks (counts[r] rows on rank r) -> concatenated [sum, ...]."""
    world = dist.get_world_size(group)
    mx = max(counts) if counts else 0
    pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    g = all_gather_rows(pad, group)
    return torch.cat([g[r, : counts[r]] for r in range(world)], dim=0)


class _ExchangeScope:
    def __init__(self, stream, tensors, join):
        self.stream, self.tensors, self.join, self.ctx = stream, tensors, join, None

    def __enter__(self):
        if self.stream is None:
            return self
        if self.join:
            ev = torch.cuda.Event()
            ev.record()
            self.stream.wait_event(ev)
            for t in self.tensors:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(self.stream)
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)
        return False


class ShardedIndex:
    def __init__(self, backend, song_pos, top_k, frame_shift_mul=1, score_alpha=0.0, group=None, always_exchange=False):
        """always_exchange: run the whole exchange protocol (bound all-gather, all-to-all, merge, all-gathers) even when
        the group has one rank -- a one-GPU box then exercises every collective of the N-rank path on its real backend."""
        self.always_exchange = always_exchange
        self.b = backend
        self.song_pos = np.asarray(song_pos, dtype=np.int64)
        self.k = top_k
        self.fsm = frame_shift_mul
        self.alpha = score_alpha
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.timing = None          # a dict: every collective of the protocol is then bracketed by a pair of timing events
        # PFANN_EXCHANGE_STREAM=1 (round 5; default off until it has been measured on xGMI): the exchange of batch i --
        # shard search, collectives, merge, owner-side matcher, winner pick -- runs on a stream of its own, so that the
        # encoder of batch i+1 (issued on the caller's stream) does not queue behind the collectives.  Callers bracket the
        # exchange with `with sharded.exchange(emb):` and read its results under `with sharded.on_exchange_stream():`
        # (or through events recorded inside the block).  Same kernels, same order within the exchange: same bytes out.
        self.xs = None
        self.scan_done = None       # (exchange stream on) event: this rank's shard scan of the exchange in flight has run
        if os.environ.get("PFANN_EXCHANGE_STREAM", "0") not in ("0", "") and torch.cuda.is_available() and hasattr(backend, "device"):
            self.xs = torch.cuda.Stream(device=backend.device)

    def exchange(self, *tensors):
        """Context: what is launched inside belongs to the exchange of one batch.  With the exchange stream on, that stream
        first waits for everything the current stream has queued so far (the fingerprints `tensors` were produced there:
        they are marked as in use on the exchange stream), then becomes the current stream.  Nothing joins back at the
        end: consumers wait on events recorded inside, or read under on_exchange_stream()."""
        return _ExchangeScope(self.xs, tensors, join=True)

    def hold_front_end(self):
        """Exchange stream on: the CALLER's stream waits until the shard scan of the exchange in flight has run; call it
        before launching the next batch's log-mel front end (the tools hang it on Engine.before_front_end).

        Why (round 5, profiles/r5/NOTES.md, tools/ubench/xs_race_probe3.py / 5 / 6): the batched fp16 scan kernel
        (scan_f16_qres_kernel) running on one stream PERTURBS an FFT kernel that stages through LDS running beside it on
        another -- this library's melspec_kernel (a handful of windows per launch get one frame's FFT bins wrong: max
        fingerprint error 3e-2) and rocFFT under torch.fft alike; the encoder's GEMMs, the matcher, sorts, softmax,
        rocBLAS are not affected, the fp32 scan does not do it, and neither kernel touches memory it does not own (direct-
        to-LDS loads, LDS ranges, scratch, descriptors and hazards were each ruled out; the trigger depends on the scan
        kernel's code generation).  Unexplained, so it is AVOIDED: on one stream the two never overlap (the default), and
        with the exchange stream on the next batch's front end waits the ~3 ms of the scan while the encoder's GEMMs -- 97 %
        of its time -- still run beside the collectives, which is what the flag is for."""
        if self.xs is not None and self.scan_done is not None:
            torch.cuda.current_stream().wait_event(self.scan_done)

    def _mark_scan_done(self):
        """(exchange stream on) records the event hold_front_end waits for: every scan-class kernel of the exchange in
        flight -- sampled pass, bound pass, full pass, selects -- has been queued in front of it, on every path of
        search_global (the one-shard shortcut included)"""
        if self.xs is not None:
            self.scan_done = torch.cuda.Event()
            self.scan_done.record()             # (hold_front_end: the next batch's log-mel kernel starts behind this)

    def on_exchange_stream(self):
        """Context: the exchange stream is current (no wait on the caller's stream) -- for reading an exchange's results
        back without queueing the copy behind the NEXT batch's encoder."""
        return _ExchangeScope(self.xs, (), join=False)

    def _timed(self, name, fn, *a):
        """runs one collective; with self.timing set, between two events on the current stream (the collective's own
        stream is joined into the current one by the process group, so the pair spans it)"""
        if self.timing is None or not torch.cuda.is_available():
            return fn(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a)
        e1.record()
        self.timing.setdefault(name, []).append((e0, e1))
        return out

    def timing_ms(self):
        """-> {collective: total milliseconds} over the event pairs collected so far (waits for them), and forgets them"""
        out = {}
        for name, pairs in (self.timing or {}).items():
            tot = 0.0
            for e0, e1 in pairs:
                e1.synchronize()
                tot += e0.elapsed_time(e1)
            out[name] = tot
        if self.timing is not None:
            self.timing = {}
        return out

    @staticmethod
    def my_rows(song_pos, rank, world):
        lo, hi = shard_songs(song_pos, world)[rank]
        return int(song_pos[lo]), int(song_pos[hi])

    def search_global(self, q):
        """q [Q, d] (identical on all ranks) -> global (D, I) [Q, k], identical on all ranks.
        The merge is itself distributed: rank r merges the per-shard lists of query slice r (all-to-all),
        then the merged slices are all-gathered -- every rank receives 2*Q*k entries instead of G*Q*k
        (an all-gather of everything was 93 MB per rank and step at 8 GPUs) and merges 1/G of the rows."""
        G, k, Q = self.world, self.k, q.shape[0]
        if G == 1 and not self.always_exchange:
            out = self.b.search(q, k)
            self._mark_scan_done()
            return out
        # Two-phase shard search: every rank extracts from its sampled pass the m best scores of each row (m different
        # real rows of its shard, lowered to bounds of their exact scores), the values are all-gathered (4 m bytes per row
        # and rank) and the k-th largest of a row's union bounds its k-th best over ALL shards from below; the full pass
        # then emits only rows that can be in the GLOBAL top-k.  Without it every shard digs out its own complete top-k:
        # the survivor density per db row -- what the scan's epilogue and the select pay for -- grows with the number of
        # shards (8 shards: almost every 32x32 block of the scan holds a survivor).  m = 2k/G + 8 covers a row whose
        # best matches are spread evenly over the shards twice over; a lopsided row just gets a looser (still valid) bound.
        q = q.contiguous()
        m = min(k, 2 * k // G + 8)
        Dl, Il = [], []
        for c0 in range(0, Q, self.b.BOUND_CHUNK):
            qc = q[c0:c0 + self.b.BOUND_CHUNK]
            lb = self.b.reduce_bound(self._timed("bound_allgather", all_gather_rows, self.b.search_bound(qc, k, m), self.group), k)
            Dc, Ic = self.b.search_bounded(qc, k, lb)
            Dl.append(Dc)
            Il.append(Ic)
        D, I = torch.cat(Dl), torch.cat(Il)
        self._mark_scan_done()
        Qs = (Q + G - 1) // G
        pad = Qs * G - Q
        if pad:
            D = torch.cat([D, torch.full((pad, k), -3.4028234663852886e38, dtype=D.dtype, device=D.device)])
            I = torch.cat([I, torch.full((pad, k), -1, dtype=I.dtype, device=I.device)])
        rD = self._timed("all_to_all", all_to_all_rows, D.view(G, Qs, k), self.group)        # [source shard, my Qs queries, k]
        rI = self._timed("all_to_all", all_to_all_rows, I.view(G, Qs, k), self.group)
        if hasattr(self.b, "merge_lists"):                        # the lists as delivered, one wavefront per row
            Dm, Im = self._timed("merge", self.b.merge_lists, rD, rI, k)
        else:
            S = rD.permute(1, 0, 2).reshape(Qs, G * k)            # shard-major: ascending labels on ties
            L = rI.permute(1, 0, 2).reshape(Qs, G * k)
            Dm, Im = self._timed("merge", self.b.merge_topk, S.contiguous(), L.contiguous(), k)
        gD = self._timed("slice_allgather", all_gather_rows, Dm, self.group).reshape(G * Qs, k)[:Q]
        gI = self._timed("slice_allgather", all_gather_rows, Im, self.group).reshape(G * Qs, k)[:Q]
        return gD.contiguous(), gI.contiguous()

    def match_global(self, q, I, qstart, qlen, want_song_scores=False, mode=0):
        """Second half of a sharded query, given the GLOBAL labels of search_global: -> (winners over all shards: device
        tensor of results, identical on all ranks; this shard's song_scores block [nQ, owned songs, 2] with alignments in
        fine frames, or None).  Nothing is read back."""
        res, ss = self.b.match(q, I, qstart, qlen, self.fsm, self.alpha, mode, True, want_song_scores, to_host=False,
                               owned_block=True)
        keys = self.b.pack_winner_keys(res)
        allk = self._timed("key_allgather", all_gather_rows, keys, self.group)
        return self.b.pick_winner(allk, to_host=False), ss

    def query_batch(self, q, qstart, qlen, to_host=True):
        """-> structured array (song, offset, shift, score) per query, identical on all ranks.
        Everything between the search and the final result stays on the device: the owner-side results are packed
        into one 128-bit orderable key per query (pfann_match_pack), all-gathered (16 bytes per query and rank) and
        reduced by pfann_match_pick; the winners come back in the step's single device-to-host copy."""
        D, I = self.search_global(q)
        res, _ = self.b.match(q, I, qstart, qlen, self.fsm, self.alpha, 0, True, False, to_host=False)
        keys = self.b.pack_winner_keys(res)                          # int64 [nQ, 2]
        allk = self._timed("key_allgather", all_gather_rows, keys, self.group)       # [G, nQ, 2]
        return self.b.pick_winner(allk, to_host=to_host)      # to_host=False: the device tensor (results_to_host later)
