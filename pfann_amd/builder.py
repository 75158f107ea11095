"""Database builder CLI, drop-in for the reference's builder.py:
    python builder.py <music list file> <db location> [config.json | model dir]

Same argv, same db directory layout (`embeddings`, `landmarkValue`, `landmarkKey`,
`songList.txt`, `configs.json`, `model.pt`; reference builder.py:30-44,71,136-148), but
the per-song loop (builder.py:75-103) runs on the MI355X: PCM is uploaded as int16,
converted/segmented/mel-ed/encoded by the HIP kernels, several songs per launch, and the
fingerprints only come back to the host once, to be written to disk.
"""
import collections
import ctypes
import os
import queue
import shutil
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import faissio
from . import lib as _l
from .engine import Engine
from .musicdata import MusicDataset
from .utils import StageTimer, init_logger, read_config


class _PinnedPool:
    """Pinned int16 staging buffers in power-of-two sizes, reused for the whole run (a pinned allocation costs about as
    much as decoding a song).  Decode workers take buffers concurrently; the main thread gives them back once the
    stream has passed the upload that read them."""

    def __init__(self):
        self.free = {}
        self.lock = threading.Lock()
        # slabs the tools' start-up thread allocated while torch was importing (prewarm.py): a pinned allocation in the
        # middle of a run stalls every launch of the process for its 10-40 ms
        from . import prewarm
        for addr, n in prewarm.take_pinned():
            t = torch.frombuffer((ctypes.c_int16 * n).from_address(addr), dtype=torch.int16)
            self.free.setdefault(n, []).append(t)

    def get(self, n):
        # NOTE: the slab handed out holds AT LEAST n samples and may be larger than its own size class (a free larger slab
        # is reused for a smaller group): callers go by the n they asked for, never by slab.numel()
        # per-file buffers: powers of two; a launch group's slab (tens of MB): the next multiple of 8 M samples -- a
        # 41 M-sample group took a 67 M-sample slab, and a slab's price is its size
        cap = 1 << max(12, int(n - 1).bit_length()) if n <= (1 << 22) else -(-n // (1 << 23)) * (1 << 23)
        with self.lock:
            lst = self.free.get(cap)
            if lst:
                return lst.pop()
            if n > (1 << 20):                   # a smaller group (the ramp, the tail) takes a free larger slab rather than a new one
                for c in sorted(self.free):
                    if c >= cap and self.free[c]:
                        return self.free[c].pop()
        # allocated pinned (torch.empty(...).pin_memory() allocates pageable memory, pins a second block and copies the
        # garbage over: 60 ms per 134 MB slab on the loader thread -- with a fresh slab per group until the first one comes
        # back, a 2000-query matcher run was fed one group per 83 ms for a GPU that needs 41)
        # ... and allocated WITHOUT the GIL (lib.pinned_int16): the loader thread takes its first slabs while the main thread
        # is launching the first groups
        return _l.pinned_int16(cap)

    def put(self, bufs):
        with self.lock:
            for b in bufs:
                self.free.setdefault(b.numel(), []).append(b)


def _decode(dataset, i, pool, native_rate):
    """Runs on a decode worker: file -> int16 PCM in a pinned buffer.  -> (pcm [n, ch] pinned tensor view, rate, buffer)."""
    taken = []

    def alloc(n):                           # MusicDataset reads the samples straight into pinned memory
        buf = pool.get(max(n, 1))
        taken.append(buf)
        return buf.numpy()[:n]
    if hasattr(dataset, "load_pcm_sr"):
        try:
            pcm, sr = dataset.load_pcm_sr(i, alloc)
        except TypeError:                   # a dataset whose load_pcm_sr takes no allocator
            pcm, sr = dataset.load_pcm_sr(i)
    else:                                   # a list of PCM the caller vouches is at the model's rate (bench.py)
        pcm, sr = dataset.load_pcm(i), native_rate
    if sr is None:
        raise ValueError("sample rate of %s unknown" % dataset.files[i])
    if isinstance(pcm, torch.Tensor):
        return pcm, sr, None
    pcm = np.asarray(pcm)
    if pcm.dtype != np.int16:
        raise ValueError("16-bit PCM expected, got %s" % pcm.dtype)
    if pcm.size == 0 and int(sr) != int(native_rate):
        raise ValueError("empty file at a foreign rate")      # the reference's resampler raises there: 0-segment song
    if taken:
        buf = taken[0]
        return buf[:pcm.size].view(pcm.shape), sr, buf
    buf = pool.get(max(pcm.size, 1))
    view = buf[:pcm.size].view(pcm.shape)
    np.copyto(view.numpy(), pcm)
    return view, sr, buf


_WAV_INFO = np.dtype([("n_frames", "<i8"), ("data_pos", "<i8"), ("n_ch", "<i4"), ("sample_rate", "<i4"), ("status", "<i4"),
                      ("reserved", "<i4")])
_WAV_ERR = {-1: "cannot open the file", -2: "not a RIFF/WAVE file, or fmt/data chunk missing", -3: "not PCM",
            -4: "wave stream currently only supports 16bit wav", -5: "read error"}


def _threaded_groups(engine, dataset, hop, batch_windows, pool, workers, ahead):
    """Generic source: any dataset with load_pcm_sr / load_pcm, decoded by a pool of Python threads.  Yields
    (items, slab, release, t_load) per launch group; items = [(index, n_seg, item)], item = ("host", pinned int16 1-D view, n)
    | ("dev", mono float wav on the device) | None (load error)."""
    seg = engine.seg_len
    native = int(engine.params["sample_rate"])
    n = len(dataset)
    ex = ThreadPoolExecutor(max_workers=max(workers, 1)) if workers > 0 else None
    futs = collections.deque()
    nxt = 0
    pending, held, n_win, t_load = [], [], 0, 0.0
    try:
        for i in range(n):
            while ex is not None and nxt < n and nxt - i < ahead:
                futs.append(ex.submit(_decode, dataset, nxt, pool, native))
                nxt += 1
            buf = None
            try:
                t0 = time.perf_counter()                # "load": what the GPU-feeding thread waited for the decoders
                pcm, sr, buf = futs.popleft().result() if ex is not None else _decode(dataset, i, pool, native)
                t_load += time.perf_counter() - t0
                n_in = pcm.shape[0]
                pinned_host = buf is not None or (isinstance(pcm, torch.Tensor) and pcm.device.type == "cpu" and
                                                  pcm.dtype == torch.int16 and pcm.is_contiguous() and pcm.is_pinned())
                if pinned_host and int(sr) == native and (pcm.dim() == 1 or pcm.shape[1] == 1):
                    item = ("host", pcm.reshape(-1), n_in)                    # the common case: uploaded with its group
                    n_out = n_in
                else:                                                         # stereo / another rate / a caller's tensor
                    wav = engine.pcm16_to_mono(pcm, sample_rate=sr)
                    item = ("dev", wav)
                    n_out = wav.shape[0]
                n_seg = (max(n_out, seg) - seg) // hop + 1
            except Exception as x:                                            # musicdata.py:95-101
                print("load %s error! (%s)" % (dataset.files[i], x))
                item, n_seg = None, 0            # (a buffer taken before the failure still goes back to the pool)
            if pending and n_win + n_seg > batch_windows:     # a group never exceeds the encoder's chunk (no small tail pass)
                yield pending, None, held, t_load
                pending, held, n_win, t_load = [], [], 0, 0.0
            # file i's pinned buffer belongs to the group file i is uploaded with -- the one it is appended to HERE, after
            # the overflow cut (it used to join `held` before the cut and was handed back to the pool with the previous
            # group, while its own upload was still queued: a decode worker could overwrite it)
            if buf is not None:
                held.append(buf)
            pending.append((i, n_seg, item))
            n_win += n_seg
            if n_win >= batch_windows:
                yield pending, None, held, t_load
                pending, held, n_win, t_load = [], [], 0, 0.0
        if pending:
            yield pending, None, held, t_load
    finally:
        if ex is not None:
            ex.shutdown(wait=True, cancel_futures=True)


def _native_groups(engine, dataset, hop, batch_windows, pool, workers, rank=0, world=1):
    """WAV files through the library's native reader (csrc/wavio.hip): a producer thread probes the headers, cuts the
    list into launch groups, and has `workers` native threads read each group's samples into ONE pinned slab, two
    groups ahead of the GPU.  Yields like _threaded_groups; slab = (pinned int16 tensor, samples used) when the group
    is uniform (every readable file mono, at the model's rate, at least one segment long: the slab IS the group's
    concatenated PCM and goes up in one copy).

    world > 1 (one process per GPU): the list is cut into ROUNDS of `world` consecutive launch groups, every rank
    probes all headers (so all ranks cut alike) but reads only the group with its own number; a fifth element is then
    yielded, the round's layout [(first file, [predicted n_seg per file])] * world, and ranks without a group in the
    last round yield empty items."""
    from . import resample
    lib = engine.lib
    seg = engine.seg_len
    native = int(engine.params["sample_rate"])
    files = dataset.files
    n = len(files)
    paths = [os.fsencode(f) for f in files]
    out_q = queue.Queue(maxsize=2)
    stop = threading.Event()
    PROBE = 1024

    def produce():
        try:
            pos = 0
            n_rounds = 0
            ramp = os.environ.get("PFANN_GROUP_RAMP", "1") != "0"
            info = np.zeros(0, _WAV_INFO)
            nseg = np.zeros(0, np.int64)
            base = 0                        # file index of info[0]
            while (pos < n or info.shape[0]) and not stop.is_set():
                if info.shape[0] == 0 or (pos < n and nseg.sum() < batch_windows * world):
                    m = min(PROBE, n - pos)
                    if m > 0:
                        more = np.zeros(m, _WAV_INFO)
                        arr = (ctypes.c_char_p * m)(*paths[pos:pos + m])
                        lib.pfann_wav_probe(arr, m, workers, more.ctypes.data_as(ctypes.POINTER(_l.WavInfo)))
                        n_out = more["n_frames"].copy()
                        for j in np.nonzero((more["status"] == 0) & (more["sample_rate"] != native))[0]:
                            if more["n_frames"][j] == 0 or more["sample_rate"][j] <= 0:
                                more["status"][j] = -2            # the reference's resampler raises on an empty signal
                            else:
                                n_out[j] = resample.piece_plan(int(more["n_frames"][j]), int(more["sample_rate"][j]), native)[1]
                        more_seg = np.where(more["status"] == 0, (np.maximum(n_out, seg) - seg) // hop + 1, 0)
                        info = np.concatenate([info, more])
                        nseg = np.concatenate([nseg, more_seg])
                        pos += m
                        continue
                # cut one ROUND of `world` groups off the front, each as many files as fit the group's window budget (at
                # least one).  The run starts with a quarter and a half group and (one rank) ends with a quarter group:
                # the GPU starts after a quarter group's decode instead of a whole one's, and what is left to do when it
                # stops (the last group's result copy and file writes) is a quarter group's (PFANN_GROUP_RAMP=0: whole
                # groups throughout).  Several ranks: the last round is cut into equal shares.
                left = int(nseg.sum())
                limit, tail = batch_windows, False
                if ramp and n_rounds < 2:
                    limit = max(1, batch_windows >> (2 - n_rounds))
                elif ramp and world == 1 and pos >= n and batch_windows // 2 < left <= batch_windows:
                    limit = left - batch_windows // 4
                if world > 1 and pos >= n and left <= world * limit:
                    limit, tail = max(1, -(-left // world)), True
                n_rounds += 1
                layout, mine = [], None
                for r in range(world):
                    if info.shape[0] == 0:
                        layout.append((base, np.zeros(0, np.int64)))
                        continue
                    c = np.cumsum(nseg)
                    if tail:        # the file that crosses the share still belongs to it (unless the encoder chunk overflows)
                        take = int(np.searchsorted(c, limit, side="left")) + 1
                        while take > 1 and c[min(take, c.shape[0]) - 1] > batch_windows:
                            take -= 1
                        take = min(take, c.shape[0])                    # (a remainder, if any, makes one more round)
                    else:
                        take = max(1, int(np.searchsorted(c, limit, side="right")))       # files whose windows still fit
                    g_info, g_seg = info[:take].copy(), nseg[:take].copy()
                    info, nseg = info[take:], nseg[take:]
                    layout.append((base, g_seg.copy()))
                    if r == rank:
                        ok = g_info["status"] == 0
                        sizes = np.where(ok, g_info["n_frames"] * g_info["n_ch"], 0)
                        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
                        tot = int(sizes.sum())
                        slab = pool.get(max(tot, 1))
                        arr = (ctypes.c_char_p * take)(*paths[base:base + take])
                        rc = lib.pfann_wav_read(arr, take, workers, g_info.ctypes.data_as(ctypes.POINTER(_l.WavInfo)),
                                                offs.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), slab.data_ptr(), slab.numel())
                        if rc != 0:
                            raise _l.PfannError("pfann_wav_read failed (%d)" % rc)
                        g_seg = np.where(g_info["status"] == 0, g_seg, 0)          # a read error after a good probe
                        mine = (base, g_info, g_seg, offs, slab, tot)
                    base += take
                out_q.put((mine, layout))
            out_q.put(None)
        except BaseException as x:          # hand the failure to the consumer instead of dying silently
            out_q.put(x)

    th = threading.Thread(target=produce, name="pfann-wav-loader", daemon=True)
    th.start()
    if _TIMELINE:
        print("timeline loader thread started at %.1f ms" % (1e3 * (time.perf_counter() - _T0[0])), file=sys.stderr)
    try:
        while True:
            t0 = time.perf_counter()
            got = out_q.get()
            t_load = time.perf_counter() - t0
            if got is None:
                break
            if isinstance(got, BaseException):
                raise got
            mine, layout = got
            if mine is None:                       # fewer groups than ranks in the last round: nothing of ours
                yield [], None, [], t_load, layout
                continue
            base, g_info, g_seg, offs, slab, tot = mine
            if _TIMELINE:
                print("timeline group received at %.1f ms (waited %.1f)" % (1e3 * (time.perf_counter() - _T0[0]), 1e3 * t_load), file=sys.stderr)
            ok = g_info["status"] == 0
            for j in np.nonzero(~ok)[0]:
                print("load %s error! (%s)" % (files[base + j], _WAV_ERR.get(int(g_info["status"][j]), "error")))
            uniform = bool(ok.any()) and bool(np.all(~ok | ((g_info["n_ch"] == 1) & (g_info["sample_rate"] == native) &
                                                            (g_info["n_frames"] >= seg))))
            items = []
            for j in range(g_info.shape[0]):
                if not ok[j]:
                    items.append((base + j, 0, None))
                    continue
                nf, ch, o = int(g_info["n_frames"][j]), int(g_info["n_ch"][j]), int(offs[j])
                if uniform:
                    items.append((base + j, int(g_seg[j]), ("slab", o, nf)))
                elif ch == 1 and int(g_info["sample_rate"][j]) == native:
                    items.append((base + j, int(g_seg[j]), ("host", slab[o:o + nf], nf)))
                else:
                    try:
                        wav = engine.pcm16_to_mono(slab[o:o + nf * ch].view(nf, ch), sample_rate=int(g_info["sample_rate"][j]))
                        assert (max(wav.shape[0], seg) - seg) // hop + 1 == int(g_seg[j])
                        items.append((base + j, int(g_seg[j]), ("dev", wav)))
                    except Exception as x:
                        print("load %s error! (%s)" % (files[base + j], x))
                        items.append((base + j, 0, None))
            if world > 1:
                yield items, ((slab, tot) if uniform else None), [slab], t_load, layout
            else:
                yield items, ((slab, tot) if uniform else None), [slab], t_load
    finally:
        stop.set()
        while th.is_alive():                # unblock a producer waiting on the full queue
            try:
                out_q.get_nowait()
            except queue.Empty:
                th.join(0.05)


_TIMELINE = os.environ.get("PFANN_TIMELINE", "0") not in ("0", "")
_T0 = [time.perf_counter()]


class Round(list):
    """What embed_file_batches yields under several ranks: the list is THIS rank's launch group
    [(index, n_seg, embeddings)], `files` = [(index, n_seg)] of every file of the round (all ranks' groups, list order,
    actual segment counts exchanged over the ranks' gloo group) and `groups[r]` = (position in files, count) of rank
    r's group."""
    files = ()
    groups = ()


def gather_round(ranks, rnd, d, device):
    """All ranks' fingerprints of one round on every rank: -> [(index, n_seg, embeddings [n_seg, d] or None)] for every
    file of the round, in list order (one ragged all-gather of the groups' rows; 512 bytes per segment)."""
    from .dist import all_gather_ragged
    counts = [int(sum(n for _, n in rnd.files[p:p + c])) for p, c in rnd.groups]
    rows = [e for _, n, e in rnd if n]
    mine = torch.cat(rows) if len(rows) > 1 else (rows[0] if rows else torch.empty((0, d), device=device, dtype=torch.float32))
    assert mine.shape[0] == counts[ranks.rank]
    allrows = all_gather_ragged(mine, counts, ranks.group)
    out, o = [], 0
    for idx, n in rnd.files:
        out.append((idx, n, allrows[o:o + n] if n else None))
        o += n
    return out


def embed_file_batches(engine, dataset, hop, batch_windows=4096, timer=None, norm=True, workers=None, ahead=None, ranks=None):
    """Yields one list per launch group: [(index, n_seg, embeddings cuda tensor [n_seg, d] or None), ...] in list order;
    a file that fails to load has n_seg = 0 (the reference's 0-segment-song convention, builder.py:82-86).

    ranks (pfann_amd.dist.Ranks, more than one rank): the list is cut into rounds of `world` consecutive launch groups
    and this rank embeds the group with its own number; a `Round` is yielded (see there; gather_round gives every rank
    the whole round).  Every rank runs the same number of rounds.

    Decode runs ahead of the GPU on `workers` host threads (the reference: DataLoader(num_workers=4), builder.py:66;
    PFANN_DECODE_WORKERS): for a MusicDataset the library's native WAV reader fills one pinned slab per group
    (PFANN_NATIVE_WAV=0: Python threads, as for any other dataset); the main thread only issues the asynchronous uploads
    and the launches and never waits for the GPU: stage times of GPU work are taken with events
    (utils.StageTimer.stage_gpu)."""
    timer = timer or StageTimer()
    _T0[0] = time.perf_counter()
    seg = engine.seg_len
    workers = int(os.environ.get("PFANN_DECODE_WORKERS", "8")) if workers is None else workers
    ahead = max(4 * workers, 16) if ahead is None else ahead
    pool = _PinnedPool()
    in_flight = collections.deque()
    # the native reader only stands in for MusicDataset's OWN file reader: a subclass that decodes differently keeps its say
    if isinstance(dataset, MusicDataset) and type(dataset).load_pcm_sr is MusicDataset.load_pcm_sr and workers > 0 and \
            os.environ.get("PFANN_NATIVE_WAV", "1") != "0":
        source = _native_groups(engine, dataset, hop, batch_windows, pool, workers, *((ranks.rank, ranks.world) if ranks else ()))
    elif ranks is not None and ranks.world > 1:
        raise _l.PfannError("a multi-rank run reads its files through the native WAV reader (a MusicDataset, "
                            "PFANN_DECODE_WORKERS > 0, PFANN_NATIVE_WAV not 0)")
    else:
        source = _threaded_groups(engine, dataset, hop, batch_windows, pool, workers, ahead)
    multi = ranks is not None and ranks.world > 1

    for got in source:
        pending, slab, release, t_load = got[:4]
        t1 = time.perf_counter()
        out = [(idx, 0, None) for idx, _, _ in pending]
        wav_all = starts = None
        if slab is not None:
            # uniform group: the slab is the concatenated PCM of the group -- one upload, one conversion
            pcm_all = slab[0][:slab[1]].to(engine.device, non_blocking=True)
            wav_all = torch.empty(slab[1], device=engine.device, dtype=torch.float32)
            _l.check(engine.lib.pfann_pcm16_to_mono(engine.handle, pcm_all.data_ptr(), slab[1], 1, wav_all.data_ptr(),
                                                    _l.current_stream_ptr(engine.device)), "pfann_pcm16_to_mono")
            live = [(it[1], n_seg) for _, n_seg, it in pending if n_seg]
            o = np.asarray([a for a, _ in live], np.int64)
            c = np.asarray([b for _, b in live], np.int64)
            first = np.concatenate([[0], np.cumsum(c)[:-1]])
            starts = np.repeat(o, c) + (np.arange(int(c.sum()), dtype=np.int64) - np.repeat(first, c)) * hop
        else:
            parts, layout, total, short = [], [], 0, False
            for _, n_seg, item in pending:
                if n_seg:
                    have = item[2] if item[0] == "host" else item[1].shape[0]
                    ln = max(have, seg)                                            # musicdata.py:82-84 zero padding
                    short |= ln > have
                    parts.append(total + np.arange(n_seg, dtype=np.int64) * hop)
                    layout.append((item, total))
                    total += ln
            if layout:
                hosts = [(it, off) for it, off in layout if it[0] == "host"]
                wav_all = (torch.zeros if short else torch.empty)(total, device=engine.device, dtype=torch.float32)
                if hosts:
                    pcm_all = (torch.zeros if short else torch.empty)(total, device=engine.device, dtype=torch.int16)
                    m = len(hosts)
                    ptrs = (ctypes.c_void_p * m)(*[it[1].data_ptr() for it, _ in hosts])
                    lens = (ctypes.c_int64 * m)(*[it[2] for it, _ in hosts])
                    offs = (ctypes.c_int64 * m)(*[off for _, off in hosts])
                    _l.check(engine.lib.pfann_pcm16_files_to_mono(engine.handle, ptrs, lens, offs, m, pcm_all.data_ptr(), total,
                                                                  wav_all.data_ptr(), _l.current_stream_ptr(engine.device)),
                             "pfann_pcm16_files_to_mono")
                for it, off in layout:
                    if it[0] == "dev":
                        wav_all[off:off + it[1].shape[0]].copy_(it[1])
                starts = np.concatenate(parts)
        timer.add("load", t_load)                      # one record per launch group (the reference: one per file)
        timer.add("stereo to mono", time.perf_counter() - t1)
        t2 = time.perf_counter()
        if wav_all is not None:
            with timer.stage_gpu("compute embedding"):
                emb = engine.embed_windows(wav_all, starts, norm=norm)
        if _TIMELINE:                                  # PFANN_TIMELINE=1: where the feeding thread's time goes, group by group
            print("timeline group of %d files, %d windows: at %.1f ms, waited %.1f ms for the loader, upload + conversion issued "
                  "in %.1f ms, embed issued in %.1f ms" % (len(pending), sum(n for _, n, _ in pending), 1e3 * (t1 - _T0[0]), 1e3 * t_load,
                                                        1e3 * (t2 - t1), 1e3 * (time.perf_counter() - t2)), file=sys.stderr)
        if wav_all is not None:
            out, o = [], 0
            for idx, n_seg, _ in pending:
                out.append((idx, n_seg, emb[o:o + n_seg] if n_seg else None))
                o += n_seg
        # the pinned buffers of this group are free again once the stream has passed their uploads
        ev = torch.cuda.Event()
        ev.record()
        in_flight.append((ev, release))
        while in_flight and in_flight[0][0].query():
            pool.put(in_flight.popleft()[1])
        # back-pressure: this thread only enqueues work, so without a bound it would run the whole file list ahead of the
        # GPU and pin every group's PCM at once (4.8 GB for 10 k songs); three groups in flight keep the GPU fed
        while len(in_flight) > 3:
            in_flight[0][0].synchronize()
            pool.put(in_flight.popleft()[1])
        timer.resolve()
        if multi:
            # the round's actual segment counts (a read error after a good probe turns a file into a 0-segment song on
            # the rank that read it): summed over the ranks on the host-side gloo group, never behind GPU work
            layout = got[4]
            sizes = [len(pred) for _, pred in layout]
            pos0 = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            vec = np.zeros(int(pos0[-1]), np.int64)
            vec[pos0[ranks.rank]:pos0[ranks.rank] + len(out)] = [n for _, n, _ in out]
            vec = ranks.sum_host(vec)
            rnd = Round(out)
            rnd.files = [(int(layout[r][0]) + j, int(vec[pos0[r] + j])) for r in range(ranks.world) for j in range(sizes[r])]
            rnd.groups = [(int(pos0[r]), sizes[r]) for r in range(ranks.world)]
            yield rnd
        else:
            yield out
    timer.resolve(wait=True)


def embed_files(engine, dataset, hop, batch_windows=4096, timer=None, norm=True, workers=None):
    """Flat form of embed_file_batches: yields (index, n_seg, embeddings cuda tensor [n_seg, d] or None) in list order."""
    for group in embed_file_batches(engine, dataset, hop, batch_windows, timer, norm, workers):
        yield from group


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 3:
        print("Usage: python %s <music list file> <db location>" % argv[0])
        return 1
    from .dist import finish_ranks, init_ranks, self_launch_if_asked
    rc = self_launch_if_asked(argv)         # PFANN_GPUS=N: N ranks of this command, one per GPU
    if rc is not None:
        return rc
    ranks = init_ranks()                    # None: a plain single-process run
    multi = ranks is not None and ranks.world > 1
    rank0 = ranks is None or ranks.rank == 0
    say = print if rank0 else (lambda *a, **k: None)
    file_list_for_db, dir_for_db = argv[1], argv[2]
    configs = argv[3] if len(argv) >= 4 else "configs/default.json"
    if os.path.isdir(configs):                                            # builder.py:38-44
        configs_path = os.path.join(configs, "configs.json")
        params = read_config(configs_path)
        params["model_dir"] = configs
        configs = configs_path
    else:
        params = read_config(configs)
    d = params["model"]["d"]
    if rank0:
        init_logger("builder")                                             # builder.py:27-28

    from .utils import StartupClock
    clock = StartupClock(say)
    say("loading model...")
    max_batch = int(os.environ.get("PFANN_MAX_BATCH", "9728"))
    import gc
    gc.freeze()       # (imports -> the collector's permanent generation: no 50 ms full collection in the middle of the run)
    model_pt = os.path.join(params["model_dir"], "model.pt")
    engine = Engine(params, ranks.device if ranks is not None else 0, max_batch=max_batch)
    # kernel variants of a full launch group for every call: a song's fingerprints -- every byte of `embeddings` -- do not
    # depend on how the list is cut into groups or spread over ranks (include/pfann_amd.h: pfann_set_plan_batch)
    engine.set_plan_batch(max_batch)
    clock.lap("engine")
    if not engine.weights_loaded:             # (else: the start-up thread read model.pt and loaded it while torch was importing)
        engine.load_state_dict(torch.load(model_pt, map_location="cpu"))
    clock.lap("weights")
    engine.warmup(windows=max_batch, group_hop=int(params["sample_rate"] * params["hop_size"]))
    clock.lap("engine warm-up")
    say("model loaded")

    params["indexer"]["frame_shift_mul"] = 1                               # builder.py:64
    dataset = MusicDataset(file_list_for_db, params)
    timer = StageTimer()
    t0 = time.time()
    landmark_key = np.zeros(len(dataset), dtype=np.int32)
    total = 0
    factory = params["indexer"].get("index_factory", "Flat")
    if factory not in ("Flat", "IDMap,Flat"):
        say("index_factory %r is approximate and out of scope: writing an exact flat "
            "inner-product index instead" % factory)
    # Every launch group's fingerprints leave while the next group computes: one asynchronous copy into a pinned buffer on
    # a side stream, and a writer thread puts them at their rows of `embeddings` and of the flat index `landmarkValue`
    # (the reference adds to the index and writes both files after its loop, builder.py:105-136; the bytes are the same).
    # No device-to-host wait in the loop, and nothing left to copy or write when the last group is through.  Several
    # ranks: every rank writes the rows of the groups IT embedded, in place; nothing is funnelled through rank 0.
    p_emb, p_idx = os.path.join(dir_for_db, "embeddings"), os.path.join(dir_for_db, "landmarkValue")
    if rank0:
        os.makedirs(dir_for_db, exist_ok=True)
        open(p_emb, "wb").close()
        with open(p_idx, "wb") as f:
            faissio.write_header(f, d, 0)
    if ranks is not None:
        ranks.barrier()
    fd_emb, fd_idx = os.open(p_emb, os.O_WRONLY), os.open(p_idx, os.O_WRONLY)
    wq = queue.Queue(maxsize=4)
    werr = []

    def pwrite_all(fd, view, off):
        done = 0
        while done < len(view):
            done += os.pwrite(fd, view[done:], off + done)

    def writer():
        while True:
            got = wq.get()
            if got is None:
                return
            if werr:                            # after a failure: keep draining so that the producer never blocks
                continue
            try:
                ev, host, rows, row0 = got
                ev.synchronize()
                if rows:
                    view = memoryview(host[:rows * d].numpy()).cast("B")
                    pwrite_all(fd_emb, view, row0 * d * 4)
                    pwrite_all(fd_idx, view, faissio.HEADER_BYTES + row0 * d * 4)
                free_bufs.put(host)
            except BaseException as x:          # noqa: B902 -- reported by the main thread
                werr.append(x)

    free_bufs = queue.Queue()
    wt = threading.Thread(target=writer, name="pfann-db-writer", daemon=True)
    wt.start()
    side = torch.cuda.Stream(device=engine.device)
    ok = False
    try:
        for items in embed_file_batches(engine, dataset, dataset.hop, batch_windows=max_batch, timer=timer, ranks=ranks):
            row0 = total                          # first row of THIS rank's group
            if multi:
                for r, (p, c) in enumerate(items.groups):
                    n_r = sum(n for _, n in items.files[p:p + c])
                    if r < ranks.rank:
                        row0 += n_r
                for idx, n_seg in items.files:    # every rank keeps the whole key: rank 0 writes it
                    landmark_key[idx] = n_seg
                    total += n_seg
            embs = []
            for idx, n_seg, emb in items:
                if n_seg:
                    embs.append(emb)
                if not multi:
                    landmark_key[idx] = n_seg
                    total += n_seg
            if werr:
                break
            if not embs:
                continue
            g = torch.cat(embs) if len(embs) > 1 else embs[0]
            rows = int(g.shape[0])
            try:
                host = free_bufs.get_nowait()
                if host.numel() < rows * d:
                    host = None
            except queue.Empty:
                host = None
            if host is None:
                host = torch.empty(max(rows, max_batch) * d, dtype=torch.float32, pin_memory=True)
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                host[:rows * d].copy_(g.reshape(-1), non_blocking=True)
                g.record_stream(side)
                ev = torch.cuda.Event()
                ev.record(side)
            wq.put((ev, host, rows, row0))         # (the writer always drains: this cannot block for good)
        with timer.stage("download embeddings"):
            wq.put(None)
            wt.join()
        if werr:
            raise werr[0]
        ok = True
    finally:
        if not ok:                                 # stop the writer, leave no half-written database behind
            if wt.is_alive():
                werr.append(RuntimeError("build aborted"))
                wq.put(None)
                wt.join()
        os.close(fd_emb)
        os.close(fd_idx)
        if not ok and rank0:
            for pth in (p_emb, p_idx):
                try:
                    os.remove(pth)
                except OSError:
                    pass
    if ranks is not None:
        ranks.barrier()                            # every rank's rows are on disk
    if rank0:
        with open(p_idx, "r+b") as f:
            faissio.write_header(f, d, total)
    say("total", total, "embeddings")
    if total == 0:
        say("The database is empty!")
    say("writing database")
    if rank0:
        landmark_key.tofile(os.path.join(dir_for_db, "landmarkKey"))
        shutil.copyfile(file_list_for_db, os.path.join(dir_for_db, "songList.txt"))
        shutil.copyfile(configs, os.path.join(dir_for_db, "configs.json"))
        shutil.copyfile(model_pt, os.path.join(dir_for_db, "model.pt"))
    for name, secs in timer.t.items():                   # one stage per line, the format tools/stat.py:17 parses
        say("%s %.6fs" % (name, secs))
    say("total build time %.3fs" % (time.time() - t0))
    finish_ranks(ranks)
    return 0


if __name__ == "__main__":
    sys.exit(main())
