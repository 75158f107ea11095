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

    def get(self, n):
        cap = 1 << max(12, int(n - 1).bit_length())
        with self.lock:
            lst = self.free.get(cap)
            if lst:
                return lst.pop()
        return torch.empty(cap, dtype=torch.int16).pin_memory()

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


def embed_file_batches(engine, dataset, hop, batch_windows=4096, timer=None, norm=True, workers=None, ahead=None):
    """Yields one list per launch group: [(index, n_seg, embeddings cuda tensor [n_seg, d] or None), ...] in list order;
    a file that fails to load has n_seg = 0 (the reference's 0-segment-song convention, builder.py:82-86).

    Decode runs ahead of the GPU on `workers` host threads (the reference: DataLoader(num_workers=4), builder.py:66;
    PFANN_DECODE_WORKERS) that read each file into pinned memory; the main thread only issues the asynchronous upload
    and the launches, and never waits for the GPU: stage times of GPU work are taken with events
    (utils.StageTimer.stage_gpu)."""
    timer = timer or StageTimer()
    seg = engine.seg_len
    native = int(engine.params["sample_rate"])
    workers = int(os.environ.get("PFANN_DECODE_WORKERS", "8")) if workers is None else workers
    ahead = max(4 * workers, 16) if ahead is None else ahead
    pool = _PinnedPool()
    pending, held, in_flight = [], [], collections.deque()
    n_win = 0
    t_load = t_mono = 0.0

    def flush():
        """pending: (index, n_seg, item) with item = ("host", pinned int16 view [n] (mono, native rate), n) -- uploaded and
        converted here, all files of the group by ONE library call -- or ("dev", float32 mono wav on the device) for
        files that needed their own path (stereo, another sample rate, PCM handed over as a tensor), or None (error)."""
        nonlocal pending, held, n_win, t_load, t_mono
        if not pending:
            return None
        t1 = time.perf_counter()
        starts, layout, total, short = [], [], 0, False
        for _, n_seg, item in pending:
            if n_seg:
                ln = max(item[2] if item[0] == "host" else item[1].shape[0], seg)          # musicdata.py:82-84 zero padding
                short |= ln > (item[2] if item[0] == "host" else item[1].shape[0])
                starts.append(total + np.arange(n_seg, dtype=np.int64) * hop)
                layout.append((item, total))
                total += ln
        out = [(idx, 0, None) for idx, _, _ in pending]
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
        t_mono += time.perf_counter() - t1
        timer.add("load", t_load)                      # one record per launch group (the reference: one per file)
        timer.add("stereo to mono", t_mono)
        t_load = t_mono = 0.0
        if layout:
            with timer.stage_gpu("compute embedding"):
                emb = engine.embed_windows(wav_all, np.concatenate(starts), norm=norm)
            out, o = [], 0
            for idx, n_seg, _ in pending:
                out.append((idx, n_seg, emb[o:o + n_seg] if n_seg else None))
                o += n_seg
        # the pinned buffers of this group are free again once the stream has passed their uploads
        ev = torch.cuda.Event()
        ev.record()
        in_flight.append((ev, held))
        while in_flight and in_flight[0][0].query():
            pool.put(in_flight.popleft()[1])
        timer.resolve()
        pending, held, n_win = [], [], 0
        return out

    n = len(dataset)
    ex = ThreadPoolExecutor(max_workers=max(workers, 1)) if workers > 0 else None
    futs = collections.deque()
    nxt = 0
    try:
        for i in range(n):
            while ex is not None and nxt < n and nxt - i < ahead:
                futs.append(ex.submit(_decode, dataset, nxt, pool, native))
                nxt += 1
            try:
                t0 = time.perf_counter()                # "load": what the GPU-feeding thread waited for the decoders
                pcm, sr, buf = futs.popleft().result() if ex is not None else _decode(dataset, i, pool, native)
                t1 = time.perf_counter()
                t_load += t1 - t0
                if buf is not None:
                    held.append(buf)
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
                t_mono += time.perf_counter() - t1
            except Exception as x:                                            # musicdata.py:95-101
                print("load %s error! (%s)" % (dataset.files[i], x))
                item, n_seg = None, 0
            if pending and n_win + n_seg > batch_windows:     # a group never exceeds the encoder's chunk (no small tail pass)
                yield flush()
            pending.append((i, n_seg, item))
            n_win += n_seg
            if n_win >= batch_windows:
                yield flush()
        out = flush()
        if out is not None:
            yield out
    finally:
        if ex is not None:
            ex.shutdown(wait=True, cancel_futures=True)
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
    init_logger("builder")                                                 # builder.py:27-28

    print("loading model...")
    max_batch = int(os.environ.get("PFANN_MAX_BATCH", "9728"))
    engine = Engine(params, 0, max_batch=max_batch)
    model_pt = os.path.join(params["model_dir"], "model.pt")
    engine.load_state_dict(torch.load(model_pt, map_location="cpu"))
    engine.warmup()
    print("model loaded")

    params["indexer"]["frame_shift_mul"] = 1                               # builder.py:64
    dataset = MusicDataset(file_list_for_db, params)
    os.makedirs(dir_for_db, exist_ok=True)
    timer = StageTimer()
    t0 = time.time()
    landmark_key = np.zeros(len(dataset), dtype=np.int32)
    chunks = []
    total = 0
    # the fingerprints stay in HBM until every file is through (1 M rows = 0.5 GB) and come back in one copy: no
    # device-to-host wait inside the loop, so the next group's uploads and launches go out while this one computes
    for idx, n_seg, emb in embed_files(engine, dataset, dataset.hop, batch_windows=max_batch, timer=timer):
        landmark_key[idx] = n_seg
        if n_seg:
            chunks.append(emb)
            total += n_seg
    with timer.stage("download embeddings"):
        embeddings = torch.cat(chunks).cpu().numpy() if chunks else np.zeros((0, d), np.float32)
    embeddings.tofile(os.path.join(dir_for_db, "embeddings"))
    print("total", total, "embeddings")
    if total == 0:
        print("The database is empty!")

    factory = params["indexer"].get("index_factory", "Flat")
    if factory not in ("Flat", "IDMap,Flat"):
        print("index_factory %r is approximate and out of scope: writing an exact flat "
              "inner-product index instead" % factory)
    print("writing database")
    faissio.write_index_flat(os.path.join(dir_for_db, "landmarkValue"), embeddings)
    landmark_key.tofile(os.path.join(dir_for_db, "landmarkKey"))
    shutil.copyfile(file_list_for_db, os.path.join(dir_for_db, "songList.txt"))
    shutil.copyfile(configs, os.path.join(dir_for_db, "configs.json"))
    shutil.copyfile(model_pt, os.path.join(dir_for_db, "model.pt"))
    for name, secs in timer.t.items():                   # one stage per line, the format tools/stat.py:17 parses
        print("%s %.6fs" % (name, secs))
    print("total build time %.3fs" % (time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
