"""Database builder CLI, drop-in for the reference's builder.py:
    python builder.py <music list file> <db location> [config.json | model dir]

Same argv, same db directory layout (`embeddings`, `landmarkValue`, `landmarkKey`,
`songList.txt`, `configs.json`, `model.pt`; reference builder.py:30-44,71,136-148), but
the per-song loop (builder.py:75-103) runs on the MI355X: PCM is uploaded as int16,
converted/segmented/mel-ed/encoded by the HIP kernels, several songs per launch, and the
fingerprints only come back to the host once, to be written to disk.
"""
import collections
import os
import shutil
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import faissio
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
    """Runs on a decode worker: file -> int16 PCM in a pinned buffer.  -> (pcm [n, ch] pinned view, rate, buffer)."""
    if hasattr(dataset, "load_pcm_sr"):
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
    workers = int(os.environ.get("PFANN_DECODE_WORKERS", "4")) if workers is None else workers
    ahead = max(4 * workers, 16) if ahead is None else ahead
    pool = _PinnedPool()
    pending, held, in_flight = [], [], collections.deque()
    n_win = 0
    t_load = t_mono = 0.0

    def flush():
        nonlocal pending, held, n_win, t_load, t_mono
        if not pending:
            return None
        timer.add("load", t_load)                      # one record per launch group (the reference: one per file)
        timer.add("stereo to mono", t_mono)
        t_load = t_mono = 0.0
        with timer.stage_gpu("compute embedding"):
            wavs, starts, base = [], [], 0
            for _, n_seg, w in pending:
                if n_seg:
                    wavs.append(w)
                    starts.append(base + np.arange(n_seg, dtype=np.int64) * hop)
                    base += w.shape[0]
            out = []
            if wavs:
                emb = engine.embed_windows(torch.cat(wavs) if len(wavs) > 1 else wavs[0], np.concatenate(starts), norm=norm)
                o = 0
                for idx, n_seg, _ in pending:
                    out.append((idx, n_seg, emb[o:o + n_seg] if n_seg else None))
                    o += n_seg
            else:
                out = [(idx, 0, None) for idx, _, _ in pending]
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
                wav = engine.pcm16_to_mono(pcm, sample_rate=sr)
                if wav.shape[0] < seg:                                        # musicdata.py:82-84
                    wav = torch.nn.functional.pad(wav, (0, seg - wav.shape[0]))
                n_seg = (wav.shape[0] - seg) // hop + 1
                t_mono += time.perf_counter() - t1
            except Exception as x:                                            # musicdata.py:95-101
                print("load %s error! (%s)" % (dataset.files[i], x))
                wav, n_seg = None, 0
            if pending and n_win + n_seg > batch_windows:     # a group never exceeds the encoder's chunk (no small tail pass)
                yield flush()
            pending.append((i, n_seg, wav))
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
