"""Database builder CLI, drop-in for the reference's builder.py:
    python builder.py <music list file> <db location> [config.json | model dir]

Same argv, same db directory layout (`embeddings`, `landmarkValue`, `landmarkKey`,
`songList.txt`, `configs.json`, `model.pt`; reference builder.py:30-44,71,136-148), but
the per-song loop (builder.py:75-103) runs on the MI355X: PCM is uploaded as int16,
converted/segmented/mel-ed/encoded by the HIP kernels, several songs per launch, and the
fingerprints only come back to the host once, to be written to disk.
"""
import os
import shutil
import sys
import time

import numpy as np
import torch

from . import faissio
from .engine import Engine
from .musicdata import MusicDataset
from .utils import StageTimer, init_logger, read_config


def embed_files(engine, dataset, hop, batch_windows=4096, timer=None, norm=True):
    """Yields (index, n_seg, embeddings cuda tensor [n_seg, d]) in list order; a file that
    fails to load yields n_seg = 0 (the reference's 0-segment-song convention,
    builder.py:82-86)."""
    timer = timer or StageTimer()
    seg = engine.seg_len
    pending = []          # (index, n_seg, mono wav device tensor)
    n_win = 0

    def flush():
        nonlocal pending, n_win
        if not pending:
            return []
        with timer.stage("compute embedding"):
            wavs, starts, base = [], [], 0
            for _, n_seg, w in pending:
                if n_seg:
                    wavs.append(w)
                    starts.append(base + np.arange(n_seg, dtype=np.int64) * hop)
                    base += w.shape[0]
            out = []
            if wavs:
                emb = engine.embed_windows(torch.cat(wavs), np.concatenate(starts), norm=norm)
                o = 0
                for idx, n_seg, _ in pending:
                    out.append((idx, n_seg, emb[o:o + n_seg]))
                    o += n_seg
            else:
                out = [(idx, 0, None) for idx, _, _ in pending]
        pending, n_win = [], 0
        return out

    for i in range(len(dataset)):
        try:
            with timer.stage("load"):
                pcm = dataset.load_pcm(i)
            with timer.stage("stereo to mono"):
                wav = engine.pcm16_to_mono(pcm, sample_rate=getattr(dataset, "last_sample_rate", None))
                if wav.shape[0] < seg:                                    # musicdata.py:82-84
                    wav = torch.nn.functional.pad(wav, (0, seg - wav.shape[0]))
            n_seg = (wav.shape[0] - seg) // hop + 1
        except Exception as x:                                            # musicdata.py:95-101
            print("load %s error! (%s)" % (dataset.files[i], x))
            wav, n_seg = None, 0
        pending.append((i, n_seg, wav))
        n_win += n_seg
        if n_win >= batch_windows:
            yield from flush()
    yield from flush()


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
    engine = Engine(params, 0, max_batch=int(os.environ.get("PFANN_MAX_BATCH", "4096")))
    model_pt = os.path.join(params["model_dir"], "model.pt")
    engine.load_state_dict(torch.load(model_pt, map_location="cpu"))
    print("model loaded")

    params["indexer"]["frame_shift_mul"] = 1                               # builder.py:64
    dataset = MusicDataset(file_list_for_db, params)
    os.makedirs(dir_for_db, exist_ok=True)
    timer = StageTimer()
    t0 = time.time()
    landmark_key = np.zeros(len(dataset), dtype=np.int32)
    chunks = []
    total = 0
    for idx, n_seg, emb in embed_files(engine, dataset, dataset.hop, timer=timer):
        landmark_key[idx] = n_seg
        if n_seg:
            chunks.append(emb.cpu().numpy())
            total += n_seg
    embeddings = np.concatenate(chunks) if chunks else np.zeros((0, d), np.float32)
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
