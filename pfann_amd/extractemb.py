"""Query-embedding extractor, drop-in for the reference's extractemb.py:
    python extractemb.py <query list> <database dir> <output embedding dir>
Writes `query_embeddings` (raw float32 [*, d], unit-norm rows), `query_index` (int64 [n, 2] =
start row, row count; a file that fails to load gets (pos, 0)), `queryList.txt`, `configs.json`
(extractemb.py:57-95).  Embedding runs on the MI355X, many files per launch."""
import os
import shutil
import sys

import numpy as np
import torch

from .builder import embed_file_batches, gather_round
from .dist import finish_ranks, init_ranks, self_launch_if_asked
from .engine import Engine
from .musicdata import MusicDataset
from .utils import StageTimer, read_config


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 4:
        print("Usage: python %s <query list> <database dir> <output embedding dir>" % argv[0])
        return 1
    rc = self_launch_if_asked(argv)         # PFANN_GPUS=N: N ranks of this command, one per GPU
    if rc is not None:
        return rc
    ranks = init_ranks()
    multi = ranks is not None and ranks.world > 1
    rank0 = ranks is None or ranks.rank == 0
    say = print if rank0 else (lambda *a, **k: None)
    file_list_for_query, dir_for_db, out_embed_dir = argv[1], argv[2], argv[3]
    configs = os.path.join(dir_for_db, "configs.json")
    import gc
    gc.freeze()       # (imports -> the collector's permanent generation: no 50 ms full collection in the middle of the run)
    params = read_config(configs)
    say("loading model...")
    max_batch = int(os.environ.get("PFANN_MAX_BATCH", "9728"))
    engine = Engine(params, ranks.device if ranks is not None else 0, max_batch=max_batch)
    engine.set_plan_batch(max_batch)         # the same bits as matcher.py / builder.py give the same file, however it is grouped
    if not engine.weights_loaded:             # (else: the start-up thread has read and loaded model.pt already)
        engine.load_state_dict(torch.load(os.path.join(dir_for_db, "model.pt"), map_location="cpu"))
    say("model loaded")
    dataset = MusicDataset(file_list_for_query, params)
    if rank0:
        os.makedirs(out_embed_dir, exist_ok=True)
    timer = StageTimer()
    idx_pos = 0
    index = np.zeros((len(dataset), 2), dtype=np.int64)
    fe = open(os.path.join(out_embed_dir, "query_embeddings"), "wb") if rank0 else None
    # several ranks: a round = one launch group per rank, embedded where it was read; the rows are all-gathered (512
    # bytes per segment) and rank 0 writes them in list order
    for group in embed_file_batches(engine, dataset, dataset.hop, batch_windows=max_batch, timer=timer, ranks=ranks):
        if multi:
            group = gather_round(ranks, group, engine.d, engine.device)
        rows = [emb for _, n_seg, emb in group if n_seg]
        if rows and rank0:
            fe.write(torch.cat(rows).cpu().numpy().tobytes())       # one device-to-host copy per launch group
        for i, n_seg, _ in group:
            index[i] = (idx_pos, n_seg)
            idx_pos += n_seg
    if rank0:
        fe.close()
        index.tofile(os.path.join(out_embed_dir, "query_index"))
        print("total", idx_pos, "embeddings")
        shutil.copyfile(file_list_for_query, os.path.join(out_embed_dir, "queryList.txt"))
        shutil.copyfile(configs, os.path.join(out_embed_dir, "configs.json"))
    finish_ranks(ranks)
    return 0


if __name__ == "__main__":
    sys.exit(main())
