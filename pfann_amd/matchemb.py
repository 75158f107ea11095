"""Matcher over pre-extracted embeddings, drop-in for the reference's matchemb.py:
    python matchemb.py <query embedding dir> <database dir> <result file>
Reads `query_embeddings` / `query_index` / `queryList.txt` (matchemb.py:33,47-51), searches and
sequence-matches on the MI355X in batches, writes the matcher's three outputs.

Zero-row queries (query_index row (pos, 0): the file failed to load at extraction, extractemb.py:70-73):
the reference hands the empty array to Database.query_embeddings (matchemb.py:60-66), where
np.concatenate([]) at database.py:140 raises ValueError and the run aborts -- it has no defined output for
them.  This tool instead writes the row matcher.py itself writes for the same unreadable file
(matcher.py:94-107: answer "error", score -inf, time 0, a zero score block) and carries on."""
import os
import sys
import time

import numpy as np
import torch

from .database import Database
from .dist import finish_ranks, init_ranks, self_launch_if_asked
from .matcher import ResultWriter
from .utils import read_config, read_file_list


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 4:
        print("Usage: python %s <query embedding dir> <database dir> <result file>" % argv[0])
        return 1
    rc = self_launch_if_asked(argv)         # PFANN_GPUS=N: N ranks of this command, one per GPU
    if rc is not None:
        return rc
    ranks = init_ranks()
    rank0 = ranks is None or ranks.rank == 0
    say = print if rank0 else (lambda *a, **k: None)
    dir_for_query, dir_for_db, result_file = argv[1], argv[2], argv[3]
    import gc
    gc.freeze()       # (imports -> the collector's permanent generation: no 50 ms full collection in the middle of the run)
    params = read_config(os.path.join(dir_for_db, "configs.json"))
    file_list = read_file_list(os.path.join(dir_for_query, "queryList.txt"))
    d = params["model"]["d"]
    say("loading database...")
    db = Database(dir_for_db, params["indexer"], params["hop_size"], device=ranks.device if ranks is not None else 0, d=d,
                  ranks=ranks)
    say("database loaded")
    # every rank reads the (small) query files: 512 bytes per segment; the database is what is sharded
    q = np.fromfile(os.path.join(dir_for_query, "query_embeddings"), dtype=np.float32).reshape([-1, d])
    query_index = np.fromfile(os.path.join(dir_for_query, "query_index"), dtype=np.int64).reshape([-1, 2])
    assert query_index.shape[0] == len(file_list)
    tm_0 = time.time()
    out = ResultWriter(result_file, len(db.songList), ranks=ranks, n_queries=len(file_list), song_range=db.song_range)
    group = int(os.environ.get("PFANN_QUERY_GROUP", "512")) * (ranks.world if ranks is not None else 1)
    qdev = torch.as_tensor(q).to(db.index.device)
    for g0 in range(0, len(file_list), group):
        ids = list(range(g0, min(g0 + group, len(file_list))))
        good = [i for i in ids if query_index[i, 1] > 0]
        res = {}
        if good:
            rows = torch.cat([qdev[query_index[i, 0]: query_index[i, 0] + query_index[i, 1]] for i in good])
            qlen = [int(query_index[i, 1]) for i in good]
            qstart = np.concatenate([[0], np.cumsum(qlen)[:-1]])
            done = g0
            for j0, j1, p in db.query_launch_chunks(rows, qstart, qlen, want_song_scores=True):
                for i, r in zip(good[j0:j1], db.query_finish(p, reuse_buffers=True)):
                    res[i] = r
                upto = good[j1 - 1] + 1 if j1 < len(good) else ids[-1] + 1
                _write(out, file_list, db, res, range(done, upto))
                done = upto
        else:
            _write(out, file_list, db, res, ids)
    out.close()
    if ranks is not None:
        ranks.barrier()
    say("total query time %.6fs" % (time.time() - tm_0))
    finish_ranks(ranks)
    return 0


def _write(out, file_list, db, res, ids):
    for i in ids:
        if i in res:
            sco, (sid, tim), song_score = res[i]
            out.write(file_list[i], db.songList[sid], sco, tim, song_score, qi=i)
        else:
            out.write_error(file_list[i], qi=i)
    out.flush()
    res.clear()


if __name__ == "__main__":
    sys.exit(main())
