"""Matcher CLI, drop-in for the reference's matcher.py:
    python matcher.py <query list> <database dir> <result file>

Same argv and outputs (matcher.py:34-42,84,158-163): `<result>` TSV "query\\tanswer",
`<result-stem>_detail.csv` with header query,answer,score,time,part_scores, and
`<result>.bin` with one float32[n_songs,2] (score, time) block per query; load errors give
an "error" row with -inf score and a zero block (matcher.py:94-107).  Queries are embedded,
searched (exact flat IP top-k) and sequence-matched on the MI355X, many queries per launch.
"""
import csv
import ctypes
import os
import sys
import time

import numpy as np
import torch

from .builder import embed_file_batches
from .database import Database
from .engine import Engine
from .musicdata import MusicDataset
from .utils import StageTimer, get_logger, init_logger, read_config


class ResultWriter:
    """The three matcher outputs (matcher.py:40-42,84,94-107,158-163; same in matchemb.py:27-29,55-78)."""

    def __init__(self, result_file, n_songs):
        self.fout = open(result_file, "w", encoding="utf8", newline="\n")
        self.fout2 = open(os.path.splitext(result_file)[0] + "_detail.csv", "w", encoding="utf8", newline="\n")
        self.fout_score = open(result_file + ".bin", "wb")
        self.detail = csv.writer(self.fout2)
        self.detail.writerow(["query", "answer", "score", "time", "part_scores"])
        self.n_songs = n_songs
        self._run = None            # [address, bytes, arrays kept alive] of the pending run of score blocks

    def write(self, name, ans, sco, tim, song_score):
        self.fout.write("%s\t%s\n" % (name, ans))
        self.detail.writerow([name, ans, sco, tim])
        # score blocks of one launch group are consecutive rows of one buffer: they go out as ONE write at flush()
        # (a write per query is a system call and a GIL hand-over per query)
        blk = np.ascontiguousarray(song_score, dtype=np.float32)
        ptr, nb = blk.ctypes.data, blk.nbytes
        if self._run is not None and ptr == self._run[0] + self._run[1]:
            self._run[1] += nb
            self._run[2].append(blk)
        else:
            self._flush_run()
            self._run = [ptr, nb, [blk]]

    def _flush_run(self):
        if self._run is not None:
            ptr, nb, _keep = self._run
            if nb:
                self.fout_score.write((ctypes.c_char * nb).from_address(ptr))
            self._run = None

    def write_error(self, name):
        self.write(name, "error", -1e999, 0, np.zeros([self.n_songs, 2], dtype=np.float32))

    def flush(self):
        self._flush_run()
        self.fout.flush()
        self.fout2.flush()

    def close(self):
        self._flush_run()
        self.fout.close()
        self.fout2.close()
        self.fout_score.close()


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 4:
        print("Usage: python %s <query list> <database dir> <result file>" % argv[0])
        return 1
    file_list_for_query, dir_for_db, result_file = argv[1], argv[2], argv[3]
    params = read_config(os.path.join(dir_for_db, "configs.json"))
    init_logger("matcher")                                                 # matcher.py:31-32

    print("loading model...")
    max_batch = int(os.environ.get("PFANN_MAX_BATCH", "9728"))
    engine = Engine(params, 0, max_batch=max_batch)
    engine.load_state_dict(torch.load(os.path.join(dir_for_db, "model.pt"), map_location="cpu"))
    engine.warmup(windows=max_batch)
    print("model loaded")
    print("loading database...")
    db = Database(dir_for_db, params["indexer"], params["hop_size"], device=0, d=params["model"]["d"])
    db.warmup(rows=max_batch)
    print("database loaded")

    dataset = MusicDataset(file_list_for_query, params)
    timer = StageTimer()
    db.timer = timer
    tm_0 = time.time()
    out = ResultWriter(result_file, len(db.songList))

    def launch(items):
        """items: one launch group of (index, n_seg, emb) in list order -> search + match in flight."""
        good = [(i, n, e) for i, n, e in items if n]
        p = None
        if good:
            emb = torch.cat([e for _, _, e in good])
            qlen = [n for _, n, _ in good]
            qstart = np.concatenate([[0], np.cumsum(qlen)[:-1]])
            p = db.query_launch(emb, qstart, qlen, want_song_scores=True)
        return items, good, p

    def finish(launched):
        items, good, p = launched
        results = {}
        if p is not None:
            for (i, _, _), r in zip(good, db.query_finish(p, reuse_buffers=True)):
                results[i] = r
        with timer.stage("output answer"):
            for i, n, _ in items:
                name = dataset.files[i]
                if n == 0:                                            # matcher.py:94-107
                    out.write_error(name)
                else:
                    sco, (sid, tim), song_score = results[i]
                    out.write(name, db.songList[sid], sco, tim, song_score)   # sid == -1 -> last song (matcher.py:138)
            out.flush()

    # matcher.py:120-126 asks the model for norm=False and L2-normalises on the CPU; the same formula runs inside the
    # projection kernel here.  One launch group = PFANN_MAX_BATCH windows (512 ten-second queries): enough 128x128
    # tiles to fill the chip in every encoder layer.  Group g+1 is decoded, uploaded and launched before group g's
    # results are read back and written.
    in_flight = None
    for items in embed_file_batches(engine, dataset, dataset.hop, batch_windows=max_batch, timer=timer):
        nxt = launch(items)
        if in_flight is not None:
            finish(in_flight)
        in_flight = nxt
    if in_flight is not None:
        finish(in_flight)
    timer.resolve(wait=True)
    out.close()
    for name, secs in timer.t.items():                   # one stage per line, the format tools/stat.py:17 parses
        print("%s %.6fs" % (name, secs))
    get_logger().info("total query time %.6fs", time.time() - tm_0)
    print("total query time %.6fs" % (time.time() - tm_0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
