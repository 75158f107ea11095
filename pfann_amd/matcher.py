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
import gc
import os
import sys
import threading
import time

import numpy as np
import torch

from .builder import embed_file_batches, gather_round
from .database import Database
from .dist import finish_ranks, init_ranks, self_launch_if_asked
from .engine import Engine
from .musicdata import MusicDataset
from .utils import StageTimer, StartupClock, get_logger, init_logger, read_config


class ResultWriter:
    """The three matcher outputs (matcher.py:40-42,84,94-107,158-163; same in matchemb.py:27-29,55-78).

    ranks (a song-sharded job, pfann_amd.dist.Ranks): rank 0 writes the two text files; the `.bin` matrix
    [n_queries, n_songs, 2] is created at its final size by rank 0 and every rank writes ITS songs' columns of each
    query's block in place (pwrite) -- the shards' score blocks never travel to another rank, and the file ends up
    byte-identical to the one a single process appends (error rows stay the zeros the file was created with)."""

    def __init__(self, result_file, n_songs, ranks=None, n_queries=0, song_range=None):
        self.n_songs = n_songs
        self.sharded = ranks is not None and ranks.sharded
        self.text = ranks is None or ranks.rank == 0
        self._run = None            # [address, bytes, arrays kept alive] of the pending run of score blocks
        if self.text:
            self.fout = open(result_file, "w", encoding="utf8", newline="\n")
            self.fout2 = open(os.path.splitext(result_file)[0] + "_detail.csv", "w", encoding="utf8", newline="\n")
            self.detail = csv.writer(self.fout2)
            self.detail.writerow(["query", "answer", "score", "time", "part_scores"])
        if not self.sharded:
            self.fout_score = open(result_file + ".bin", "wb")
        else:
            if ranks.rank == 0:
                with open(result_file + ".bin", "wb") as f:
                    f.truncate(int(n_queries) * n_songs * 8)
            ranks.barrier()
            self.fd = os.open(result_file + ".bin", os.O_WRONLY)
            self.lo, self.hi = song_range

    def write(self, name, ans, sco, tim, song_score, qi=None):
        if self.text:
            self.fout.write("%s\t%s\n" % (name, ans))
            self.detail.writerow([name, ans, sco, tim])
        if self.sharded:
            if self.hi > self.lo:                   # this shard's columns of query qi's block
                blk = memoryview(np.ascontiguousarray(song_score, dtype=np.float32)).cast("B")
                off, done = (int(qi) * self.n_songs + self.lo) * 8, 0
                while done < len(blk):
                    done += os.pwrite(self.fd, blk[done:], off + done)
            return
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

    def write_error(self, name, qi=None):
        if self.sharded:                            # the block is already zero
            if self.text:
                self.fout.write("%s\t%s\n" % (name, "error"))
                self.detail.writerow([name, "error", -1e999, 0])
            return
        self.write(name, "error", -1e999, 0, np.zeros([self.n_songs, 2], dtype=np.float32))

    def flush(self):
        self._flush_run()
        if self.text:
            self.fout.flush()
            self.fout2.flush()

    def close(self):
        self._flush_run()
        if self.text:
            self.fout.close()
            self.fout2.close()
        if self.sharded:
            os.close(self.fd)
        else:
            self.fout_score.close()


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 4:
        print("Usage: python %s <query list> <database dir> <result file>" % argv[0])
        return 1
    rc = self_launch_if_asked(argv)         # PFANN_GPUS=N: N ranks of this command, one per GPU
    if rc is not None:
        return rc
    ranks = init_ranks()                    # None: a plain single-process run
    multi = ranks is not None and ranks.world > 1
    rank0 = ranks is None or ranks.rank == 0
    dev = ranks.device if ranks is not None else 0
    say = print if rank0 else (lambda *a, **k: None)
    file_list_for_query, dir_for_db, result_file = argv[1], argv[2], argv[3]
    params = read_config(os.path.join(dir_for_db, "configs.json"))
    if rank0:
        init_logger("matcher")                                             # matcher.py:31-32

    clock = StartupClock(say)
    say("loading model...")
    max_batch = int(os.environ.get("PFANN_MAX_BATCH", "9728"))
    # the database (file read, upload, fp16 copy: host- and copy-engine work) loads on a thread while this one builds the
    # engine and warms it up (weights, workspace allocation, first launch of every kernel)
    db_box = []

    def load_db():
        try:
            db_box.append(Database(dir_for_db, params["indexer"], params["hop_size"], device=dev, d=params["model"]["d"],
                                   ranks=ranks))
        except BaseException as x:                  # noqa: B902 -- re-raised on the main thread
            db_box.append(x)
    db_thread = threading.Thread(target=load_db, name="pfann-db-load")
    db_thread.start()
    # everything imported so far (torch: a million objects) goes to the collector's permanent generation: a full collection
    # in the middle of the run -- it fell between the first and the second launch group, every time -- took 52 ms
    gc.freeze()
    dataset = MusicDataset(file_list_for_query, params)
    engine = Engine(params, dev, max_batch=max_batch)
    # kernel variants of a full launch group for every call: a query's fingerprints -- and with them every byte of the
    # three output files -- do not depend on how the list is cut into groups or spread over ranks
    engine.set_plan_batch(max_batch)
    clock.lap("engine")
    if not engine.weights_loaded:             # (else: the start-up thread read model.pt and loaded it while torch was importing)
        engine.load_state_dict(torch.load(os.path.join(dir_for_db, "model.pt"), map_location="cpu"))
    clock.lap("weights")
    # a short list never fills a launch group: warm up (and size the search workspace) for what will really come
    warm = min(max_batch, max(64, len(dataset) * int(os.environ.get("PFANN_WARMUP_SEGMENTS_PER_FILE", "19"))))
    engine.warmup(windows=warm, group_hop=dataset.hop)
    clock.lap("engine warm-up")
    say("model loaded")
    say("loading database...")
    db_thread.join()
    if isinstance(db_box[0], BaseException):
        raise db_box[0]
    db = db_box[0]
    db.attach_engine(engine)
    clock.lap("database (rest of its load after the engine was ready)")
    db.warmup(rows=warm * (ranks.world if multi else 1))
    clock.lap("database warm-up")
    say("database loaded")
    timer = StageTimer()
    db.timer = timer
    tm_0 = time.time()
    out = ResultWriter(result_file, len(db.songList), ranks=ranks, n_queries=len(dataset), song_range=db.song_range)

    def launch(items):
        """items: one launch group of (index, n_seg, emb) in list order -> search + match in flight."""
        good = [(i, n, e) for i, n, e in items if n]
        ps = []
        if good:
            emb = torch.cat([e for _, _, e in good])
            qlen = [n for _, n, _ in good]
            qstart = np.concatenate([[0], np.cumsum(qlen)[:-1]])
            ps = db.query_launch_chunks(emb, qstart, qlen, want_song_scores=True)
        return items, good, ps

    def finish(launched):
        items, good, ps = launched
        it = iter(items)
        for j0, j1, p in ps:
            results = {}
            for (i, _, _), r in zip(good[j0:j1], db.query_finish(p, reuse_buffers=True)):
                results[i] = r
            last = good[j1 - 1][0]
            with timer.stage("output answer"):
                for i, n, _ in it:                                        # list order, error rows in their places
                    write_one(i, n, results)
                    if i == last:
                        break
                out.flush()
        with timer.stage("output answer"):
            for i, n, _ in it:                                            # what follows the last query with segments
                write_one(i, n, {})
            out.flush()

    def write_one(i, n, results):
        name = dataset.files[i]
        if n == 0:                                                        # matcher.py:94-107
            out.write_error(name, qi=i)
        else:
            sco, (sid, tim), song_score = results[i]
            out.write(name, db.songList[sid], sco, tim, song_score, qi=i)   # sid == -1 -> last song (matcher.py:138)

    # matcher.py:120-126 asks the model for norm=False and L2-normalises on the CPU; the same formula runs inside the
    # projection kernel here.  One launch group = PFANN_MAX_BATCH windows (512 ten-second queries): enough 128x128
    # tiles to fill the chip in every encoder layer.  Group g+1 is decoded, uploaded and launched before group g's
    # results are read back and written.  Several ranks: a round = one group per rank, embedded where it was read,
    # all-gathered, then searched by every rank in its own shard of the database.
    in_flight = None
    for items in embed_file_batches(engine, dataset, dataset.hop, batch_windows=max_batch, timer=timer, ranks=ranks):
        if multi:
            items = gather_round(ranks, items, engine.d, engine.device)
        _t = time.perf_counter()
        nxt = launch(items)
        _t1 = time.perf_counter()
        if in_flight is not None:
            finish(in_flight)
        if os.environ.get("PFANN_TIMELINE"):
            print("timeline matcher: launch %.1f ms, finish of the previous group %.1f ms" % (1e3 * (_t1 - _t), 1e3 * (time.perf_counter() - _t1)), file=sys.stderr)
        in_flight = nxt
    if in_flight is not None:
        finish(in_flight)
    timer.resolve(wait=True)
    out.close()
    if ranks is not None:
        ranks.barrier()                                  # every rank's columns of the score matrix are on disk
    for name, secs in timer.t.items():                   # one stage per line, the format tools/stat.py:17 parses
        say("%s %.6fs" % (name, secs))
    if rank0:
        get_logger().info("total query time %.6fs", time.time() - tm_0)
    say("total query time %.6fs" % (time.time() - tm_0))
    finish_ranks(ranks)
    return 0


if __name__ == "__main__":
    sys.exit(main())
