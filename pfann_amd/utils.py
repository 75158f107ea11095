"""Small host helpers mirroring the reference's simpleutils.py interface
(read_config simpleutils.py:30-32, read_file_list simpleutils.py:34-47)."""
import csv
import json
import logging
import time


def read_config(path):
    with open(path, "r") as fin:
        return json.load(fin)


def read_file_list(list_file):
    """Text list: one path per line; .csv list: first column, header skipped."""
    if list_file.endswith(".csv"):
        with open(list_file, "r") as fin:
            rows = list(csv.reader(fin))
        return [r[0] for r in rows[1:]]
    with open(list_file, "r", encoding="utf8") as fin:
        return [ln[:-1] if ln.endswith("\n") else ln for ln in fin]


_logger = logging.getLogger("pfann_amd")


def get_logger():
    return _logger


def init_logger(app_name):
    """logs/<app>-<date>.log with the reference's record format (simpleutils.py:72-85), so the reference's
    tools/stat.py:17 can aggregate the stage lines this package logs ('<stage> <seconds>s')."""
    import datetime
    import os
    os.makedirs("logs", exist_ok=True)
    name = app_name + "-" + datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
    _logger.setLevel(logging.INFO)
    handler = logging.FileHandler("logs/%s.log" % name, encoding="utf8")
    handler.setFormatter(logging.Formatter("[%(asctime)s] [%(processName)s/%(levelname)s] %(message)s"))
    _logger.addHandler(handler)
    return "logs/%s.log" % name


class StageTimer:
    """Wall time per stage under the reference's stage names (tools/stat.py:17: load, resample, stereo to mono,
    compute embedding, search, rerank, output answer, total query time).  Every measurement is also logged as
    its own '<stage> %.6fs' record (the reference logs them per file, musicdata.py:70,91, builder.py:103,
    matcher.py:131,165, database.py:165; here a record covers one batch of files), one stage per line."""

    def __init__(self):
        self.t = {}

    def add(self, name, dt):
        self.t[name] = self.t.get(name, 0.0) + dt
        _logger.info("%s %.6fs", name, dt)

    class _Ctx:
        def __init__(self, owner, name):
            self.o, self.n = owner, name

        def __enter__(self):
            self.t0 = time.time()

        def __exit__(self, *a):
            self.o.add(self.n, time.time() - self.t0)

    def stage(self, name):
        return StageTimer._Ctx(self, name)

    # ---- stages that run asynchronously on the GPU: timed with events on the launching stream, so that the host never
    # has to wait for a stage just to log it (the next batch's uploads and launches go out while this one computes);
    # resolve() turns the finished pairs into the same '<stage> <seconds>s' records
    class _GpuCtx:
        def __init__(self, owner, name):
            self.o, self.n = owner, name

        def __enter__(self):
            import torch
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

        def __exit__(self, *a):
            self.e1.record()
            self.o._pending.append((self.n, self.e0, self.e1))

    def stage_gpu(self, name):
        if not hasattr(self, "_pending"):
            self._pending = []
        return StageTimer._GpuCtx(self, name)

    def mark_gpu(self, name, e0, e1):
        """a stage bracketed by two already recorded timing events"""
        if not hasattr(self, "_pending"):
            self._pending = []
        self._pending.append((name, e0, e1))

    def resolve(self, wait=False):
        """account every event pair that has completed (all of them with wait=True)"""
        keep = []
        for name, e0, e1 in getattr(self, "_pending", []):
            if wait:
                e1.synchronize()
            if e1.query():
                self.add(name, e0.elapsed_time(e1) * 1e-3)
            else:
                keep.append((name, e0, e1))
        self._pending = keep


class StartupClock:
    """PFANN_STARTUP_TIMING=1: the CLIs print one 'startup <what> <seconds>s' line per start-up stage (tools/cli_bench.py
    reads them; PFANN_T0 = the wall-clock time the process was spawned at, so that interpreter start + imports are a stage
    too).  Off by default: the reference's tools print nothing of the kind."""

    def __init__(self, say=print):
        import os
        self.on = os.environ.get("PFANN_STARTUP_TIMING", "0") not in ("0", "")
        self.say = say
        self.t = time.time()
        t0 = os.environ.get("PFANN_T0")
        if self.on and t0:
            self.say("startup interpreter and imports %.3fs" % (self.t - float(t0)))

    def lap(self, what):
        now = time.time()
        if self.on:
            self.say("startup %s %.3fs" % (what, now - self.t))
        self.t = now

