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


class StageTimer:
    """Accumulates wall time per stage under the reference's stage names
    (tools/stat.py:17: load, resample, stereo to mono, compute embedding, search, rerank,
    output answer, total query time)."""

    def __init__(self):
        self.t = {}

    def add(self, name, dt):
        self.t[name] = self.t.get(name, 0.0) + dt

    class _Ctx:
        def __init__(self, owner, name):
            self.o, self.n = owner, name

        def __enter__(self):
            self.t0 = time.time()

        def __exit__(self, *a):
            self.o.add(self.n, time.time() - self.t0)

    def stage(self, name):
        return StageTimer._Ctx(self, name)
