"""Reader/writer for the faiss on-disk format of an IndexFlatIP/IndexFlatL2 ("landmarkValue",
reference builder.py:136, database.py:88).  faiss is un-vendored and absent here, so this
follows the published layout of faiss' `write_index` for IndexFlat (index_write.cpp:
fourcc "IxFI"/"IxF2", header {d:int32, ntotal:int64, dummy:int64 x2, is_trained:uint8,
metric_type:int32}, then a size_t-prefixed float vector).  PARITY UNPINNED: not checked
against a real faiss build (none is installable); the builder also always writes the raw
`embeddings` file, which the reference itself falls back to (database.py:96-97).
"""
import os
import struct

import numpy as np

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


def _header(d, n, metric):
    return ((b"IxFI" if metric == METRIC_INNER_PRODUCT else b"IxF2") + struct.pack("<iqqqBi", d, n, 1 << 20, 1 << 20, 1, metric) +
            struct.pack("<Q", n * d))


def _write_rows(f, xb):
    if xb.size:                                   # (a memoryview of an empty array cannot be cast)
        f.write(memoryview(xb).cast("B"))


def write_index_flat(path, xb, metric=METRIC_INNER_PRODUCT):
    xb = np.ascontiguousarray(xb, dtype="<f4")
    n, d = xb.shape
    with open(path, "wb") as f:
        f.write(_header(d, n, metric))
        _write_rows(f, xb)


class FlatIndexWriter:
    """The same file written incrementally (the builder appends each launch group's rows while the next group computes):
    header with n = 0 first, rows appended, the header rewritten with the final count on close."""

    def __init__(self, path, d, metric=METRIC_INNER_PRODUCT):
        self.d, self.n, self.metric = int(d), 0, metric
        self.f = open(path, "wb")
        self.f.write(_header(self.d, 0, metric))

    def append(self, xb):
        xb = np.ascontiguousarray(xb, dtype="<f4")
        assert xb.ndim == 2 and xb.shape[1] == self.d
        _write_rows(self.f, xb)
        self.n += xb.shape[0]

    def close(self):
        self.f.seek(0)
        self.f.write(_header(self.d, self.n, self.metric))
        self.f.close()


HEADER_BYTES = len(_header(1, 0, METRIC_INNER_PRODUCT))


def write_header(f, d, n, metric=METRIC_INNER_PRODUCT):
    """(re)writes the header of a flat index whose rows are written positionally (the N-rank builder: every rank
    pwrite()s its rows at HEADER_BYTES + row * d * 4)"""
    os.pwrite(f.fileno() if hasattr(f, "fileno") else f, _header(d, n, metric), 0)


def read_index_flat(path, rows=None):
    """-> (xb float32[n,d], metric); raises ValueError for any other index type.  rows=(lo, hi): only that row range is
    read (a rank of a song-sharded job reads its shard and nothing else); the row count of the whole index is then
    returned as a third value."""
    if rows is not None:
        with open(path, "rb") as f:
            fourcc = f.read(4)
            if fourcc not in (b"IxFI", b"IxF2", b"IxFl"):
                raise ValueError("landmarkValue is not a flat index (fourcc %r): only exact Flat "
                                 "indexes are in scope" % fourcc)
            d, n, _, _, _, metric = struct.unpack("<iqqqBi", f.read(4 + 8 * 3 + 1 + 4))
            if metric > 1:
                f.read(4)
            (cnt,) = struct.unpack("<Q", f.read(8))
            if cnt != n * d:
                raise ValueError("corrupt flat index: %d floats for %d x %d" % (cnt, n, d))
            lo, hi = max(0, int(rows[0])), min(int(rows[1]), n)
            f.seek(lo * d * 4, 1)
            xb = np.fromfile(f, dtype="<f4", count=max(hi - lo, 0) * d)
        if xb.size != max(hi - lo, 0) * d:
            raise ValueError("truncated flat index")
        return xb.reshape(-1, d).astype(np.float32, copy=False), metric, n
    with open(path, "rb") as f:
        fourcc = f.read(4)
        if fourcc not in (b"IxFI", b"IxF2", b"IxFl"):
            raise ValueError("landmarkValue is not a flat index (fourcc %r): only exact Flat "
                             "indexes are in scope" % fourcc)
        d, n, _, _, _, metric = struct.unpack("<iqqqBi", f.read(4 + 8 * 3 + 1 + 4))
        if metric > 1:
            f.read(4)  # metric_arg
        (cnt,) = struct.unpack("<Q", f.read(8))
        xb = np.frombuffer(f.read(cnt * 4), dtype="<f4")
    if cnt != n * d:
        raise ValueError("corrupt flat index: %d floats for %d x %d" % (cnt, n, d))
    return xb.reshape(n, d).astype(np.float32), metric
