"""Reader/writer for the faiss on-disk format of an IndexFlatIP/IndexFlatL2 ("landmarkValue",
reference builder.py:136, database.py:88).  faiss is un-vendored and absent here, so this
follows the published layout of faiss' `write_index` for IndexFlat (index_write.cpp:
fourcc "IxFI"/"IxF2", header {d:int32, ntotal:int64, dummy:int64 x2, is_trained:uint8,
metric_type:int32}, then a size_t-prefixed float vector).  PARITY UNPINNED: not checked
against a real faiss build (none is installable); the builder also always writes the raw
`embeddings` file, which the reference itself falls back to (database.py:96-97).
"""
import struct

import numpy as np

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


def write_index_flat(path, xb, metric=METRIC_INNER_PRODUCT):
    xb = np.ascontiguousarray(xb, dtype="<f4")
    n, d = xb.shape
    with open(path, "wb") as f:
        f.write(b"IxFI" if metric == METRIC_INNER_PRODUCT else b"IxF2")
        f.write(struct.pack("<iqqqBi", d, n, 1 << 20, 1 << 20, 1, metric))
        f.write(struct.pack("<Q", n * d))
        f.write(xb.tobytes())


def read_index_flat(path):
    """-> (xb float32[n,d], metric); raises ValueError for any other index type."""
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
