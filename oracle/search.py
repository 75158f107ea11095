"""Oracle a7: exact inner-product top-k (faiss IndexFlatIP.search semantics as used at
reference database.py:121: k largest q.x per query row, scores descending, labels int64,
-1 labels when fewer than k rows exist).  faiss is un-vendored; exact flat IP is
definitional, so this is a restatement of the definition, not of faiss code."""
import numpy as np


def flat_ip_topk(query, db, k):
    query = np.ascontiguousarray(query, dtype=np.float32)
    db = np.ascontiguousarray(db, dtype=np.float32).reshape(-1, query.shape[1])
    nq, n = query.shape[0], db.shape[0]
    D = np.full((nq, k), -np.finfo(np.float32).max, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if n == 0:
        return D, I
    s = query @ db.T
    kk = min(k, n)
    # stable: ties broken by smaller label
    order = np.argsort(-s, axis=1, kind="stable")[:, :kk]
    D[:, :kk] = np.take_along_axis(s, order, axis=1)
    I[:, :kk] = order
    return D, I


def flat_ip_topk_blas(query, db, k):
    """Same result as flat_ip_topk, via BLAS sgemm + argpartition (the faiss-cpu IndexFlatIP
    stand-in named in BASELINE.md §3 for CPU-baseline timing)."""
    query = np.ascontiguousarray(query, dtype=np.float32)
    nq, n = query.shape[0], db.shape[0]
    D = np.full((nq, k), -np.finfo(np.float32).max, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if n == 0:
        return D, I
    s = query @ db.T
    kk = min(k, n)
    if kk < n:
        part = np.argpartition(-s, kk - 1, axis=1)[:, :kk]
    else:
        part = np.tile(np.arange(n), (nq, 1))
    ps = np.take_along_axis(s, part, axis=1)
    o = np.lexsort((part, -ps), axis=1)
    D[:, :kk] = np.take_along_axis(ps, o, axis=1)
    I[:, :kk] = np.take_along_axis(part, o, axis=1)
    return D, I


def flat_ip_topk_f16(query, db, k):
    """fp16-storage semantics (faiss GpuMultipleClonerOptions.useFloat16, database.py:101-104): rows and
    queries rounded to IEEE fp16, exact products, wide accumulation -- the k largest
    s16 = sum_i fl16(q_i) * fl16(x_i).  Computed in float64 here (products of two fp16 are exact in fp32, so
    an fp32-accumulating device differs only by summation rounding, ~1e-7 of the norm product)."""
    q16 = np.asarray(query, np.float32).astype(np.float16).astype(np.float64)
    x16 = np.asarray(db, np.float32).reshape(-1, q16.shape[1]).astype(np.float16).astype(np.float64)
    nq, n = q16.shape[0], x16.shape[0]
    D = np.full((nq, k), -np.finfo(np.float32).max, dtype=np.float64)
    I = np.full((nq, k), -1, dtype=np.int64)
    if n == 0:
        return D, I
    s = q16 @ x16.T
    kk = min(k, n)
    order = np.argsort(-s, axis=1, kind="stable")[:, :kk]
    D[:, :kk] = np.take_along_axis(s, order, axis=1)
    I[:, :kk] = order
    return D, I
