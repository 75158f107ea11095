"""ctypes access to the C restatement (oracle/seqscore_c.c).  TEST INFRASTRUCTURE."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "seqscore_c.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", src, "-lm", "-o", _SO])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        f32p, i64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        _lib.oracle_seq_score.argtypes = [f32p, ctypes.c_int, i64p, ctypes.c_int, f32p, ctypes.c_int,
                                          i64p, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_float]
        _lib.oracle_seq_score.restype = ctypes.c_int
        _lib.oracle_flat_ip_topk.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int,
                                             ctypes.c_int, f32p, i64p]
        _lib.oracle_flat_ip_topk.restype = None
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def seq_score(db, song_pos, query, labels, frame_shift_mul=1, score_alpha=0.0):
    """-> (best_song, song_scores float32[n_songs,2] with offsets in frames)."""
    db = np.ascontiguousarray(db, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    labels = np.ascontiguousarray(labels, np.int64)
    song_pos = np.ascontiguousarray(song_pos, np.int64)
    n_songs = song_pos.shape[0] - 1
    d = query.shape[1]
    ss = np.zeros((n_songs, 2), np.float32)
    best = lib().oracle_seq_score(_p(db, ctypes.c_float), d, _p(song_pos, ctypes.c_int64), n_songs,
                                  _p(query, ctypes.c_float), query.shape[0],
                                  _p(labels, ctypes.c_int64), labels.shape[1],
                                  _p(ss, ctypes.c_float), frame_shift_mul, score_alpha)
    return best, ss


def flat_ip_topk(query, db, k):
    db = np.ascontiguousarray(db, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    nq, d = query.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    lib().oracle_flat_ip_topk(_p(db, ctypes.c_float), db.shape[0], d, _p(query, ctypes.c_float), nq, k,
                              _p(D, ctypes.c_float), _p(I, ctypes.c_int64))
    return D, I
