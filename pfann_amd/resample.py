"""Host side of the non-native-sample-rate path (reference datautil/musicdata.py:28-65): the polyphase filter table of
`julius.ResampleFrac(file_sr, sample_rate)` and the reference's minute-wise piece plan.  The filtering itself runs on the
GPU (`pfann_resample_to_mono`, csrc/mel.hip).

`julius` is an un-vendored dependency of the reference (version unpinned) and is not in this image: the table is built from
its published algorithm (zeros = 24, rolloff = 0.945, Hann-windowed sinc per output phase, every phase normalised to unit
sum) with the same fp32 torch ops, so this path's parity with the reference is unpinned."""
import math

import numpy as np
import torch

ZEROS, ROLLOFF = 24, 0.945
_cache = {}


def reduced_rates(file_sr, sr):
    g = math.gcd(int(file_sr), int(sr))
    return int(file_sr) // g, int(sr) // g


def filter_table(file_sr, sr):
    """-> (float32 numpy [new, 2*width + old], old, new, width), cached per rate pair."""
    key = reduced_rates(file_sr, sr)
    if key not in _cache:
        old, new = key
        base = min(new, old) * ROLLOFF
        width = math.ceil(ZEROS * old / base)
        idx = torch.arange(-width, width + old).float()
        rows = []
        for i in range(new):
            t = (-i / new + idx / old) * base
            t = t.clamp_(-ZEROS, ZEROS)
            t *= math.pi
            k = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * torch.cos(t / ZEROS / 2) ** 2
            k.div_(k.sum())
            rows.append(k)
        _cache[key] = (np.ascontiguousarray(torch.stack(rows).numpy()), old, new, width)
    return _cache[key]


def piece_plan(n_in, file_sr, sr):
    """int64 [n_pieces, 5] = (in_start, in_len, out_skip, out_keep, out_off): 60 s pieces starting every 59 s, the first half
    second of every piece but the first and the last half second of every piece but the tail dropped (musicdata.py:33-65;
    the stream arrives in 1024-frame blocks, so a piece is cut whenever a whole minute is left).  -> (plan, n_out)."""
    old, new = reduced_rates(file_sr, sr)
    minute, second = file_sr * 60, file_sr
    new_min, new_sec = sr * 60, sr
    rows, start, strip, off = [], 0, 0, 0
    while n_in - start >= minute:
        keep = new_min - new_sec // 2 - strip
        rows.append((start, minute, strip, keep, off))
        off += keep
        start += minute - second
        strip = new_sec // 2
    tail = n_in - start
    keep = max(int(new * tail / old) - strip, 0)
    rows.append((start, tail, strip, keep, off))
    return np.asarray(rows, dtype=np.int64), off + keep
