"""Oracle a1, non-native sample rates: the reference resamples every file with `julius.ResampleFrac(file_sr, 8000)`
minute by minute (reference datautil/musicdata.py:28-65).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: `julius` is an un-vendored dependency of the
reference (module `julius`, version unpinned: readme.md lists it without one; call sites musicdata.py:29,56,65) and is
absent from this image, and the reference holds no fixture for this path.  What follows restates the published algorithm
of `julius.resample.ResampleFrac` (defaults zeros = 24, rolloff = 0.945); the filter table is evaluated in float64 numpy
(see kernels()), the filtering itself with torch's conv1d in fp32:

    gcd-reduce (old, new);  sr = min(old, new) * rolloff;  width = ceil(zeros * old / sr)
    idx = arange(-width, width + old) (float32);  for every output phase i in [0, new):
        t = clamp((-i / new + idx / old) * sr, -zeros, zeros) * pi
        kernel_i = sinc(t) * cos(t / zeros / 2)^2, divided by its sum            (a constant signal is preserved)
    y = conv1d(replicate_pad(x, width, width + old), kernels, stride = old), phases interleaved,
        cut to int(new * len / old) samples

and the reference's chunking around it (musicdata.py:33-65): 60 s pieces that start every 59 s, each resampled on its own
(so its edges see replicate padding, not the neighbouring audio), of which the first half second (except in the first
piece) and the last half second (except in the tail piece) are thrown away."""
import math

import numpy as np
import torch
import torch.nn.functional as F

ZEROS, ROLLOFF = 24, 0.945


def reduced(old_sr, new_sr):
    g = math.gcd(int(old_sr), int(new_sr))
    return int(old_sr) // g, int(new_sr) // g


def kernels(old_sr, new_sr):
    """-> (float32 [new, 2*width + old] tensor, width) for gcd-reduced rates.

    Evaluated in float64 numpy straight from the formula in the header (np.sinc, one phase at a time) and rounded to
    float32 once at the end -- deliberately NOT the fp32 torch op sequence of the product's host code
    (pfann_amd/resample.py:filter_table, which imitates julius op by op), so that the two are independent statements;
    tests/test_host.py measures the gap and uses it as the tolerance."""
    old, new = reduced(old_sr, new_sr)
    sr = min(new, old) * ROLLOFF
    width = math.ceil(ZEROS * old / sr)
    idx = np.arange(-width, width + old, dtype=np.float64)
    ks = np.empty((new, idx.shape[0]), dtype=np.float64)
    for i in range(new):
        t = np.clip((idx / old - i / new) * sr, -ZEROS, ZEROS)              # in units of the low-pass's zero crossings
        k = np.sinc(t) * np.cos(0.5 * math.pi * t / ZEROS) ** 2            # sin(pi t)/(pi t) under a Hann lobe of 2*ZEROS crossings
        ks[i] = k / k.sum()
    return torch.from_numpy(ks.astype(np.float32)), width


def resample_frac(x, old_sr, new_sr, table=None):
    """x float32 [ch, n] -> [ch, int(new * n / old)] (one call of the reference's resampler).  table: a filter table
    to use instead of kernels() (float32 [new, 2*width + old]) -- the device-kernel parity test hands in the table the
    kernel was given, so that it checks the filtering arithmetic to fp32 rounding; how far that table is from the float64
    definition is measured separately (tests/test_host.py)."""
    x = torch.as_tensor(np.asarray(x, np.float32))
    old, new = reduced(old_sr, new_sr)
    if old == new:
        return x.numpy()
    n = x.shape[-1]
    if n == 0:
        return np.zeros((x.shape[0], 0), np.float32)
    k, width = kernels(old, new)
    if table is not None:
        assert tuple(table.shape) == tuple(k.shape)
        k = torch.as_tensor(np.asarray(table, np.float32))
    xp = F.pad(x[:, None], (width, width + old), mode="replicate")
    ys = F.conv1d(xp, k.view(new, 1, -1), stride=old)              # [ch, new, frames]
    y = ys.transpose(1, 2).reshape(x.shape[0], -1)
    return y[:, : int(new * n / old)].numpy()


def chunk_plan(n_in, file_sr, sr):
    """The reference's minute-wise pieces as (in_start, in_len, out_skip, out_keep) in samples: piece k starts 59 s after
    piece k-1; a piece is cut as soon as a whole minute is available (the stream arrives in 1024-frame blocks,
    audio.py:142-149), the rest is the tail piece."""
    minute, second = file_sr * 60, file_sr
    new_min, new_sec = sr * 60, sr
    plan, start, strip = [], 0, 0
    # the reference tests `n >= minute` after every 1024-frame block: with blocks shorter than a minute that is
    # "while a whole minute is left from the piece start"
    while n_in - start >= minute:
        plan.append((start, minute, strip, new_min - new_sec // 2 - strip))
        start += minute - second
        strip = new_sec // 2
    tail = n_in - start
    out_len = int(reduced(file_sr, sr)[1] * tail / reduced(file_sr, sr)[0]) if file_sr != sr else tail
    plan.append((start, tail, strip, max(out_len - strip, 0)))
    return plan


def resample_chunked(x, file_sr, sr, table=None):
    """x float32 [ch, n] at file_sr -> float32 [ch, n'] at sr, as musicdata.py:33-65 assembles it."""
    x = np.asarray(x, np.float32)
    out = []
    for start, n, skip, keep in chunk_plan(x.shape[1], file_sr, sr):
        y = resample_frac(x[:, start:start + n], file_sr, sr, table)
        out.append(y[:, skip:skip + keep])
    return np.concatenate(out, axis=1)
