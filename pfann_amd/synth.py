"""Seeded synthetic inputs for tests and bench: weights, 8 kHz songs, noisy queries.

No datasets or trained weights exist in this environment, so every
test/bench input is regenerated from integer seeds with a counter-based PRNG written
here in pure integer arithmetic (splitmix64 finaliser).  The same numbers therefore come
out on the build container and on the GPU box without shipping 68 MB of weights.

Specs followed (behaviour only, no code shared with the reference):
  * weights: state_dict names/shapes of FpNetwork (reference model.py:14-34,75-95,108-120)
  * query crops: genquery.py:42-53 (seed 9000+index, random crop), noise mix at a given
    SNR by the formula of datautil/noise.py:96-109, peak normalise (genquery.py:94),
    16-bit quantise.
"""
import math
import wave

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def uniform01(seed: int, name: str, n: int) -> np.ndarray:
    """n float32 values in [0,1), a pure function of (seed, name, index)."""
    base = np.uint64((_fnv1a(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base) & _M64
    bits = _splitmix(idx) >> np.uint64(40)  # top 24 bits
    return (bits.astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


def normal(seed: int, name: str, n: int) -> np.ndarray:
    """n float32 standard normals (Box-Muller on uniform01)."""
    m = (n + 1) // 2
    u1 = uniform01(seed, name + "/u1", m).astype(np.float64)
    u2 = uniform01(seed, name + "/u2", m).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    z = np.concatenate([r * np.cos(2 * math.pi * u2), r * np.sin(2 * math.pi * u2)])[:n]
    return z.astype(np.float32)


# ----------------------------------------------------------------------------- weights
def model_dims(params):
    """Derive (d, h, u, F, T) the way builder.py:46-51 does."""
    m = params["model"]
    segn = int(params["segment_size"] * params["sample_rate"])
    T = (segn + params["stft_hop"] - 1) // params["stft_hop"]
    return m["d"], m["h"], m["u"], params["n_mels"], T


def layer_plan(params):
    """Per separable block: dict(ci, co, F, T, s_t, s_f, T1, F2, pad1, pad2, depthwise).

    Shapes follow reference model.py:15-30 (padding = (in-1)//s*s + k - in, split
    left=pad//2, right=pad-pad//2) and model.py:79-93 (channel ladder, stride table).
    """
    d, h, u, F, T = model_dims(params)
    m = params["model"]
    ch = [1, d, d, 2 * d, 2 * d, 4 * d, 4 * d, h, h]
    strides = m.get("strides")
    plan = []
    for i in range(8):
        s_t, s_f = 2, 2
        if strides is not None:
            s_t, s_f = strides[i][0][1], strides[i][1][0]
        k = 3
        p1 = (T - 1) // s_t * s_t + k - T
        p2 = (F - 1) // s_f * s_f + k - F
        T1 = (T - 1) // s_t + 1
        F2 = (F - 1) // s_f + 1
        plan.append(dict(ci=ch[i], co=ch[i + 1], F=F, T=T, s_t=s_t, s_f=s_f, T1=T1, F2=F2,
                         pad1=(p1 // 2, p1 - p1 // 2), pad2=(p2 // 2, p2 - p2 // 2),
                         depthwise=not m.get("fuller", False)))
        F, T = F2, T1
    assert F == 1 and T == 1, "output must be 1x1"
    return plan


def state_dict_spec(params):
    """Ordered [(name, shape, kind, fan_in)] of the 68 tensors of FpNetwork.state_dict()."""
    d, h, u, _, _ = model_dims(params)
    spec = []
    for i, L in enumerate(layer_plan(params)):
        p = "f.convs.%d." % i
        spec.append((p + "conv1.weight", (L["co"], L["ci"], 1, 3), "w", L["ci"] * 3))
        spec.append((p + "conv1.bias", (L["co"],), "b", L["ci"] * 3))
        spec.append((p + "ln1.weight", (L["co"], L["F"], L["T1"]), "lnw", 0))
        spec.append((p + "ln1.bias", (L["co"], L["F"], L["T1"]), "lnb", 0))
        ci2 = 1 if L["depthwise"] else L["co"]
        spec.append((p + "conv2.weight", (L["co"], ci2, 3, 1), "w", ci2 * 3))
        spec.append((p + "conv2.bias", (L["co"],), "b", ci2 * 3))
        spec.append((p + "ln2.weight", (L["co"], L["F2"], L["T1"]), "lnw", 0))
        spec.append((p + "ln2.bias", (L["co"], L["F2"], L["T1"]), "lnb", 0))
    v = h // d
    spec.append(("g.linear1.weight", (d * u, v, 1), "w", v))
    spec.append(("g.linear1.bias", (d * u,), "b", v))
    spec.append(("g.linear2.weight", (d, u, 1), "w", u))
    spec.append(("g.linear2.bias", (d,), "b", u))
    return spec


def make_state_dict(params, seed=123):
    """Seeded numpy state_dict: convs uniform(+-1/sqrt(fan_in)), LN weight 1+0.1u, bias 0.1u."""
    sd = {}
    for name, shape, kind, fan_in in state_dict_spec(params):
        n = int(np.prod(shape))
        u01 = uniform01(seed, name, n)
        sym = (u01 * np.float32(2.0) - np.float32(1.0)).astype(np.float32)
        if kind in ("w", "b"):
            val = sym * np.float32(1.0 / math.sqrt(fan_in))
        elif kind == "lnw":
            val = np.float32(1.0) + np.float32(0.1) * sym
        else:
            val = np.float32(0.1) * sym
        sd[name] = val.astype(np.float32).reshape(shape)
    return sd


# ------------------------------------------------------------------------------- audio
def make_song(song_id: int, seconds: float = 30.0, sr: int = 8000) -> np.ndarray:
    """Seeded mono int16 song: 6-12 partials with slow AM/FM in 300-3800 Hz + -20 dB noise."""
    seed = 1000 + song_id
    n = int(seconds * sr)
    t = np.arange(n, dtype=np.float64) / sr
    u = uniform01(seed, "song/params", 64).astype(np.float64)
    n_part = 6 + int(u[0] * 7)
    x = np.zeros(n, dtype=np.float64)
    # piecewise "notes": every partial re-tunes a few times so that segments differ in time
    for p in range(n_part):
        q = uniform01(seed, "song/p%d" % p, 64).astype(np.float64)
        n_notes = 8 + int(q[0] * 8)
        edges = np.sort(np.concatenate([[0.0], q[1:n_notes] * seconds, [seconds]]))
        amp = 0.3 + 0.7 * q[20]
        for j in range(n_notes):
            a, b = int(edges[j] * sr), int(edges[j + 1] * sr)
            if b <= a:
                continue
            f0 = 300.0 + 3500.0 * uniform01(seed, "song/p%d/n%d" % (p, j), 4).astype(np.float64)
            tt = t[a:b] - t[a]
            fm = 1.0 + 0.004 * np.sin(2 * math.pi * (0.5 + 4 * f0[1] / 3800.0) * tt)
            am = 0.6 + 0.4 * np.sin(2 * math.pi * (0.3 + 2.0 * f0[2] / 3800.0) * tt + 6.28 * f0[3] / 3800.0)
            ph = 2 * math.pi * np.cumsum(f0[0] * fm) / sr
            x[a:b] += amp * am * np.sin(ph + 6.28 * q[30 + (j % 30)])
    x /= max(np.sqrt(np.mean(x * x)), 1e-9)
    nz = normal(seed, "song/noise", n).astype(np.float64)
    nz = np.cumsum(nz) * 0.02 + nz  # a little low-frequency tilt
    nz /= max(np.sqrt(np.mean(nz * nz)), 1e-9)
    x = x + 0.1 * nz
    x /= np.max(np.abs(x)) + 1e-12
    return np.round(x * 32000.0).astype(np.int16)


def make_query(song: np.ndarray, index: int, seconds: float = 10.0, snr_db: float = 0.0,
               sr: int = 8000):
    """Crop + white/pink-ish noise at snr_db + peak normalise + 16-bit quantise.

    Returns (int16 query, time_offset_seconds).
    """
    seed = 9000 + index
    sel = int(seconds * sr)
    u = uniform01(seed, "query/off", 2).astype(np.float64)
    if song.shape[0] >= sel:
        hi = max(song.shape[0] - sel, 1)
        off = int(u[0] * hi)
        x = song[off:off + sel].astype(np.float64) / 32768.0
    else:
        off = 0
        x = np.pad(song.astype(np.float64) / 32768.0, (0, sel - song.shape[0]))
    nz = normal(seed, "query/noise", sel).astype(np.float64)
    eps = 1e-12
    vol_x = math.sqrt(max(float(np.mean(x * x)), eps))
    vol_n = math.sqrt(max(float(np.mean(nz * nz)), eps))
    ratio = vol_x / vol_n * 10.0 ** (-snr_db / 20.0)
    y = x + ratio * nz
    y /= np.max(np.abs(y)) + 1e-12
    return np.round(y * 32767.0).astype(np.int16), off / sr


def write_wav(path: str, pcm: np.ndarray, sr: int = 8000):
    """pcm int16 [n] mono or [n, ch]."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    with wave.open(path, "wb") as w:
        w.setnchannels(1 if pcm.ndim == 1 else pcm.shape[1])
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())


def unit_rows(seed: int, name: str, n: int, d: int) -> np.ndarray:
    """Seeded unit-norm Gaussian rows f32[n,d] (scan-only benches)."""
    x = normal(seed, name, n * d).reshape(n, d).astype(np.float64)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    return x.astype(np.float32)


# ------------------------------------------------- large-scale synthetic audio (torch, any device)
# BASELINE configs 2-5 need 10 k - 100 k songs and thousands of queries: the numpy generators above
# (~40 ms per song) would take an hour, so the same kind of signal is generated with torch ops that run on the
# GPU box's device (100 k songs in seconds) -- and on the CPU here, for the calibration constants below.
# Everything is a pure function of integer ids (splitmix64 in wrapping int64 arithmetic); fp64 phases.
_SM1, _SM2, _SM3 = -7046029254386353131, -4658895280553007687, -7723592293110705685   # splitmix64 constants as int64


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _hash64(x):
    x = x + _SM1
    x = (x ^ _lsr(x, 30)) * _SM2
    x = (x ^ _lsr(x, 27)) * _SM3
    return x ^ _lsr(x, 31)


def _u01_t(h, which=0):
    """24-bit uniform in [0,1) from bits [40,64) (which=0) or [8,32) (which=1) of a hash, float64."""
    import torch
    bits = _lsr(h, 40) if which == 0 else (_lsr(h, 8) & 0xFFFFFF)
    return bits.to(torch.float64) * (1.0 / (1 << 24))


def make_songs_torch(song_ids, seconds=30.0, sr=8000, device="cpu"):
    """-> int16 [S, n] mono songs: 8 partials of piecewise notes (0.9 - 3.4 s) with FM and AM in 300-3800 Hz
    plus -20 dB white noise, peak-normalised; song s is a pure function of its id."""
    import torch
    ids = torch.as_tensor(song_ids, dtype=torch.int64, device=device).reshape(-1, 1)        # [S,1]
    n = int(seconds * sr)
    t = torch.arange(n, dtype=torch.int64, device=device).reshape(1, -1)                     # [1,n]
    x = torch.zeros((ids.shape[0], n), dtype=torch.float32, device=device)
    two_pi = 2.0 * math.pi
    for p in range(8):
        hp = _hash64(ids * 1000003 + p * 7919 + 17)                                          # per (song, partial)
        L = int((0.9 + 0.36 * p) * sr)
        off = (_u01_t(hp) * L).to(torch.int64)                                               # [S,1]
        amp = (0.3 + 0.7 * _u01_t(hp, 1)).to(torch.float32)
        tt = t + off
        note, tau = tt // L, (tt % L).to(torch.float64) / sr                                 # [S,n]
        h1 = _hash64(ids * 2000003 + p * 104729 + note * 15485863 + 29)
        h2 = _hash64(h1 + 0x5851F42D)
        f0 = 300.0 + 3500.0 * _u01_t(h1)
        fm_rate, fm_depth = 0.5 + 4.0 * _u01_t(h1, 1), 2.0 * _u01_t(h2)
        ph0 = _u01_t(h2, 1)
        h3 = _hash64(h2 + 0x2545F491)
        am_rate, am_ph = 0.3 + 2.0 * _u01_t(h3), _u01_t(h3, 1)
        turns = f0 * tau + ph0
        phase = (two_pi * (turns - torch.floor(turns))).to(torch.float32) + \
            (fm_depth.to(torch.float32) * torch.sin((two_pi * fm_rate * tau).to(torch.float32)))
        am = 0.6 + 0.4 * torch.sin((two_pi * (am_rate * tau + am_ph)).to(torch.float32))
        x += amp * am * torch.sin(phase)
        del h1, h2, h3, f0, fm_rate, fm_depth, ph0, am_rate, am_ph, turns, phase, am, note, tau, tt
    x /= x.square().mean(dim=1, keepdim=True).sqrt().clamp_min(1e-9)
    hn = _hash64(ids * 4294967311 + t)
    hn2 = _hash64(hn + 0x632BE5AB)
    nz = ((_u01_t(hn) + _u01_t(hn, 1) + _u01_t(hn2) + _u01_t(hn2, 1) - 2.0) * math.sqrt(3.0)).to(torch.float32)
    x += 0.1 * nz
    x /= x.abs().amax(dim=1, keepdim=True) + 1e-12
    return torch.round(x * 32000.0).to(torch.int16)


def make_queries_torch(song_pcm, query_ids, seconds=10.0, snr_db=0.0, sr=8000):
    """song_pcm int16 [Q, n] (row j = the source song of query j), query_ids [Q] -> (int16 [Q, sel] queries,
    float64 [Q] offsets in seconds): random crop (genquery.py:42-53), white noise at snr_db by the formula of
    datautil/noise.py:96-109, peak normalise (genquery.py:94), 16-bit quantise.  Seeded by 9000 + query id."""
    import torch
    dev = song_pcm.device
    Q, n = song_pcm.shape
    sel = int(seconds * sr)
    assert n >= sel
    qid = torch.as_tensor(query_ids, dtype=torch.int64, device=dev).reshape(-1, 1) + 9000
    off = (_u01_t(_hash64(qid * 6700417 + 5)) * max(n - sel, 1)).to(torch.int64)            # [Q,1]
    t = torch.arange(sel, dtype=torch.int64, device=dev).reshape(1, -1)
    x = torch.gather(song_pcm, 1, off + t).to(torch.float32) / 32768.0
    h = _hash64(qid * 4294967311 + t + (1 << 40))
    u1, u2 = _u01_t(h).clamp_min(2.0 ** -25), _u01_t(h, 1)
    nz = (torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)).to(torch.float32)
    vol_x = x.square().mean(dim=1, keepdim=True).clamp_min(1e-12).sqrt()
    vol_n = nz.square().mean(dim=1, keepdim=True).clamp_min(1e-12).sqrt()
    y = x + (vol_x / vol_n * (10.0 ** (-snr_db / 20.0))) * nz
    y /= y.abs().amax(dim=1, keepdim=True) + 1e-12
    return torch.round(y * 32767.0).to(torch.int16), off.reshape(-1).to(torch.float64) / sr


# An UNTRAINED FpNetwork maps every input to nearly the same point (pairwise cosine 0.986 +- 0.003 between
# segments of different songs with the seeded weights above: ReLU + LayerNorm stacks contract angles), which makes a
# large "real" database degenerate: every score within 0.02 of every other.  No trained weights exist here, so for
# the large-scale workloads the seeded state_dict is CALIBRATED the way data-dependent initialisers do it: the head's
# output bias g.linear2.bias is shifted by minus the mean un-normalised output over a calibration set (24 synthetic
# songs, computed once by tools/make_synth_calib.py with the CPU oracle and stored in synth_calib.json), so the
# embeddings spread over the sphere (cosine between different songs 0.00 +- 0.15).  Same architecture, same
# arithmetic; the constants are data shipped with the package, so weights are identical on every box.
def calib_key(params, seed):
    m = params["model"]
    return "d%d_h%d_u%d_fuller%d_%s_seed%d" % (m["d"], m["h"], m["u"], 1 if m.get("fuller", False) else 0,
                                                m.get("conv_activation", "ReLU"), seed)


def make_state_dict_calibrated(params, seed=123):
    import json
    import os
    sd = make_state_dict(params, seed)
    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_calib.json")))
    key = calib_key(params, seed)
    if key not in table:
        raise KeyError("no calibration constants for %s: run tools/make_synth_calib.py" % key)
    sd["g.linear2.bias"] = (sd["g.linear2.bias"].astype(np.float64) - np.asarray(table[key], np.float64)).astype(np.float32)
    return sd
