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
