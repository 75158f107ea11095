"""Oracle a2: float32[B, L] -> float32[B, n_mels, n_frames] log-mel
(reference datautil/melspec.py:33-50 around torchaudio.transforms.MelSpectrogram).

PARITY UNPINNED: torchaudio is not installed here and its version is not pinned by the
reference (readme.md:14), so the transform is restated from its documented semantics:
torch.stft(n_fft, hop, window=hann_window(n_fft) periodic, center=True, pad_mode,
normalized=False, onesided=True) -> |.|**power -> fb^T @ spec, with fb[n_freqs, n_mels]
the triangular bank over bins at linspace(0, sr//2, n_freqs) Hz with n_mels+2 edges equally spaced in
mel between f_min and f_max (written here filter by filter in float64, see mel_filterbank).
Everything outside the transform follows melspec.py:33-50 line by line.
Second opinion (not a pin): tests/test_oracle.py compares this restatement with
transformers.audio_utils (an unrelated numpy/fp64 implementation of the same semantics).
"""
import math

import numpy as np
import torch


def _hz_to_mel(f, scale):
    """HTK: 2595 log10(1 + f/700).  Slaney: linear (200/3 Hz per mel) below 1 kHz, logarithmic above (27 steps per
    factor 6.4).  float64."""
    if scale == "htk":
        return 2595.0 * math.log10(1.0 + f / 700.0)
    if f >= 1000.0:
        return 15.0 + 27.0 * math.log(f / 1000.0) / math.log(6.4)
    return 3.0 * f / 200.0


def _mel_to_hz(m, scale):
    if scale == "htk":
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    if m >= 15.0:
        return 1000.0 * 6.4 ** ((m - 15.0) / 27.0)
    return 200.0 * m / 3.0


def mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max, naf_mode=False):
    """fb float32[n_fft//2+1, n_mels]: htk scale / no normalisation by default, slaney scale / slaney (area) normalisation
    in naf_mode (melspec.py:27-30).

    Written from the PUBLISHED DEFINITION of the triangular bank, filter by filter, in float64 -- deliberately not the
    vectorised fp32 torch op sequence the product's host code (pfann_amd/engine.py:mel_filterbank, which imitates how
    torchaudio builds its bank) uses, so that the two are independent statements: n_mels + 2 band edges equally spaced
    on the mel scale between f_min and f_max; filter m rises linearly from 0 at edge m to 1 at edge m+1 and falls back to
    0 at edge m+2; bin k sits at k * (sample_rate // 2) / (n_freqs - 1) Hz.  tests/test_host.py measures the gap between
    the two (fp32 vs fp64 construction) and uses it as the tolerance."""
    scale = "slaney" if naf_mode else "htk"
    n_freqs = n_fft // 2 + 1
    lo_mel, hi_mel = _hz_to_mel(float(f_min), scale), _hz_to_mel(float(f_max), scale)
    edges = [_mel_to_hz(lo_mel + (hi_mel - lo_mel) * j / (n_mels + 1), scale) for j in range(n_mels + 2)]
    fb = np.zeros((n_freqs, n_mels), dtype=np.float64)
    step = (sample_rate // 2) / (n_freqs - 1)
    for m in range(n_mels):
        left, centre, right = edges[m], edges[m + 1], edges[m + 2]
        for k in range(int(math.floor(left / step)), min(int(math.ceil(right / step)) + 1, n_freqs)):
            f = k * step
            if left < f < right:
                fb[k, m] = min((f - left) / (centre - left), (right - f) / (right - centre))
        if naf_mode:
            fb[:, m] *= 2.0 / (right - left)
    return torch.from_numpy(fb.astype(np.float32))


def mel_filterbank_torchaudio(sample_rate, n_fft, n_mels, f_min, f_max, naf_mode=False):
    """The SAME bank the way torchaudio itself builds it: torchaudio.functional.melscale_fbanks +
    _create_triangular_filterbank (module torchaudio, version unpinned by the reference; restated from its published
    source), i.e. in float32 torch ops: bin frequencies torch.linspace(0, sr // 2, n_freqs); n_mels + 2 points
    torch.linspace(mel(f_min), mel(f_max)) mapped back to Hz in float32; slopes = f_pts[None, :] - all_freqs[:, None];
    fb = max(0, min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:])); slaney: columns scaled by
    2 / (f_pts[2:] - f_pts[:-2]).

    Why both exist (round 5, tools/embedding_error_budget.py): around 4 kHz a float32 frequency carries 2.4e-4 Hz of
    rounding against filters ~14 Hz wide, so the float32 construction and mel_filterbank()'s float64 one differ by up to
    3.8e-5 of a unit-peak weight -- and on clean tonal material, whose quiet mel bins sit at the log's 1e-8 floor, that
    alone moves a fingerprint by up to 1.9e-4 (118,000 database rows), more than every fp32 rounding of either side
    together (GPU 6e-6 .. 2.6e-5, torch-CPU 3e-6 .. 1.8e-5 from float64).  mel_filterbank() stays the default of
    melspec() -- the independent statement every fixture-level test was written against; the population-scale tools
    evaluate with THIS one first, because it is what a reference installation computes, and report the other beside it."""
    scale = "slaney" if naf_mode else "htk"
    n_freqs = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(float(f_min), scale), _hz_to_mel(float(f_max), scale), n_mels + 2)
    if scale == "htk":
        f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    else:
        logstep = math.log(6.4) / 27.0
        f_pts = torch.where(m_pts >= 15.0, 1000.0 * torch.exp(logstep * (m_pts - 15.0)), (200.0 / 3) * m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down, up = (-1.0 * slopes[:, :-2]) / f_diff[:-1], slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    if naf_mode:
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb.to(torch.float32)


def melspec(x, params, bank=None):
    """x float32 [B, L] (numpy or torch) -> numpy float32 [B, n_mels, 1 + L//hop].  bank: a filter bank
    float32 [n_freqs, n_mels] to use instead of mel_filterbank() -- the device-kernel parity tests hand in the bank the
    kernel was given (so that they check the STFT / mel / log arithmetic to fp32 rounding); how far that bank is from
    this module's float64 statement is bounded separately (tests/test_host.py)."""
    x = torch.as_tensor(np.asarray(x, dtype=np.float32))
    naf = params.get("naf_mode", False)
    mel_log = params.get("mel_log", "log")
    spec_norm = params.get("spec_norm", "l2")
    n_fft, hop = params["stft_n"], params["stft_hop"]
    p = float("inf") if spec_norm == "max" else 2
    x = torch.nn.functional.normalize(x, p=p, dim=-1)               # melspec.py:35-36
    spec = torch.stft(x, n_fft, hop_length=hop, win_length=n_fft,
                      window=torch.hann_window(n_fft), center=True,
                      pad_mode="constant" if naf else "reflect",
                      normalized=False, onesided=True, return_complex=True)
    spec = spec.abs().pow(1.0 if naf else 2.0)                      # melspec.py:27
    fb = mel_filterbank(params["sample_rate"], n_fft, params["n_mels"],
                        params["f_min"], params["f_max"], naf) if bank is None else torch.as_tensor(np.asarray(bank, np.float32))
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)
    mel = mel + (0.06 if naf else 1e-8)                             # melspec.py:38-41
    if mel_log == "log10":
        mel = torch.log10(mel)
    elif mel_log == "log":
        mel = torch.log(mel)
    if spec_norm == "max":
        mel = mel - torch.amax(mel, dim=(-2, -1), keepdim=True)     # melspec.py:48-49
    return mel.numpy()


def melspec_f64(x, params, bank=None):
    """Independent float64 check of the default mode (explicit frame gather + numpy rfft);
    used only to size the fp32 error of melspec() and of the HIP kernel."""
    x = np.asarray(x, dtype=np.float64)
    n_fft, hop = params["stft_n"], params["stft_hop"]
    x = x / np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), 1e-12)
    L = x.shape[-1]
    n_frames = 1 + L // hop
    n = np.arange(n_fft)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
    idx = np.arange(n_frames)[:, None] * hop - n_fft // 2 + n[None, :]
    idx = np.where(idx < 0, -idx, idx)
    idx = np.where(idx > L - 1, 2 * (L - 1) - idx, idx)
    frames = x[..., idx] * win
    power = np.abs(np.fft.rfft(frames, axis=-1)) ** 2              # [B, frames, bins]
    fb = (mel_filterbank(params["sample_rate"], n_fft, params["n_mels"], params["f_min"], params["f_max"], False)
          if bank is None else torch.as_tensor(np.asarray(bank, np.float32))).double().numpy()
    mel = np.einsum("...tk,km->...mt", power, fb)
    return np.log(mel + 1e-8)


def melspec_f64_torch(x, params, bank=None, device=None):
    """melspec_f64 with torch float64 ops on `device` (gather, window, torch.fft.rfft, matmul, log) -> torch float64
    tensor [B, n_mels, frames] on that device: the front half of the float64 yardstick when it is evaluated on the GPU
    (oracle/encoder.py: encode(device=...)); tools/embedding_error_budget.py checks it against melspec_f64."""
    x = torch.as_tensor(np.asarray(x, dtype=np.float64), device=device)
    n_fft, hop = params["stft_n"], params["stft_hop"]
    x = x / torch.clamp(torch.linalg.norm(x, dim=-1, keepdim=True), min=1e-12)
    L = x.shape[-1]
    n_frames = 1 + L // hop
    n = torch.arange(n_fft, device=device)
    win = 0.5 - 0.5 * torch.cos(2 * math.pi * n.double() / n_fft)
    idx = torch.arange(n_frames, device=device)[:, None] * hop - n_fft // 2 + n[None, :]
    idx = torch.where(idx < 0, -idx, idx)
    idx = torch.where(idx > L - 1, 2 * (L - 1) - idx, idx)
    frames = x[..., idx] * win
    power = torch.fft.rfft(frames, dim=-1).abs() ** 2
    fb = (mel_filterbank(params["sample_rate"], n_fft, params["n_mels"], params["f_min"], params["f_max"], False)
          if bank is None else torch.as_tensor(np.asarray(bank, np.float32))).double().to(device)
    mel = torch.matmul(power, fb).transpose(-1, -2)
    return torch.log(mel + 1e-8)
