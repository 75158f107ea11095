"""Oracle a2: float32[B, L] -> float32[B, n_mels, n_frames] log-mel
(reference datautil/melspec.py:33-50 around torchaudio.transforms.MelSpectrogram).

PARITY UNPINNED: torchaudio is not installed here and its version is not pinned by the
reference (readme.md:14), so the transform is restated from its documented semantics:
torch.stft(n_fft, hop, window=hann_window(n_fft) periodic, center=True, pad_mode,
normalized=False, onesided=True) -> |.|**power -> fb^T @ spec, with fb[n_freqs, n_mels]
the triangular bank built in fp32 from linspace(0, sr//2, n_freqs) and
linspace(mel(f_min), mel(f_max), n_mels+2) as max(0, min(down, up)).
Everything outside the transform follows melspec.py:33-50 line by line.
Second opinion (not a pin): tests/test_oracle.py compares this restatement with
transformers.audio_utils (an unrelated numpy/fp64 implementation of the same semantics).
"""
import math

import numpy as np
import torch


def _hz_to_mel(f, scale):
    if scale == "htk":
        return 2595.0 * math.log10(1.0 + f / 700.0)
    f_sp = 200.0 / 3
    if f >= 1000.0:
        return 15.0 + math.log(f / 1000.0) / (math.log(6.4) / 27.0)
    return f / f_sp


def _mel_to_hz(m, scale):
    if scale == "htk":
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    logstep = math.log(6.4) / 27.0
    return torch.where(m >= 15.0, 1000.0 * torch.exp(logstep * (m - 15.0)), freqs)


def mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max, naf_mode=False):
    """fb float32[n_fft//2+1, n_mels]; htk/no-norm by default, slaney/slaney in naf_mode
    (melspec.py:27-30)."""
    scale = "slaney" if naf_mode else "htk"
    n_freqs = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min, scale), _hz_to_mel(f_max, scale), n_mels + 2)
    f_pts = _mel_to_hz(m_pts, scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    if naf_mode:
        enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
        fb = fb * enorm.unsqueeze(0)
    return fb.to(torch.float32)


def melspec(x, params):
    """x float32 [B, L] (numpy or torch) -> numpy float32 [B, n_mels, 1 + L//hop]."""
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
                        params["f_min"], params["f_max"], naf)
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)
    mel = mel + (0.06 if naf else 1e-8)                             # melspec.py:38-41
    if mel_log == "log10":
        mel = torch.log10(mel)
    elif mel_log == "log":
        mel = torch.log(mel)
    if spec_norm == "max":
        mel = mel - torch.amax(mel, dim=(-2, -1), keepdim=True)     # melspec.py:48-49
    return mel.numpy()


def melspec_f64(x, params):
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
    fb = mel_filterbank(params["sample_rate"], n_fft, params["n_mels"],
                        params["f_min"], params["f_max"], False).double().numpy()
    mel = np.einsum("...tk,km->...mt", power, fb)
    return np.log(mel + 1e-8)
