"""Oracle a1: WAV -> float32[n_seg, segment_size] (reference datautil/musicdata.py:21-93).

Restated for 16-bit PCM WAV input, mono or stereo.  At the model sample rate the reference's julius
resampler is the identity and its minute-wise chunking (musicdata.py:33-65) reduces to a plain
concatenation (pinned: tests/golden/segmenter.npz); other rates go through oracle/resample.py (parity unpinned:
julius is absent).
"""
import wave

import numpy as np


def read_wav_int16(path):
    """-> (int16[n_frames, n_ch], sample_rate); raises on anything but 16-bit PCM
    (reference datautil/audio.py:130-149 supports only 16-bit in its wave path)."""
    with wave.open(path, "rb") as w:
        if w.getsampwidth() != 2:
            raise NotImplementedError("16-bit wav only")
        n_ch, sr, n = w.getnchannels(), w.getframerate(), w.getnframes()
        pcm = np.frombuffer(w.readframes(n), dtype=np.int16).reshape(-1, n_ch)
    return pcm, sr


def pcm_to_mono(pcm, file_sr=None, sr=None, resample_table=None):
    """int16[n, ch] -> float32[n]: scale 1/32768 in fp32 (musicdata.py:48), resampling when the file's rate is not the
    model's (musicdata.py:28-65), fake-stereo fix (musicdata.py:74-79), channel mean (musicdata.py:80)."""
    x = np.multiply(pcm, 1 / 32768, dtype=np.float32).T.copy()  # [ch, n]
    if file_sr is not None and sr is not None and file_sr != sr:
        from . import resample
        x = np.ascontiguousarray(resample.resample_chunked(x, file_sr, sr, resample_table))
    if x.shape[0] == 2:
        pow1 = np.mean((x[0] - x[1]) ** 2, dtype=np.float32)
        pow2 = np.mean((x[0] + x[1]) ** 2, dtype=np.float32)
        if pow1 > pow2 * 1000:
            x[1] *= -1
    return x.mean(axis=0, dtype=np.float32)


def segment(wav, segment_size, hop):
    """float32[n] -> float32[n_seg, segment_size]: zero-pad short (musicdata.py:82-84),
    unfold (musicdata.py:87), per-segment mean removal (musicdata.py:88)."""
    wav = np.asarray(wav, dtype=np.float32)
    if wav.shape[0] < segment_size:
        wav = np.pad(wav, (0, segment_size - wav.shape[0]))
    n_seg = (wav.shape[0] - segment_size) // hop + 1
    idx = np.arange(n_seg)[:, None] * hop + np.arange(segment_size)[None, :]
    seg = wav[idx]
    return (seg - seg.mean(axis=1, dtype=np.float32, keepdims=True)).astype(np.float32)


def load_segments(path, params):
    """Full a1 for one file; any failure -> float32[0, segment_size] (musicdata.py:95-101)."""
    sr = params["sample_rate"]
    seg_n = int(sr * params["segment_size"])
    hop = int(sr * params["hop_size"]) // params["indexer"].get("frame_shift_mul", 1)
    try:
        pcm, file_sr = read_wav_int16(path)
        return segment(pcm_to_mono(pcm, file_sr, sr), seg_n, hop)
    except Exception:
        return np.zeros((0, seg_n), dtype=np.float32)
