"""Host mirror of the reference's segmenter interface (datautil/musicdata.py:9-104) for 16-bit PCM WAV input.
At the model sample rate the reference's resampler is the identity; files at other rates are resampled on the
device (Engine.pcm16_to_mono(pcm, sample_rate=...), pfann_amd/resample.py; parity with julius unpinned).  Other
container formats need ffmpeg: out of scope.

Two ways to consume a file:
  * `MusicDataset[i] -> (i, path, float32[n_seg, seg_len])`  -- the reference's contract
    (materialised unfold on the host; used by the operator-seam API and tests);
  * `MusicDataset.load_pcm_sr(i) -> (int16[n, ch], rate)`    -- raw PCM for the fused device
    path (Engine.pcm16_to_mono + Engine.embed_wav), which never materialises the unfold.
"""
import os
import struct
import time
import wave

import numpy as np
import torch

from .utils import get_logger, read_file_list


def read_wav_pcm16(path, alloc=None):
    """-> (int16[n_frames, n_ch], sample_rate).  16-bit PCM only (audio.py:130-149: the reference reads WAV files with
    the `wave` module and refuses other sample widths).  The RIFF chunks are walked here so that the samples can be read
    by ONE readinto() straight into the caller's buffer: alloc(n_int16) -> writable int16 numpy array (the decode
    workers hand out pinned memory); without alloc a fresh array is returned."""
    with open(path, "rb", buffering=0) as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise wave.Error("file does not start with RIFF id / not a WAVE file")
        fmt = None
        while True:
            ch = f.read(8)
            if len(ch) < 8:
                raise wave.Error("fmt chunk and/or data chunk missing")
            cid, size = ch[:4], struct.unpack("<I", ch[4:])[0]
            if cid == b"fmt ":
                body = f.read(size + (size & 1))
                if len(body) < 16:
                    raise wave.Error("fmt chunk too short")
                tag, n_ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
                if tag == 0xFFFE and len(body) >= 26:                     # WAVE_FORMAT_EXTENSIBLE: the sub-format's tag
                    tag = struct.unpack("<H", body[24:26])[0]
                if tag != 1:
                    raise wave.Error("unknown format: %r" % (tag,))
                if n_ch < 1:
                    raise wave.Error("bad # of channels")
                if (bits + 7) // 8 != 2:
                    raise NotImplementedError("wave stream currently only supports 16bit wav")
                fmt = (n_ch, sr)
            elif cid == b"data":
                if fmt is None:
                    raise wave.Error("data chunk before fmt chunk")
                n_ch, sr = fmt
                # streamed WAVs (ffmpeg pipe output) declare 0xFFFFFFFF or 0 bytes: never trust the header beyond what the
                # file holds (csrc/wavio.hip clamps the same way), or alloc() would pin gigabytes for a few real samples
                try:
                    left = max(os.fstat(f.fileno()).st_size - f.tell(), 0)
                    size = min(size, left)
                except OSError:
                    pass
                n = size // (2 * n_ch) * n_ch                               # whole frames only
                buf = alloc(n) if alloc is not None else np.empty(n, dtype=np.int16)
                got = f.readinto(memoryview(buf).cast("B")[: n * 2]) if n else 0
                while 0 < got < n * 2:                                      # short reads (pipes, network file systems)
                    more = f.readinto(memoryview(buf).cast("B")[got: n * 2])
                    if not more:
                        break
                    got += more
                n = (got or 0) // (2 * n_ch) * n_ch                         # truncated file: what is there
                return buf[:n].reshape(-1, n_ch), sr
            else:
                f.seek(size + (size & 1), 1)


class MusicDataset:
    def __init__(self, file_list, params):
        self.params = params
        self.sample_rate = params["sample_rate"]
        self.segment_size = int(self.sample_rate * params["segment_size"])
        self.hop_size = int(self.sample_rate * params["hop_size"])
        self.frame_shift_mul = params["indexer"].get("frame_shift_mul", 1)
        self.files = read_file_list(file_list) if isinstance(file_list, str) else list(file_list)

    @property
    def hop(self):
        return self.hop_size // self.frame_shift_mul

    def n_segments(self, n_samples):
        n = max(n_samples, self.segment_size)
        return (n - self.segment_size) // self.hop + 1

    def load_pcm_sr(self, index, alloc=None):
        """-> (int16 [n, ch] as stored, the file's sample rate).  Stateless (decode workers call it concurrently, each
        with an `alloc` that hands out pinned memory); the device path resamples when the rate is not the model's:
        Engine.pcm16_to_mono(pcm, sample_rate=sr)."""
        return read_wav_pcm16(self.files[index], alloc)

    def load_pcm(self, index):
        """-> int16 [n, ch]; kept for callers that know their files are at the model's rate (raises otherwise)."""
        pcm, sr = self.load_pcm_sr(index)
        if sr != self.sample_rate:
            raise ValueError("%s is at %d Hz, not %d: use load_pcm_sr and hand the rate to Engine.pcm16_to_mono"
                             % (self.files[index], sr, self.sample_rate))
        return pcm

    def unsafe_getitem(self, index):
        log = get_logger()
        t0 = time.time()
        pcm, sr = self.load_pcm_sr(index)
        if sr != self.sample_rate:                             # host-side dataset contract: native rate only
            raise NotImplementedError("resampling %d -> %d Hz runs on the device path only" % (sr, self.sample_rate))
        t1 = time.time()
        x = np.multiply(pcm, 1 / 32768, dtype=np.float32).T.copy()
        if x.shape[0] == 2:                                    # musicdata.py:72-79
            pow1 = np.mean((x[0] - x[1]) ** 2, dtype=np.float32)
            pow2 = np.mean((x[0] + x[1]) ** 2, dtype=np.float32)
            if pow1 > pow2 * 1000:
                log.warning("fake stereo with opposite phase detected: %s", self.files[index])
                x[1] *= -1
        wav = x.mean(axis=0, dtype=np.float32)
        if wav.shape[0] < self.segment_size:
            wav = np.pad(wav, (0, self.segment_size - wav.shape[0]))
        n_seg = (wav.shape[0] - self.segment_size) // self.hop + 1
        seg = np.lib.stride_tricks.as_strided(wav, (n_seg, self.segment_size),
                                              (wav.strides[0] * self.hop, wav.strides[0]))
        seg = seg - seg.mean(axis=1, dtype=np.float32, keepdims=True)
        log.info("load %.6fs stereo to mono %.6fs", t1 - t0, time.time() - t1)
        return index, self.files[index], torch.from_numpy(np.ascontiguousarray(seg, dtype=np.float32))

    def __getitem__(self, index):
        try:
            return self.unsafe_getitem(index)
        except Exception as x:                                  # musicdata.py:95-101
            get_logger().exception(x)
            return index, self.files[index], torch.zeros(0, self.segment_size)

    def __len__(self):
        return len(self.files)

    def __iter__(self):                     # __getitem__ swallows every exception, IndexError included
        return (self[i] for i in range(len(self)))
