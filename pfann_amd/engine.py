"""Engine: one pfann_ctx (front-end + encoder) on one GPU, driven with torch tensors.

PyTorch is plumbing here (device memory + streams); all arithmetic runs in
libpfann_amd.so through the C ABI.
"""
import ctypes

import numpy as np
import torch

from . import lib as _l
from .config import config_from_params          # noqa: F401 -- (re-exported: callers import it from here)
from .synth import model_dims


def _mel_to_hz_tensor(m, scale):
    if scale == "htk":
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    import math
    logstep = math.log(6.4) / 27.0
    return torch.where(m >= 15.0, 1000.0 * torch.exp(logstep * (m - 15.0)), (200.0 / 3) * m)


def _hz_to_mel(f, scale):
    import math
    if scale == "htk":
        return 2595.0 * math.log10(1.0 + f / 700.0)
    if f >= 1000.0:
        return 15.0 + math.log(f / 1000.0) / (math.log(6.4) / 27.0)
    return f / (200.0 / 3)


def mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max, naf_mode=False):
    """Triangular mel bank fb[n_fft//2+1, n_mels] in fp32 torch ops: what
    torchaudio.transforms.MelSpectrogram(mel_scale='htk', norm=None) (default mode) or
    (mel_scale='slaney', norm='slaney') (naf_mode) builds for reference melspec.py:19-31."""
    scale = "slaney" if naf_mode else "htk"
    n_freqs = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min, scale), _hz_to_mel(f_max, scale), n_mels + 2)
    f_pts = _mel_to_hz_tensor(m_pts, scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    fb = torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)
    if naf_mode:
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb.to(torch.float32).contiguous()


class Engine:
    def __init__(self, params, device=0, max_batch=512, encoder_only=False):
        """encoder_only: the context serves `encode` alone (FpNetwork built without a full config, model.py:132-146
        knows only d, h, u, F, T and the "model" block); the front-end entry points then refuse to run."""
        _l.require_gpu()
        self.encoder_only = encoder_only
        self.lib = _l.load()
        self.params = params
        self.device = torch.device("cuda", device if isinstance(device, int) else (device.index or 0))
        self.cfg = config_from_params(params, max_batch)
        self.d, self.h, self.u, self.F, self.T = model_dims(params)
        self.seg_len = self.cfg.segment_len
        # the tools' start-up thread may already have built this very context, weights included (prewarm.py)
        from . import prewarm
        self.handle = prewarm.take_engine(self.cfg, self.device.index) if not encoder_only else None
        self.weights_loaded = self.handle is not None
        if self.handle is None:
            self.handle = self.lib.pfann_create(ctypes.byref(self.cfg), self.device.index)
        self._resample_tables = {}
        # a callable run (on the caller's stream) before every launch of the log-mel front end; the sharded tools hang
        # dist.ShardedIndex.hold_front_end here when the exchange stream is on (see there for why)
        self.before_front_end = None
        if not self.handle:
            raise _l.PfannError("pfann_create failed: " + _l.last_error())
        n_frames = 1 + self.seg_len // self.cfg.stft_hop
        if n_frames != self.T:
            raise _l.PfannError("stft yields %d frames but the encoder expects T=%d" % (n_frames, self.T))
        fb = mel_filterbank(params["sample_rate"], params["stft_n"], params["n_mels"], params["f_min"],
                            params["f_max"], params.get("naf_mode", False)).numpy()
        _l.check(self.lib.pfann_set_melbank(self.handle, fb.ctypes.data, fb.shape[0], fb.shape[1]),
                 "pfann_set_melbank")

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.pfann_destroy(h)

    # ---- weights ---------------------------------------------------------------------
    def load_state_dict(self, sd):
        """sd: name -> torch tensor / numpy array, reference state_dict names (68 tensors)."""
        for name, val in sd.items():
            arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            _l.check(self.lib.pfann_load_weight(self.handle, name.encode(), arr.ctypes.data, arr.size),
                     "pfann_load_weight(%s)" % name)
        missing = self.lib.pfann_weights_missing(self.handle)
        if missing:
            raise _l.PfannError("state_dict incomplete: %d tensors missing" % missing)

    # ---- operators -------------------------------------------------------------------
    def _stream(self):
        return _l.current_stream_ptr(self.device)

    def _prep(self, x):
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        return x.to(self.device, torch.float32).contiguous()

    def _need_front_end(self, what):
        if self.encoder_only:
            raise _l.PfannError("%s: this Engine was built encoder-only (FpNetwork without a full config); build it from "
                                "the whole configs/*.json dict" % what)

    def melspec(self, segs):
        """MelSpec.forward: [..., seg_len] -> [..., n_mels, T]."""
        self._need_front_end("melspec")
        if self.before_front_end is not None:
            self.before_front_end()
        x = self._prep(segs)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.seg_len)
        out = torch.empty((x2.shape[0], self.F, self.T), device=self.device, dtype=torch.float32)
        if x2.shape[0]:
            _l.check(self.lib.pfann_melspec(self.handle, x2.data_ptr(), x2.shape[0], self.seg_len, 0,
                                            out.data_ptr(), self._stream()), "pfann_melspec")
        return out.reshape(*lead, self.F, self.T)

    def encode(self, mel, norm=True):
        """FpNetwork.forward: [B, F, T] -> [B, d]."""
        x = self._prep(mel)
        assert x.dim() == 3 and x.shape[1] == self.F and x.shape[2] == self.T, x.shape
        out = torch.empty((x.shape[0], self.d), device=self.device, dtype=torch.float32)
        if x.shape[0]:
            _l.check(self.lib.pfann_encode(self.handle, x.data_ptr(), x.shape[0], out.data_ptr(),
                                           1 if norm else 0, self._stream()), "pfann_encode")
        return out

    def embed_wav(self, wav, hop, n_seg=None, norm=True):
        """Fused path: mono float wav [L] on device -> [n_seg, d] embeddings of the windows
        wav[i*hop : i*hop+seg_len] (mean removal, mel, encoder, optional L2 norm)."""
        self._need_front_end("embed_wav")
        if self.before_front_end is not None:
            self.before_front_end()
        w = self._prep(wav).reshape(-1)
        if w.shape[0] < self.seg_len:                      # musicdata.py:82-84
            w = torch.nn.functional.pad(w, (0, self.seg_len - w.shape[0]))
        if n_seg is None:
            n_seg = (w.shape[0] - self.seg_len) // hop + 1
        assert (n_seg - 1) * hop + self.seg_len <= w.shape[0]
        out = torch.empty((n_seg, self.d), device=self.device, dtype=torch.float32)
        if n_seg:
            _l.check(self.lib.pfann_segment_embed(self.handle, w.data_ptr(), n_seg, hop, out.data_ptr(),
                                                  1 if norm else 0, self._stream()), "pfann_segment_embed")
        return out

    def embed_windows(self, wav, starts, norm=True):
        """wav: device float mono buffer (many recordings back to back); starts: int64 window
        start offsets (device tensor or array) -> [len(starts), d]."""
        self._need_front_end("embed_windows")
        if self.before_front_end is not None:
            self.before_front_end()
        w = self._prep(wav).reshape(-1)
        if isinstance(starts, torch.Tensor):
            st = starts.to(self.device, torch.int64).contiguous()
        else:
            st = _l.upload_async(starts, self.device, np.int64)
        B = st.shape[0]
        out = torch.empty((B, self.d), device=self.device, dtype=torch.float32)
        if B:
            _l.check(self.lib.pfann_segment_embed_at(self.handle, w.data_ptr(), st.data_ptr(), B, out.data_ptr(),
                                                     1 if norm else 0, self._stream()), "pfann_segment_embed_at")
        return out

    def pcm16_to_mono(self, pcm, sample_rate=None):
        """int16 [n] or [n, ch] (torch / numpy) -> float32 mono [n] on device.  sample_rate: the file's rate when it is
        not the model's (musicdata.py:28-65): the signal is resampled on the device first (pfann_amd/resample.py).  A tensor in PINNED host memory is uploaded
        asynchronously (the host does not wait for the stream, which may still be busy with the previous batch's encoder):
        the caller keeps it unchanged until the stream has passed this point."""
        p = pcm if isinstance(pcm, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(pcm))
        p = p.to(self.device, torch.int16, non_blocking=bool(p.device.type == "cpu" and p.is_pinned())).contiguous()
        n_ch = 1 if p.dim() == 1 else p.shape[1]
        n = p.shape[0]
        if sample_rate is not None and int(sample_rate) != int(self.params["sample_rate"]) and n:
            from . import resample
            key = resample.reduced_rates(sample_rate, self.params["sample_rate"])
            if key not in self._resample_tables:
                tab, old, new, width = resample.filter_table(*key)
                self._resample_tables[key] = (torch.from_numpy(tab).to(self.device), old, new, width)
            tab, old, new, width = self._resample_tables[key]
            plan, n_out = resample.piece_plan(n, int(sample_rate), int(self.params["sample_rate"]))
            plan_dev = _l.upload_async(plan, self.device, np.int64)
            tmp = torch.empty((n_ch, n_out), device=self.device, dtype=torch.float32)
            out = torch.empty((n_out,), device=self.device, dtype=torch.float32)
            if n_out:
                _l.check(self.lib.pfann_resample_to_mono(self.handle, p.data_ptr(), n_ch, tab.data_ptr(), old, new, width,
                                                         plan_dev.data_ptr(), plan.shape[0], n_out, tmp.data_ptr(),
                                                         out.data_ptr(), self._stream()), "pfann_resample_to_mono")
            return out
        out = torch.empty((n,), device=self.device, dtype=torch.float32)
        if n:
            _l.check(self.lib.pfann_pcm16_to_mono(self.handle, p.data_ptr(), n, n_ch, out.data_ptr(),
                                                  self._stream()), "pfann_pcm16_to_mono")
        return out

    def warmup(self, windows=512, group_hop=0):
        """Allocates the activation workspace (max_batch segments: 29 GB at 9728) and makes the runtime load every
        kernel of the path by embedding `windows` windows of silence once.  The CLIs call it while "loading model...",
        so that first-use costs are not billed to the first batch of files.  group_hop (samples between the windows of
        the files the tools are about to read): also reserve the device blocks a launch group of `windows` windows needs."""
        n = max(1, min(int(windows), int(self.cfg.max_batch)))
        wav = torch.zeros(self.seg_len + (n - 1) * 16, device=self.device, dtype=torch.float32)
        self.embed_windows(wav, np.arange(n, dtype=np.int64) * 16)
        self.pcm16_to_mono(torch.zeros((64, 1), dtype=torch.int16))
        torch.cuda.synchronize(self.device)
        if group_hop:
            # the caching allocator gets the blocks a launch group's PCM and mono signal will want (two groups in flight),
            # so that the first groups of a run do not each stop for a hipMalloc (8-10 ms per new size)
            samples = int(n * int(group_hop) * 1.12) + self.seg_len        # (every file ends on a partly used hop: 10 s -> 19 windows)
            keep = [torch.empty(samples, dtype=dt, device=self.device) for dt in (torch.int16, torch.float32) for _ in range(2)]
            del keep

    def set_fused_layernorm(self, on=True):
        """-> True if the LayerNorm-fused GEMM path is now active."""
        return bool(self.lib.pfann_set_fused_layernorm(self.handle, 1 if on else 0))

    def set_plan_batch(self, n=0):
        """Kernel variants chosen as for a batch of n segments whatever the call's own size (0: per call, the default): a
        segment's fingerprint then has the same bits in every batch (include/pfann_amd.h: pfann_set_plan_batch)."""
        return int(self.lib.pfann_set_plan_batch(self.handle, int(n)))

    def set_encoder_precision(self, mode=0):
        """0 = fp32 MFMA (default, exact); 1 = 3-term fp16 split on the fp16 MFMA.  -> mode in effect."""
        return int(self.lib.pfann_set_encoder_precision(self.handle, int(mode)))

    # ---- verification taps -----------------------------------------------------------
    def debug_keep(self, on=True):
        self.lib.pfann_debug_keep(self.handle, 1 if on else 0)

    def debug_activation(self, idx, B):
        from .synth import layer_plan
        L = layer_plan(self.params)[idx // 2]
        if idx % 2 == 0:
            shape = (L["co"], L["F"], L["T1"])
        else:
            shape = (L["co"], L["F2"], L["T1"])
        buf = np.empty((B,) + shape, dtype=np.float32)
        torch.cuda.synchronize(self.device)
        n = _l.check(self.lib.pfann_debug_activation(self.handle, idx, B, buf.ctypes.data, buf.size),
                     "pfann_debug_activation")
        return buf.reshape(-1)[:n].reshape((-1,) + shape)
