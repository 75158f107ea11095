"""configs/*.json dict -> the C ABI's pfann_config (torch-free: the tools' start-up thread builds the context while the
interpreter is still importing torch, pfann_amd/prewarm.py)."""
from . import lib as _l
from .synth import model_dims


def config_from_params(params, max_batch=512):
    """configs/*.json dict -> pfann_config (keys read exactly where the reference reads them:
    builder.py:46-51, melspec.py:52-63, model.py:135-140)."""
    d, h, u, F, T = model_dims(params)
    m = params["model"]
    naf = params.get("naf_mode", False)
    cfg = _l.Config()
    cfg.segment_len = int(params["segment_size"] * params["sample_rate"])
    cfg.stft_n = params["stft_n"]
    cfg.stft_hop = params["stft_hop"]
    cfg.n_mels = params["n_mels"]
    cfg.power = 1 if naf else 2
    cfg.pad_reflect = 0 if naf else 1
    cfg.log_mode = {"log": 1, "log10": 2}.get(params.get("mel_log", "log"), 0)
    cfg.spec_norm_max = 1 if params.get("spec_norm", "l2") == "max" else 0
    cfg.log_eps = 0.06 if naf else 1e-8
    cfg.d, cfg.h, cfg.u = d, h, u
    cfg.fuller = 1 if m.get("fuller", False) else 0
    cfg.activation = {"ReLU": 0, "ELU": 1}[m.get("conv_activation", "ReLU")]
    cfg.relu_after_bn = 1 if m.get("relu_after_bn", True) else 0
    strides = m.get("strides")
    for i in range(8):
        cfg.stride_t[i] = 2 if strides is None else strides[i][0][1]
        cfg.stride_f[i] = 2 if strides is None else strides[i][1][0]
    cfg.max_batch = max_batch
    return cfg
