"""Oracle a3-a5: log-mel float32[B, F, T] -> embedding float32[B, d]
(reference model.py:54-73 SeparableConv2d.forward, 101-106 MyF.forward, 122-130
MyG.forward, 148-153 FpNetwork.forward), restated functionally over a state_dict.
"""
import numpy as np
import torch
import torch.nn.functional as Fn



def model_dims(params):
    """(d, h, u, F, T) as builder.py:46-51 derives them (T = ceil(segment samples / stft_hop))."""
    m = params["model"]
    segn = int(params["segment_size"] * params["sample_rate"])
    return m["d"], m["h"], m["u"], params["n_mels"], (segn + params["stft_hop"] - 1) // params["stft_hop"]


def layer_plan(params):
    """The oracle's OWN statement of the 8 separable blocks (kept here so the checker does not share
    its padding/stride plan with the product): channel ladder and per-block strides of model.py:79-93,
    "same"-style padding of model.py:15-30 (pad = (in-1)//s*s + 3 - in, left = pad//2, rest right)."""
    d, h, u, F, T = model_dims(params)
    m = params["model"]
    chans = [1, d, d, 2 * d, 2 * d, 4 * d, 4 * d, h, h]
    strides = m.get("strides") or [[(1, 2), (2, 1)]] * 8
    out = []
    for i in range(8):
        s_t, s_f = int(strides[i][0][1]), int(strides[i][1][0])
        pt = (T - 1) // s_t * s_t + 3 - T
        pf = (F - 1) // s_f * s_f + 3 - F
        out.append(dict(ci=chans[i], co=chans[i + 1], s_t=s_t, s_f=s_f, pad1=(pt // 2, pt - pt // 2),
                        pad2=(pf // 2, pf - pf // 2), depthwise=not m.get("fuller", False)))
        T, F = (T - 1) // s_t + 1, (F - 1) // s_f + 1
    assert (F, T) == (1, 1), "output must be 1x1 (model.py:94)"
    return out


def _act(x, name):
    return Fn.relu(x) if name == "ReLU" else Fn.elu(x)


def encode(mel, sd, params, norm=True, taps=None, dtype=np.float32, device=None):
    """mel [B,F,T]; sd: name -> numpy/torch tensors (reference state_dict names).
    taps: optional list that receives each of the 16 post-(LN,act) activations (NCHW numpy).
    dtype: np.float32 = the reference's arithmetic (torch CPU fp32, what parity is judged against);
    np.float64 = the SAME op sequence carried out in double on the float32 weights -- the "exact" value
    tools/embedding_error_budget.py triangulates the GPU path and the fp32 oracle against (neither is the truth:
    both round; this says by how much each).  The result keeps that dtype.
    device: None = the host (what every parity check uses).  A torch device (e.g. "cuda") is accepted ONLY for the
    float64 yardstick: the same op sequence executed by torch's own float64 kernels there (im2col + dgemm, not the
    product's HIP kernels) -- the host's float64 evaluation costs as much as the whole fp32 oracle, and the GPU suite's
    time goes to the oracle; tools/embedding_error_budget.py cross-checks the two evaluations against each other.
    """
    assert device is None or dtype == np.float64, "the fp32 parity oracle runs on the host"
    m = params["model"]
    act = m.get("conv_activation", "ReLU")
    after_bn = m.get("relu_after_bn", True)
    d, h, u, _, _ = model_dims(params)
    g = lambda k: torch.as_tensor(np.asarray(np.asarray(sd[k], dtype=np.float32), dtype=dtype), device=device)
    with torch.no_grad():
        x = (mel.to(device) if isinstance(mel, torch.Tensor) else torch.as_tensor(np.asarray(mel, dtype=dtype), device=device)).unsqueeze(1)   # model.py:102
        for i, L in enumerate(layer_plan(params)):
            p = "f.convs.%d." % i
            x = Fn.pad(x, (L["pad1"][0], L["pad1"][1], 0, 0))                  # model.py:56
            x = Fn.conv2d(x, g(p + "conv1.weight"), g(p + "conv1.bias"), stride=(1, L["s_t"]))
            w, b = g(p + "ln1.weight"), g(p + "ln1.bias")
            if after_bn:                                                       # model.py:58-63
                x = _act(Fn.layer_norm(x, w.shape, w, b, 1e-5), act)
            else:
                x = Fn.layer_norm(_act(x, act), w.shape, w, b, 1e-5)
            if taps is not None:
                taps.append(x.cpu().numpy().copy())
            x = Fn.pad(x, (0, 0, L["pad2"][0], L["pad2"][1]))                  # model.py:65
            x = Fn.conv2d(x, g(p + "conv2.weight"), g(p + "conv2.bias"), stride=(L["s_f"], 1),
                          groups=L["co"] if L["depthwise"] else 1)             # model.py:26-29
            w, b = g(p + "ln2.weight"), g(p + "ln2.bias")
            if after_bn:
                x = _act(Fn.layer_norm(x, w.shape, w, b, 1e-5), act)
            else:
                x = Fn.layer_norm(_act(x, act), w.shape, w, b, 1e-5)
            if taps is not None:
                taps.append(x.cpu().numpy().copy())
        x = x.reshape(-1, h, 1)                                                # model.py:123
        x = Fn.conv1d(x, g("g.linear1.weight"), g("g.linear1.bias"), groups=d)
        x = Fn.elu(x)
        x = Fn.conv1d(x, g("g.linear2.weight"), g("g.linear2.bias"), groups=d)
        x = x.reshape(-1, d)
        if norm:
            x = Fn.normalize(x, p=2.0)                                         # model.py:128-129
        return x.cpu().numpy()
