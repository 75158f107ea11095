"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle and the
golden vectors generated from the reference.  Run with `pytest -m gpu` on an MI355X."""
import ctypes
import json
import os

import numpy as np
import pytest

import make_golden as mg
from pfann_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfg(name):
    return json.load(open(os.path.join(REPO, "configs", name + ".json")))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


# ----------------------------------------------------------------------------- front-end
def _segments(params):
    from oracle import segmenter
    pcm = synth.make_song(3, seconds=4.0)
    segs = segmenter.segment(segmenter.pcm_to_mono(pcm[:, None]), 8000, 4000)           # 7 musical segments
    noise = synth.normal(5, "t/noise", 8000).reshape(1, 8000) * 0.1
    tone = (0.5 * np.sin(2 * np.pi * 1000.0 * np.arange(8000) / 8000.0)).astype(np.float32).reshape(1, 8000)
    quiet = (segs[:1] * 1e-4).astype(np.float32)
    zero = np.zeros((1, 8000), np.float32)
    return np.concatenate([segs, noise, tone, quiet, zero]).astype(np.float32)


def test_melspec_vs_oracle(torch_cuda):
    """a2.  Tolerance: fp32 STFT implementations differ by ~1e-6 of the frame's peak
    magnitude, which is a large RELATIVE error only in bins ~1e-7 below the peak; so the
    log-mel is compared (i) tightly where the bin carries energy and (ii) in linear power
    relative to the segment's peak everywhere."""
    from oracle import melspec as om
    from pfann_amd.engine import Engine
    params = cfg("default")
    eng = Engine(params, 0)
    x = _segments(params)
    got = eng.melspec(torch_cuda.as_tensor(x).cuda()).cpu().numpy()
    # the kernel's arithmetic: the oracle runs with the very bank the kernel was given (fp32, built the way torchaudio
    # builds it); the oracle's own bank (float64 from the definition) follows below
    from pfann_amd.engine import mel_filterbank
    bank = mel_filterbank(params["sample_rate"], params["stft_n"], params["n_mels"], params["f_min"], params["f_max"]).numpy()
    ref32 = om.melspec(x, params, bank)
    ref64 = om.melspec_f64(x, params, bank)
    assert got.shape == ref32.shape == (x.shape[0], 256, 32)
    p_got, p_ref = np.exp(got.astype(np.float64)), np.exp(ref64)
    peak = p_ref.max(axis=(1, 2), keepdims=True)
    lin_err = np.abs(p_got - p_ref) / peak
    loud = ref64 > (np.log(peak) - 11.5)             # within 1e-5 of the segment's peak power
    log_err = np.abs(got - ref64)
    print("mel: max lin err/peak %.3e, max log err (loud bins) %.3e, vs fp32-oracle max %.3e" %
          (lin_err.max(), log_err[loud].max(), np.abs(got - ref32)[loud].max()))
    assert lin_err.max() < 2e-6
    assert log_err[loud].max() < 2e-3
    # the torch.stft restatement is no closer to the fp64 truth than the kernel is
    assert log_err[loud].max() <= max(3 * np.abs(ref32 - ref64)[loud].max(), 1e-4)
    # against the oracle's OWN bank (independent float64 statement; tests/test_host.py bounds the per-weight gap by
    # 4.5e-5): in linear power relative to the segment's peak the difference is a few bank gaps
    own = np.exp(om.melspec_f64(x, params))
    own_err = np.abs(p_got - own) / own.max(axis=(1, 2), keepdims=True)
    print("mel vs the oracle's own float64 bank: max lin err/peak %.3e" % own_err.max())
    assert own_err.max() < 2e-4


def test_melspec_fused_mean_removal_matches_operator_form(torch_cuda):
    from oracle import segmenter
    from pfann_amd.engine import Engine
    params = cfg("default")
    eng = Engine(params, 0)
    pcm = synth.make_song(4, seconds=3.0)
    wav = segmenter.pcm_to_mono(pcm[:, None]) + np.float32(0.05)      # DC offset: mean removal matters
    segs = segmenter.segment(wav, 8000, 4000)
    a = eng.melspec(torch_cuda.as_tensor(segs).cuda()).cpu().numpy()
    lib = eng.lib
    w = torch_cuda.as_tensor(wav).cuda()
    out = torch_cuda.empty((segs.shape[0], 256, 32), device="cuda")
    rc = lib.pfann_melspec(eng.handle, w.data_ptr(), segs.shape[0], 4000, 1, out.data_ptr(), None)
    assert rc == 0
    torch_cuda.cuda.synchronize()
    b = out.cpu().numpy()
    loud = a > a.max() - 11.5
    assert np.abs(a - b)[loud].max() < 1e-3


@pytest.mark.parametrize("variant", ["naf", "log10max", "fft512"])
def test_melspec_variants(torch_cuda, variant):
    """naf_mode / log10 / spec_norm='max' (melspec.py:27-30,38-49) against the oracle; "fft512": another
    STFT size (radix-2 LDS FFT instead of the 8x8x8 register one, 63 frames: whole-tile output path)."""
    from oracle import melspec as om
    from pfann_amd.engine import Engine
    params = cfg("default")
    if variant == "naf":
        params.update(naf_mode=True, mel_log="log10")
    elif variant == "fft512":
        params.update(stft_n=512, stft_hop=128, n_mels=96)
    else:
        params.update(mel_log="log10", spec_norm="max")
    eng = Engine(params, 0)
    x = _segments(params)[:8]
    got = eng.melspec(torch_cuda.as_tensor(x).cuda()).cpu().numpy()
    from pfann_amd.engine import mel_filterbank
    ref = om.melspec(x, params, mel_filterbank(params["sample_rate"], params["stft_n"], params["n_mels"], params["f_min"],
                                              params["f_max"], params.get("naf_mode", False)).numpy())
    err = np.abs(got - ref)
    loud = ref > ref.max(axis=(1, 2), keepdims=True) - 4.5
    print(variant, "max err loud", err[loud].max(), "overall", err.max())
    assert err[loud].max() < 2e-3


def test_pcm16_to_mono_and_segment_embed(torch_cuda):
    """a1 on device: int16 -> mono (bit-exact), fake-stereo flip, then the fused
    window/mean/mel/encode path equals the operator-seam path."""
    from oracle import segmenter
    from pfann_amd.engine import Engine
    params = cfg("tiny")
    eng = Engine(params, 0)
    eng.load_state_dict(synth.make_state_dict(params))
    ins = mg.segmenter_inputs()
    for name in ("stereo12", "fakestereo12", "short", "len_8001"):
        pcm = ins[name]
        pcm2 = pcm if pcm.ndim == 2 else pcm[:, None]
        want = segmenter.pcm_to_mono(pcm2)
        got = eng.pcm16_to_mono(pcm2).cpu().numpy()
        assert np.array_equal(got, want), name
        segs = segmenter.segment(want, 8000, 4000)
        e_ref = eng.encode(eng.melspec(torch_cuda.as_tensor(segs).cuda())).cpu().numpy()
        e_fused = eng.embed_wav(torch_cuda.as_tensor(want).cuda(), 4000).cpu().numpy()
        assert e_fused.shape == e_ref.shape == (segs.shape[0], 16)
        assert np.abs(e_fused - e_ref).max() < 1e-4, name


# ------------------------------------------------------------------------------- encoder
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", ["tiny", "nafstyle", "n640d64", "seg", "default", "elu_full", "strides_pow2", "strides_np2"])
def test_encoder_vs_reference_golden(torch_cuda, name, fused):
    """a3-a5: embeddings within 1e-4 of the reference's own outputs (golden), and every one
    of the 16 sub-layer activations against the oracle -- for both encoder paths
    (LayerNorm fused into the GEMMs / separate LayerNorm kernels)."""
    from oracle import encoder as oe
    from pfann_amd.engine import Engine
    z = np.load(os.path.join(G, "encoder_%s.npz" % name))
    params = json.loads(str(z["params"]))
    _, _, _, F, T = synth.model_dims(params)
    sd = synth.make_state_dict(params, seed=123)
    eng = Engine(params, 0, max_batch=8)
    eng.load_state_dict(sd)
    if eng.set_fused_layernorm(fused) != fused:
        # depthwise conv2 (fuller=false) or a non-power-of-two Fo*To (strides_np2): the library keeps the
        # separate-LayerNorm kernels and reports it through the return value (and a one-line notice)
        assert name in ("nafstyle", "n640d64", "seg", "strides_np2"), "fused path unexpectedly refused for " + name
        pytest.skip("fused LayerNorm path not available for this model (reported by pfann_set_fused_layernorm)")
    eng.debug_keep(True)
    x = mg.encoder_inputs(F, T)
    xt = torch_cuda.as_tensor(x).cuda()
    emb = eng.encode(xt, norm=True).cpu().numpy()
    taps_ref = []
    oe.encode(x, sd, params, norm=True, taps=taps_ref)
    worst = 0.0
    for i, tr in enumerate(taps_ref):
        tg = eng.debug_activation(i, x.shape[0])
        assert tg.shape == tr.shape, (i, tg.shape, tr.shape)
        e = np.abs(tg - tr).max()
        worst = max(worst, e)
        assert e < 2e-4 * max(1.0, np.abs(tr).max()), "sub-layer %d: max err %g" % (i, e)
    raw = eng.encode(xt, norm=False).cpu().numpy()
    print(name, "emb err %.3e raw err %.3e worst tap err %.3e" %
          (np.abs(emb - z["emb"]).max(), np.abs(raw - z["raw"]).max(), worst))
    assert np.abs(emb - z["emb"]).max() < 1e-4
    assert np.abs(raw - z["raw"]).max() < 1e-4 * max(1.0, np.abs(z["raw"]).max())
    # without verification taps the fused path folds the first conv into the second one's loader
    # (the 2 MiB/segment tensor is never written; LayerNorm statistics from the Gram form)
    eng.debug_keep(False)
    emb2 = eng.encode(xt, norm=True).cpu().numpy()
    assert np.abs(emb2 - emb).max() < 2e-6
    assert np.abs(emb2 - z["emb"]).max() < 1e-4


@pytest.mark.parametrize("fused", [True, False])
def test_encoder_large_mean_activations(torch_cuda, fused):
    """LayerNorm statistics under a large mean-to-std ratio (the fused path combines fp32 per-tile
    (sum, sum of squares) partials as E[z^2] - mean^2): LayerNorm biases x20 (activations entering every conv sit
    far from zero) and every conv bias +3 (pre-LayerNorm mean several std away from 0).  Same 1e-4 embedding bar
    against the CPU oracle (torch layer_norm, two-pass), and every sub-layer activation within 2e-4 relative."""
    from oracle import encoder as oe
    from pfann_amd.engine import Engine
    params = cfg("default")
    _, _, _, F, T = synth.model_dims(params)
    sd = synth.make_state_dict(params, seed=321)
    for name in sd:
        if ".ln" in name and name.endswith(".bias"):
            sd[name] = (sd[name] * 20.0).astype(np.float32)
        if ".conv" in name and name.endswith(".bias"):
            sd[name] = (sd[name] + 3.0).astype(np.float32)
    eng = Engine(params, 0, max_batch=8)
    eng.load_state_dict(sd)
    assert eng.set_fused_layernorm(fused) == fused
    eng.debug_keep(True)
    x = mg.encoder_inputs(F, T)
    xt = torch_cuda.as_tensor(x).cuda()
    emb = eng.encode(xt, norm=True).cpu().numpy()
    taps_ref = []
    ref = oe.encode(x, sd, params, norm=True, taps=taps_ref)
    ratios, worst = [], 0.0
    for i, tr in enumerate(taps_ref):
        tg = eng.debug_activation(i, x.shape[0])
        e = np.abs(tg - tr).max() / max(1.0, np.abs(tr).max())
        worst = max(worst, e)
    print("large-mean (%s): emb err %.3e, worst tap rel err %.3e" % ("fused" if fused else "unfused", np.abs(emb - ref).max(), worst))
    assert worst < 2e-4
    assert np.abs(emb - ref).max() < 1e-4
    eng.debug_keep(False)
    assert np.abs(eng.encode(xt, norm=True).cpu().numpy() - ref).max() < 1e-4


def test_encoder_large_batch_128_tiles_vs_oracle(torch_cuda):
    """A batch large enough for the 128x128 GEMM tiles (the golden tests run 3 segments, i.e. the 64x64 tiles):
    embeddings against the CPU oracle for a batch that is not a multiple of anything convenient."""
    from oracle import encoder as oe
    from pfann_amd.engine import Engine
    params = cfg("default")
    _, _, _, F, T = synth.model_dims(params)
    sd = synth.make_state_dict(params, seed=123)
    B = 37
    x = (synth.normal(78, "t/persist", B * F * T).reshape(B, F, T) * 3.0 - 6.0).astype(np.float32)
    ref = oe.encode(x, sd, params, norm=True)
    eng = Engine(params, 0, max_batch=64)
    eng.load_state_dict(sd)
    e1 = eng.encode(torch_cuda.as_tensor(x).cuda(), norm=True).cpu().numpy()
    print("128-tile kernels vs oracle %.3e" % np.abs(e1 - ref).max())
    assert np.abs(e1 - ref).max() < 1e-4


@pytest.mark.parametrize("variant", ["default", "elu_pre_ln"])
def test_encoder_five_block_kernel_vs_oracle(torch_cuda, variant):
    """A batch large enough (130 segments) for the stride-2 layers with 1024 / 512 / 256 rows per sample to run on
    conv_gemm_ln_w22_kernel (five channel blocks per output pair instead of six: conv along T and along F, N = 128 and
    256, a last tile with rows >= M): every sub-layer activation and the embeddings against the CPU oracle.  "elu_pre_ln":
    the same model with ELU applied BEFORE LayerNorm (model.py:58-72 with relu_after_bn=False), i.e. the kernel's generic
    <false, *> instantiations incl. the folded first conv with its pre-activation.  The sums are
    associated differently from the plain kernel's, so this is an fp32-rounding-level comparison, same bars as the
    golden tests (2e-4 relative on activations, 1e-4 on embeddings)."""
    from oracle import encoder as oe
    from pfann_amd.engine import Engine
    params = cfg("default")
    if variant == "elu_pre_ln":
        params["model"].update(conv_activation="ELU", relu_after_bn=False)
    _, _, _, F, T = synth.model_dims(params)
    sd = synth.make_state_dict(params, seed=123)
    B = 130
    x = (synth.normal(79, "t/w22", B * F * T).reshape(B, F, T) * 3.0 - 6.0).astype(np.float32)
    taps_ref = []
    ref = oe.encode(x, sd, params, norm=True, taps=taps_ref)
    eng = Engine(params, 0, max_batch=160)
    eng.load_state_dict(sd)
    eng.debug_keep(True)
    e1 = eng.encode(torch_cuda.as_tensor(x).cuda(), norm=True).cpu().numpy()
    worst = 0.0
    for i, tr in enumerate(taps_ref):
        tg = eng.debug_activation(i, 8)
        e = np.abs(tg - tr[:8]).max() / max(1.0, np.abs(tr[:8]).max())
        worst = max(worst, e)
        assert e < 2e-4, "sub-layer %d: rel err %g" % (i, e)
    print("five-block kernels vs oracle: emb %.3e, worst tap rel %.3e" % (np.abs(e1 - ref).max(), worst))
    assert np.abs(e1 - ref).max() < 1e-4
    eng.debug_keep(False)
    e2 = eng.encode(torch_cuda.as_tensor(x).cuda(), norm=True).cpu().numpy()
    assert np.abs(e2 - ref).max() < 1e-4
    # run to run bit-reproducible
    assert np.array_equal(e2, eng.encode(torch_cuda.as_tensor(x).cuda(), norm=True).cpu().numpy())


def test_encoder_mid_batch_plan_vs_oracle(torch_cuda):
    """Round 6: the middle of the batch curve.  At 304 windows the plan puts the five big stride-2
    layers on the 128-tile five-block kernel, the layers with 16 .. 128 rows per window on 64x64 tiles, and SPLITS K on the
    deep layers (1 .. 8 rows per window, K loops of 48-96 K-tiles: encoder_fused.hip splitk_plan) with the LayerNorm
    statistics taken by the split-K reduction.  Embeddings against the CPU oracle (1e-4), run-to-run bit-reproducible, and
    under pfann_set_plan_batch(304) the first 76 windows alone embed to the same bits as inside the 304-window launch
    (the K cut depends on the layer only; the reduction order is fixed)."""
    from oracle import encoder as oe
    from pfann_amd.engine import Engine
    params = cfg("default")
    _, _, _, F, T = synth.model_dims(params)
    sd = synth.make_state_dict(params, seed=123)
    B = 304
    x = (synth.normal(83, "t/mid", B * F * T).reshape(B, F, T) * 3.0 - 6.0).astype(np.float32)
    ref = np.concatenate([oe.encode(x[i:i + 76], sd, params, norm=True) for i in range(0, B, 76)])
    eng = Engine(params, 0, max_batch=B)
    eng.load_state_dict(sd)
    xt = torch_cuda.as_tensor(x).cuda()
    e1 = eng.encode(xt, norm=True).cpu().numpy()
    print("mid-batch plan (304 windows) vs oracle %.3e" % np.abs(e1 - ref).max())
    assert np.abs(e1 - ref).max() < 1e-4
    assert np.array_equal(e1, eng.encode(xt, norm=True).cpu().numpy())
    assert eng.set_plan_batch(B) == B
    e_all = eng.encode(xt, norm=True).cpu().numpy()
    e_head = eng.encode(xt[:76].contiguous(), norm=True).cpu().numpy()
    assert np.array_equal(e_all, e1)
    assert np.array_equal(e_head, e_all[:76]), "a window's bits depend on the batch under a pinned plan"
    eng.set_plan_batch(0)
    e76 = eng.encode(xt[:76].contiguous(), norm=True).cpu().numpy()          # its own plan (64x64 tiles everywhere)
    assert np.abs(e76 - ref[:76]).max() < 1e-4


def test_encoder_split_precision_matches_fp32(torch_cuda):
    """Opt-in encoder arithmetic (pfann_set_encoder_precision = 1): conv products as three fp16 MFMA
    terms of two-term operand splits, fp32 accumulation.  Must stay fp32-grade: embeddings within 2e-5
    of the exact fp32-MFMA path on a batch large enough that the 128x128 GEMM tiles (the only ones
    with a split instantiation) are used, and within the 1e-4 bar of the CPU oracle."""
    from oracle import encoder as oe
    from pfann_amd.engine import Engine
    params = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                         "configs", "default.json")))
    _, _, _, F, T = synth.model_dims(params)
    sd = synth.make_state_dict(params, seed=123)
    eng = Engine(params, 0, max_batch=512)
    eng.load_state_dict(sd)
    B = 512
    x = (synth.normal(77, "t/split", B * F * T).reshape(B, F, T) * 3.0 - 6.0).astype(np.float32)
    xt = torch_cuda.as_tensor(x).cuda()
    assert eng.set_encoder_precision(0) == 0
    e0 = eng.encode(xt, norm=True).cpu().numpy()
    r0 = eng.encode(xt, norm=False).cpu().numpy()
    assert eng.set_encoder_precision(1) == 1
    e1 = eng.encode(xt, norm=True).cpu().numpy()
    r1 = eng.encode(xt, norm=False).cpu().numpy()
    assert eng.set_encoder_precision(0) == 0
    d_emb, d_raw = np.abs(e1 - e0).max(), np.abs(r1 - r0).max() / max(1.0, np.abs(r0).max())
    print("split vs fp32: emb %.3e raw(rel) %.3e" % (d_emb, d_raw))
    assert np.isfinite(e1).all()
    assert d_emb < 2e-5 and d_raw < 2e-5
    ref = oe.encode(x[:6], sd, params, norm=True)
    assert np.abs(e1[:6] - ref).max() < 1e-4


def test_encoder_batch_independence_and_chunking(torch_cuda):
    """Batch 37 through max_batch=16 chunks equals per-sample results."""
    from pfann_amd.engine import Engine
    params = cfg("tiny")
    eng = Engine(params, 0, max_batch=16)
    eng.load_state_dict(synth.make_state_dict(params))
    x = (-17 + 19 * synth.uniform01(1, "t/batch", 37 * 256 * 32)).reshape(37, 256, 32).astype(np.float32)
    xt = torch_cuda.as_tensor(x).cuda()
    all_ = eng.encode(xt).cpu().numpy()
    one = np.concatenate([eng.encode(xt[i:i + 1]).cpu().numpy() for i in (0, 15, 16, 36)])
    assert np.array_equal(all_[[0, 15, 16, 36]], one)
    assert eng.encode(xt[:0]).shape == (0, 16)


def test_missing_weights_fail_loudly(torch_cuda):
    from pfann_amd import lib
    from pfann_amd.engine import Engine
    params = cfg("tiny")
    eng = Engine(params, 0)
    with pytest.raises(lib.PfannError):
        eng.encode(torch_cuda.zeros(1, 256, 32).cuda())
    sd = synth.make_state_dict(params)
    with pytest.raises(lib.PfannError):
        eng.load_state_dict({"f.convs.0.conv1.weight": sd["f.convs.0.conv1.weight"][:1]})
    with pytest.raises(lib.PfannError):
        eng.load_state_dict({"bogus.weight": np.zeros(3, np.float32)})


# -------------------------------------------------------------------------------- search
def _check_topk(torch, db, q, k, prefilter=True):
    from oracle import search as osr
    from pfann_amd.database import DeviceIndex
    d = q.shape[1]
    idx = DeviceIndex(d, 0)
    pos = np.array([0, db.shape[0]], np.int64)
    idx.load(db, pos, 0)
    idx.set_prefilter(prefilter)        # fp16 pre-filter + exact re-scoring vs all-fp32 scan: same answer
    D, I = idx.search(torch.as_tensor(q).cuda(), k)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    # (big cases: the argpartition form of the same definition -- a full stable argsort of 2100 x 150,000 scores took 20 s)
    Dr, Ir = (osr.flat_ip_topk_blas if q.shape[0] * db.shape[0] > (1 << 24) else osr.flat_ip_topk)(q, db, k)
    n = min(k, db.shape[0])
    assert (I[:, n:] == -1).all() and (I[:, :n] >= 0).all()
    assert (D[:, n:] == -np.finfo(np.float32).max).all()
    if n == 0:
        return D, I
    # scores descending and equal to the oracle's to fp32 rounding
    assert (np.diff(D[:, :n], axis=1) <= 0).all()
    assert np.abs(D[:, :n] - Dr[:, :n]).max() < 2e-6
    # identical label sets, except where the k-th score is tied to rounding
    for r in range(q.shape[0]):
        a, b = set(I[r, :n].tolist()), set(Ir[r, :n].tolist())
        if a != b:
            kth = Dr[r, n - 1]
            for lab in a ^ b:
                assert abs(float(db[lab] @ q[r]) - kth) < 2e-6, "row %d label %d not a k-th tie" % (r, lab)
    # reported scores belong to the reported labels
    chk = np.einsum("qkd,qd->qk", db[I[:, :n].clip(0)], q)
    assert np.abs(chk - D[:, :n]).max() < 2e-6
    return D, I


@pytest.mark.parametrize("n,d,nq,k", [
    (0, 128, 3, 10), (7, 128, 19, 100), (5000, 128, 19, 100), (8192, 64, 5, 1), (8193, 16, 33, 20),
    (100000, 128, 19, 100), (300000, 128, 130, 100), (140000, 128, 64, 300), (200000, 64, 40, 1000),
])
@pytest.mark.parametrize("prefilter", [True, False])
def test_search_topk_exact(torch_cuda, n, d, nq, k, prefilter):
    db = synth.unit_rows(11, "t/db%d" % n, max(n, 1), d)[:n]
    q = synth.unit_rows(12, "t/q%d" % n, nq, d)
    if n > 100:                      # plant near-duplicates of db rows so real matches exist
        q[::3] = db[(np.arange(0, nq, 3) * 7919) % n] * 0.8 + 0.2 * q[::3]
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    _check_topk(torch_cuda, db, q.astype(np.float32), k, prefilter)


def test_search_topk_clustered_and_duplicate_rows(torch_cuda):
    """Non-iid db: long runs of near-identical rows (one 'song') and exact duplicates; the
    sampled thresholds must stay valid lower bounds (never lose a true top-k row)."""
    d, n = 128, 120000
    base = synth.unit_rows(21, "t/cl", n, d)
    c = synth.unit_rows(22, "t/center", 1, d)
    base[40000:40600] = c + 0.05 * base[40000:40600]       # 600 rows around one centre
    base[70000:70300] = base[70000]                        # 300 exact duplicates
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    q = np.concatenate([c, base[70000:70001], synth.unit_rows(23, "t/q", 17, d)]).astype(np.float32)
    _check_topk(torch_cuda, base.astype(np.float32), q, 100)
    # batched (fp16 pre-filter) form of the same, both ways
    qb = np.concatenate([q, synth.unit_rows(24, "t/qb", 120, d)]).astype(np.float32)
    _check_topk(torch_cuda, base.astype(np.float32), qb, 100, True)
    _check_topk(torch_cuda, base.astype(np.float32), qb, 100, False)


@pytest.mark.parametrize("n,d,nq,k", [(40000, 128, 1100, 100), (70001, 64, 1030, 20), (150000, 128, 2100, 300)])
def test_search_topk_large_batch_sublists(torch_cuda, n, d, nq, k):
    """nq >= 1024 takes the query-stationary fp16 scan (survivors in per-slice sub-lists, gathered
    by the select kernel); same exact answer, including a ragged last query tile and db tile."""
    db = synth.unit_rows(41, "t/lb%d" % n, n, d)
    q = synth.unit_rows(42, "t/lbq%d" % n, nq, d)
    q[::7] = db[(np.arange(len(q[::7])) * 911) % n] + 0.2 * q[::7]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    _check_topk(torch_cuda, db.astype(np.float32), q.astype(np.float32), k, True)


@pytest.mark.parametrize("n,d,nq", [(250007, 128, 33), (250007, 128, 76), (250007, 128, 304), (250007, 128, 1000), (130001, 64, 200)])
def test_search_topk_mid_batch_takes_the_query_stationary_kernels(torch_cuda, n, d, nq):
    """Round 6: 33 .. 1023 query rows (one to eight query tiles) run the sampled group-maximum pass + the full pass with
    sub-lists too -- up to 64 db slices per query tile, three tile buffers where the launch leaves most of the chip empty
    (csrc/search_f16.hip: qres_min_nq, NBUF) -- instead of the survivor ladder on the generic kernel.  Same exact answer on
    a db of 'songs' (runs of 40 similar rows), ragged last db tile and query tile."""
    db = synth.unit_rows(51, "t/mb%d" % n, n, d)
    heads = np.repeat(db[::40], 40, axis=0)[:n]
    db = heads + 0.6 * db
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = synth.unit_rows(52, "t/mbq%d" % n, nq, d)
    q[::2] = db[(np.arange(len(q[::2])) * 7919) % n] + 0.5 * q[::2]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    _check_topk(torch_cuda, db.astype(np.float32), q.astype(np.float32), 100, True)


@pytest.mark.parametrize("nq", [130, 2100])
def test_search_topk_private_lists_take_whole_songs(torch_cuda, nq):
    """Round 6: the query-stationary scan keeps four PRIVATE survivor lists per (query row, db slice) -- one per owner lane --
    instead of one list behind an LDS counter, and hands groups of four consecutive db rows to the four owners in turn.  A db
    made of runs of 40 nearly identical rows whose query sits in the middle of a run puts 40 survivors of one query row into one
    64-row tile: they must spread over the private lists (no list overflow -> no exact fallback -> the fp32 scores come from
    the select's own re-scoring) and the answer is the exact top-k, for one query tile (64 slices, lists of 32) and for
    seventeen."""
    d, n, k = 128, 150000, 100
    db = synth.unit_rows(61, "t/pl", n, d)
    for s0 in range(0, n, 40):
        db[s0:s0 + 40] = db[s0] + 0.05 * db[s0:s0 + 40]
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = synth.unit_rows(62, "t/plq", nq, d)
    q[::2] = db[(np.arange(len(q[::2])) * 7919 + 20) % n] + 0.3 * q[::2]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    _check_topk(torch_cuda, db.astype(np.float32), q.astype(np.float32), k, True)


def test_search_topk_sublist_overflow_falls_back(torch_cuda):
    """More near-identical rows inside ONE interleaved db slice than a sub-list holds (256 at 32
    slices): the rows that lost survivors are recomputed exactly by the device-side fallback kernel."""
    d, n, nq = 128, 60000, 1100
    db = synth.unit_rows(43, "t/of", n, d)
    c = synth.unit_rows(44, "t/ofc", 1, d)
    for u in range(3):                                   # db tiles 5, 37, 69 all belong to slice 5 of 32
        lo = (5 + 32 * u) * 128
        db[lo:lo + 128] = c + 0.01 * db[lo:lo + 128]
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = synth.unit_rows(45, "t/ofq", nq, d)
    q[:3] = c + 0.05 * q[:3]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    _check_topk(torch_cuda, db.astype(np.float32), q.astype(np.float32), 300, True)


@pytest.mark.parametrize("nq", [80, 1100])
def test_search_topk_more_than_4096_survivors(torch_cuda, nq):
    """5000 near-identical rows all clear the sampled threshold of the queries that match them: those
    rows' survivor lists (> 4096, < 8192 entries) are left to the large select kernel, the others go
    through the 256-thread one; both in the same call (nq = 1100 additionally overflows the per-slice
    sub-lists and takes the one-list-per-row rescan)."""
    d, n = 128, 100000
    db = synth.unit_rows(51, "t/big", n, d)
    c = synth.unit_rows(52, "t/bigc", 1, d)
    db[20000:25000] = c + 0.003 * db[20000:25000]
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = synth.unit_rows(53, "t/bigq", nq, d)
    q[:5] = c + 0.05 * q[:5]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    _check_topk(torch_cuda, db.astype(np.float32), q.astype(np.float32), 100, True)


@pytest.mark.parametrize("case", range(8))
def test_search_topk_random_shapes(torch_cuda, case):
    """Seeded random (n, d, nq, k) across the regimes (single-query kernel, generic batched kernel,
    query-stationary sub-list kernel, ragged tiles, k up to 300) against the oracle."""
    u = synth.uniform01(900 + case, "t/shape", 8)
    d = 128 if u[0] < 0.6 else 64
    n = int(2000 + u[1] ** 2 * 180000)
    nq = [1, 19, 33, 70, 300, 1030, 1500, 2300][case]
    k = int(1 + u[2] ** 2 * 299)
    db = synth.unit_rows(910 + case, "t/rs", n, d)
    q = synth.unit_rows(920 + case, "t/rsq", nq, d)
    q[::3] = db[(np.arange(len(q[::3])) * 7919 + case) % n] + 0.3 * q[::3]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    _check_topk(torch_cuda, db.astype(np.float32), q.astype(np.float32), k, True)


def test_search_prefilter_near_ties_stay_exact(torch_cuda):
    """Adversarial for the fp16 pre-filter: thousands of rows whose exact scores differ by ~1e-6
    (far below fp16 resolution, 1e-3) around the k-th best.  The re-scoring window (2 eps below
    the k-th best approximate score) must hand the exact fp32 order to the final sort."""
    d, n, nq = 128, 60000, 96
    base = synth.unit_rows(31, "t/tie", n, d).astype(np.float64)
    c = synth.unit_rows(32, "t/tiec", 1, d).astype(np.float64)[0]
    for lo, cnt, amp in ((1000, 3000, 0.02), (20000, 500, 0.002)):
        base[lo:lo + cnt] = c + amp * base[lo:lo + cnt]
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    q = synth.unit_rows(33, "t/tieq", nq, d).astype(np.float64)
    q[:40] = c + 0.05 * q[:40]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    D, I = _check_topk(torch_cuda, base.astype(np.float32), q.astype(np.float32), 100, True)
    # the interesting rows really are inside the fp16 ambiguity: k-th and (k+50)-th exact scores
    s = np.sort((q[:1].astype(np.float32) @ base.astype(np.float32).T)[0])[::-1]
    assert s[99] - s[149] < 1e-3



# ------------------------------------------------------------ round-2 search additions
def test_search_small_batch_sublist_overflow_device_fallback(torch_cuda):
    """Small-batch path (nq <= 32): 900 near-identical rows exactly 32 tiles apart land in one or two of a query
    row's 32 sub-lists (256 slots each), and 9000 rows tying EXACTLY at the top overflow every list of their query
    row.  Both rows are recomputed exactly by the device-side fallback kernel (no host retry, no error); exact ties
    resolve to ascending row ids like the oracle's stable sort."""
    d, n = 64, 1000000
    rng = np.random.default_rng(61)
    db = rng.standard_normal((n, d)).astype(np.float32)
    c = rng.standard_normal((2, d)).astype(np.float32)
    db[50000:59000] = c[0]                                   # 9000 identical rows: all tie at score 1
    for u in range(900):                                     # rows 32 tiles apart -> same sub-list (tile & 31)
        lo = 70001 + u * 32 * 32
        db[lo] = c[1] + 0.01 * db[lo]
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = np.concatenate([c, rng.standard_normal((17, d)).astype(np.float32)])
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    D, I = _check_topk(torch_cuda, db, q.astype(np.float32), 300)
    assert np.array_equal(I[0, :300], np.arange(50000, 50300))          # exact ties -> lowest rows first
    # the same handle semantics again: the per-row flags were cleared by the fallback kernel
    _check_topk(torch_cuda, db, q[2:].astype(np.float32), 100)


@pytest.mark.parametrize("prefilter", [True, False])
def test_search_small_batch_d128_tail_launch(torch_cuda, prefilter):
    """d = 128, nq <= 32: the five-launch small-batch path (query preparation inside the group-maximum pass, select,
    then ONE launch for the big select and the exact fallback).  Query row 0: 5000 near-identical rows (more than 4096
    survivors: the big-select half of the tail launch); row 1: 9000 rows tying exactly at the top (every sub-list
    overflows: the fallback half, exact ties -> ascending row ids); row 2: 900 near-identical rows 32 tiles apart (one
    sub-list overflows); the rest ordinary rows through the 256-thread select."""
    d, n = 128, 600000
    rng = np.random.default_rng(71)
    db = rng.standard_normal((n, d)).astype(np.float32)
    c = rng.standard_normal((3, d)).astype(np.float32)
    db[20000:25000] = c[0] + 0.003 * db[20000:25000]
    db[50000:59000] = c[1]
    for u in range(900):
        lo = 70001 + u * 32 * 32
        if lo < n:
            db[lo] = c[2] + 0.01 * db[lo]
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = np.concatenate([c, rng.standard_normal((16, d)).astype(np.float32)])
    q[0] = c[0] + 0.05 * q[3]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    D, I = _check_topk(torch_cuda, db, q.astype(np.float32), 300, prefilter)
    assert np.array_equal(I[1, :300], np.arange(50000, 50300))
    # the flags the tail launch cleared: the same handle again, ordinary rows only
    _check_topk(torch_cuda, db, q[3:].astype(np.float32), 100, prefilter)


def test_search_small_batch_fp16_storage_fallback_d128(torch_cuda):
    """fp16-only storage, d = 128, nq <= 32: 9000 rows tying exactly at the top overflow every sub-list of query row 0; the
    fallback half of the combined tail launch streams the fp16 rows (its ELT = 2 form) and returns the lowest row ids."""
    from oracle import search as osr
    from pfann_amd.database import DeviceIndex
    d, n, k = 128, 300000, 300
    rng = np.random.default_rng(91)
    db = rng.standard_normal((n, d)).astype(np.float32)
    c = rng.standard_normal((1, d)).astype(np.float32)
    db[50000:59000] = c[0]
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = np.concatenate([c, rng.standard_normal((6, d)).astype(np.float32)])
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    idx = DeviceIndex(d, 0, storage="f16")
    idx.load(db, np.array([0, n], np.int64), 0)
    D, I = idx.search(torch_cuda.as_tensor(q).cuda(), k)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    Dr, Ir = osr.flat_ip_topk_f16(q, db, k)
    assert np.array_equal(I[0], np.arange(50000, 50000 + k))
    assert np.abs(D - Dr).max() < 2e-5
    for r in range(1, q.shape[0]):                       # ordinary rows: same label sets up to ties at the k-th score
        assert len(set(I[r]) ^ set(Ir[r])) <= 4, r


def test_search_all_scores_tie_zero_query(torch_cuda):
    """q = 0: every row scores 0, the survivor lists overflow at any threshold; exact answer = rows 0..k-1."""
    d, n = 128, 30000
    db = synth.unit_rows(64, "t/zq", n, d).astype(np.float32)
    for nq in (3, 100):
        q = np.zeros((nq, d), np.float32)
        q[-1] = db[77]
        D, I = _check_topk(torch_cuda, db, q, 50)
        assert np.array_equal(I[0], np.arange(50)) and I[-1, 0] == 77


def test_search_prefilter_margin_scales_with_norms(torch_cuda):
    """The fp16 pre-filter's margin must hold for non-unit rows too: db rows of norm ~40 and queries of norm
    ~25 with thousands of exact scores 1e-3 relative apart around the k-th best; and a query row far outside
    fp16's range (norm 3e5) is answered exactly through the fallback instead of through inf/NaN halves."""
    d, n, nq = 128, 60000, 96
    base = synth.unit_rows(71, "t/ms", n, d).astype(np.float64)
    c = synth.unit_rows(72, "t/msc", 1, d).astype(np.float64)[0]
    base[1000:4000] = c + 0.02 * base[1000:4000]
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    base *= (20.0 + 40.0 * synth.uniform01(73, "t/msn", n).astype(np.float64))[:, None]
    q = synth.unit_rows(74, "t/msq", nq, d).astype(np.float64)
    q[:40] = c + 0.05 * q[:40]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q *= 25.0
    q[50] *= 12000.0
    from oracle import search as osr
    from pfann_amd.database import DeviceIndex
    db32, q32 = base.astype(np.float32), q.astype(np.float32)
    idx = DeviceIndex(d, 0)
    idx.load(db32, np.array([0, n], np.int64), 0)
    assert idx.set_prefilter(True)
    D, I = idx.search(torch_cuda.as_tensor(q32).cuda(), 100)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    s = q.astype(np.float64) @ base.T
    for r in range(nq):
        kth = np.sort(s[r])[::-1][99]
        tol = 4e-7 * np.linalg.norm(q[r]) * 60.0               # fp32 rounding of scores of this magnitude
        got = s[r, I[r]]
        assert (got >= kth - tol).all(), "row %d returned a row below the k-th best" % r
        assert np.abs(D[r] - got).max() < 4 * tol


@pytest.mark.parametrize("n,d,nq,k", [(7, 128, 19, 100), (5000, 128, 19, 100), (100000, 128, 19, 100),
                                      (90000, 64, 1, 20), (150000, 128, 130, 100), (70001, 64, 1100, 100),
                                      (40000, 128, 40, 300)])
def test_search_fp16_storage(torch_cuda, n, d, nq, k):
    """fp16-only storage (pfann_db_set_storage(db, PFANN_DB_F16)): the result is the top-k by
    s16 = sum fl16(q_i) fl16(x_i) -- against the oracle's float64 evaluation of exactly that (scores to fp32
    summation rounding; label sets identical except ties at the k-th score) on every regime (small-batch
    streaming kernel, generic and query-stationary batched kernels)."""
    from oracle import search as osr
    from pfann_amd.database import DeviceIndex
    db = synth.unit_rows(81, "t/h%d" % n, n, d).astype(np.float32)
    q = synth.unit_rows(82, "t/hq%d" % n, nq, d)
    q[::3] = db[(np.arange(len(q[::3])) * 7919) % n] * 0.8 + 0.2 * q[::3]
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    idx = DeviceIndex(d, 0, storage="f16")
    idx.load(db, np.array([0, n], np.int64), 0)
    assert idx.lib.pfann_db_bytes(idx.handle) == n * d * 2
    D, I = idx.search(torch_cuda.as_tensor(q).cuda(), k)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    Dr, Ir = osr.flat_ip_topk_f16(q, db, k)
    kk = min(k, n)
    assert (I[:, kk:] == -1).all() and (I[:, :kk] >= 0).all()
    assert (np.diff(D[:, :kk], axis=1) <= 0).all()
    assert np.abs(D[:, :kk] - Dr[:, :kk]).max() < 2e-6
    x16 = db.astype(np.float16).astype(np.float64)
    q16 = q.astype(np.float16).astype(np.float64)
    for r in range(nq):
        a, b = set(I[r, :kk].tolist()), set(Ir[r, :kk].tolist())
        for lab in a ^ b:
            assert abs(float(x16[lab] @ q16[r]) - Dr[r, kk - 1]) < 2e-6, "row %d label %d not a k-th tie" % (r, lab)


def test_match_on_fp16_storage_uses_stored_rows(torch_cuda):
    """Sequence matcher over fp16-only storage: scores are dot products with the STORED (fp16-rounded) rows."""
    from oracle import seqscore as osq
    from pfann_amd.database import DeviceIndex
    z = np.load(os.path.join(G, "database.npz"))
    name = "random_noisy"
    db, q, labels = z[name + "_db"], z[name + "_q"], z[name + "_labels"]
    pos = osq.song_pos_from_key(z[name + "_key"])
    idx = DeviceIndex(db.shape[1], 0, storage="f16")
    idx.load(db, pos, 0)
    res, ss = idx.match(torch_cuda.as_tensor(q).cuda(), torch_cuda.as_tensor(labels).cuda(), [0], [q.shape[0]],
                        want_song_scores=True)
    db16 = db.astype(np.float16).astype(np.float32)
    score, (song, sec), ss_ref = osq.query_embeddings_base(q, labels, db16, pos, 0.5, 1)
    assert int(res[0]["song"]) == song and int(res[0]["offset"]) * 0.5 == sec
    assert abs(float(res[0]["score"]) - score) < 1e-6


def test_topk_merge(torch_cuda):
    from pfann_amd.database import DeviceIndex
    idx = DeviceIndex(128, 0)
    rng = np.random.default_rng(3)
    S = rng.standard_normal((9, 800)).astype(np.float32)
    L = rng.permutation(10 ** 6)[:9 * 800].reshape(9, 800).astype(np.int64) + (1 << 33)
    L[0, :700] = -1
    D, I = idx.merge_topk(torch_cuda.as_tensor(S).cuda(), torch_cuda.as_tensor(L).cuda(), 100)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    for r in range(9):
        ok = L[r] >= 0
        o = np.argsort(-S[r][ok], kind="stable")[:100]
        assert np.array_equal(I[r, :len(o)], L[r][ok][o])
        assert np.array_equal(D[r, :len(o)], S[r][ok][o])


# --------------------------------------------------------------------- sequence matcher
DB_CASES = ["clean_hit", "negative_offset", "past_end", "k_gt_ntotal", "duplicate_songs",
            "nonpositive_best", "no_candidates", "frame_shift_mul2", "random_noisy"]


def _index_for(z, name):
    from oracle import seqscore as osq
    from pfann_amd.database import DeviceIndex
    db = z[name + "_db"]
    pos = osq.song_pos_from_key(z[name + "_key"])
    idx = DeviceIndex(db.shape[1], 0)
    idx.load(db, pos, 0)
    return idx, db, pos


@pytest.mark.parametrize("name", DB_CASES)
def test_match_python_path_vs_reference_golden(torch_cuda, name):
    """a8-a9, decisions exact: (song, time) identical to the reference's
    query_embeddings_base on the same (query, labels); score to fp32 rounding."""
    from pfann_amd.database import _fine_to_time
    z = np.load(os.path.join(G, "database.npz"))
    idx, db, pos = _index_for(z, name)
    q, labels, fsm = z[name + "_q"], z[name + "_labels"], int(z[name + "_fsm"])
    res, ss = idx.match(torch_cuda.as_tensor(q).cuda(), torch_cuda.as_tensor(labels).cuda(), [0], [q.shape[0]],
                        fsm, 0.0, 0, False, True)
    r = res[0]
    want_song, want_sec, want_score = int(z[name + "_song"]), float(z[name + "_sec"]), float(z[name + "_score"])
    assert int(r["song"]) == want_song
    if want_song >= 0:
        assert (int(r["offset"]) - int(r["shift"]) / fsm) * 0.5 == want_sec
        assert abs(float(r["score"]) - want_score) < 1e-6
    else:
        assert float(r["score"]) == -np.inf
    got_ss = ss.cpu().numpy()[0]
    got_ss[:, 1] = _fine_to_time(got_ss[:, 1].astype(np.int64), fsm, 0.5)
    assert np.allclose(got_ss, z[name + "_song_score"], atol=1e-6)
    assert np.array_equal(got_ss[:, 1], z[name + "_song_score"][:, 1])


@pytest.mark.parametrize("name", DB_CASES)
def test_seq_score_c_abi_vs_oracle(torch_cuda, name):
    """The reference's ctypes seam (database.py:15-32,178-189) against the C restatement of
    cpp/seqscore.cpp, incl. score_alpha > 0."""
    from oracle import native
    from pfann_amd import lib as L
    z = np.load(os.path.join(G, "database.npz"))
    idx, db, pos = _index_for(z, name)
    q = np.ascontiguousarray(z[name + "_q"], np.float32)
    labels = np.ascontiguousarray(z[name + "_labels"], np.int64)
    fsm = int(z[name + "_fsm"])
    lib = L.load()
    assert lib.version() == 20220625002
    for alpha in (0.0, 3.0):
        ss = np.zeros((pos.shape[0] - 1, 2), np.float32)
        best = lib.seq_score(idx.handle, pos.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), pos.shape[0] - 1,
                             q.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), q.shape[0],
                             labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), labels.shape[1],
                             ss.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), fsm, alpha)
        wbest, wss = native.seq_score(db, pos, q, labels, fsm, alpha)
        assert best == wbest, (name, alpha, L.last_error())
        assert np.array_equal(ss[:, 1], wss[:, 1])
        assert np.allclose(ss[:, 0], wss[:, 0], atol=2e-6)


@pytest.mark.parametrize("nq_total", [40, 7])
def test_match_batched_random_vs_oracle(torch_cuda, nq_total):
    """Many queries in one launch, real search output as labels, vs the python-path oracle
    (7 queries: the phased launch that spreads each query's candidates over the whole GPU)."""
    from oracle import seqscore as osq
    from pfann_amd.database import DeviceIndex
    d = 128
    key = [int(x) for x in (30 + 60 * synth.uniform01(31, "t/key", 300))]
    key[7] = 0
    db = synth.unit_rows(32, "t/mdb", sum(key), d)
    pos = osq.song_pos_from_key(key)
    idx = DeviceIndex(d, 0)
    idx.load(db, pos, 0)
    qs, qstart, qlen = [], [], []
    n = 0
    for j in range(nq_total):
        s = (j * 37) % 300
        if key[s] < 25:
            s += 1
        ln = 5 + (j % 15)
        off = (j * 13) % (key[s] - ln)
        qq = db[pos[s] + off: pos[s] + off + ln] + (0.3 + 0.03 * j) * synth.unit_rows(100 + j, "t/qn", ln, d)
        qs.append(qq / np.linalg.norm(qq, axis=1, keepdims=True))
        qstart.append(n)
        qlen.append(ln)
        n += ln
    q = np.concatenate(qs).astype(np.float32)
    qt = torch_cuda.as_tensor(q).cuda()
    D, I = idx.search(qt, 100)
    res, _ = idx.match(qt, I, qstart, qlen, 1, 0.0, 0, False, False)
    In = I.cpu().numpy()
    for j in range(nq_total):
        sl = slice(qstart[j], qstart[j] + qlen[j])
        score, (song, sec), _ = osq.query_embeddings_base(q[sl], In[sl], db, pos, 0.5, 1)
        assert int(res[j]["song"]) == song and int(res[j]["offset"]) * 0.5 == sec, j
        assert abs(float(res[j]["score"]) - score) < 1e-6


def test_match_long_query_uses_hbm_scratch(torch_cuda):
    """A 60 s query (119 rows x top-100 = 11900 candidates) exceeds the LDS candidate buffer and
    goes through the HBM scratch path; decisions still equal the python-path oracle.  Mixed with a
    short query in the same launch."""
    from oracle import seqscore as osq
    from pfann_amd.database import DeviceIndex
    d = 64
    key = [150 + 7 * (i % 9) for i in range(120)]
    db = synth.unit_rows(61, "t/longdb", sum(key), d)
    pos = osq.song_pos_from_key(key)
    idx = DeviceIndex(d, 0)
    idx.load(db, pos, 0)
    ql = [119, 9, 119]
    starts = [(33, 10), (70, 3), (5, 20)]
    qs = []
    for j, (s, off) in enumerate(starts):
        qq = db[pos[s] + off: pos[s] + off + ql[j]] + 0.5 * synth.unit_rows(200 + j, "t/longq", ql[j], d)
        qs.append(qq / np.linalg.norm(qq, axis=1, keepdims=True))
    q = np.concatenate(qs).astype(np.float32)
    qstart = [0, 119, 128]
    qt = torch_cuda.as_tensor(q).cuda()
    D, I = idx.search(qt, 100)
    res, ss = idx.match(qt, I, qstart, ql, 1, 0.0, 0, False, True)
    In = I.cpu().numpy()
    for j in range(3):
        sl = slice(qstart[j], qstart[j] + ql[j])
        score, (song, sec), ss_ref = osq.query_embeddings_base(q[sl], In[sl], db, pos, 1.0, 1)
        assert int(res[j]["song"]) == song == starts[j][0] and int(res[j]["offset"]) == int(sec) == starts[j][1]
        assert abs(float(res[j]["score"]) - score) < 1e-6
        got = ss.cpu().numpy()[j]
        assert np.allclose(got, ss_ref, atol=1e-6) and np.array_equal(got[:, 1], ss_ref[:, 1])


def test_search_chunks_large_query_batches(torch_cuda):
    """pfann_search_topk walks batches above 16384 rows in chunks (bounded survivor workspace)."""
    from oracle import search as osr
    from pfann_amd.database import DeviceIndex
    d, n, nq = 16, 3000, 16384 + 700
    db = synth.unit_rows(71, "t/cdb", n, d)
    q = synth.unit_rows(72, "t/cq", nq, d)
    idx = DeviceIndex(d, 0)
    idx.load(db, np.array([0, n], np.int64), 0)
    D, I = idx.search(torch_cuda.as_tensor(q).cuda(), 5)
    I = I.cpu().numpy()
    rows = [0, 1, 16383, 16384, 16385, nq - 1]
    Dr, Ir = osr.flat_ip_topk(q[rows], db, 5)
    assert np.array_equal(I[rows], Ir)


def test_winner_keys_pack_and_pick_vs_numpy_statement(torch_cuda):
    """pfann_match_pack / pfann_match_pick (device-side multi-GPU winner selection) against the numpy statement of the
    same 128-bit key in tests/oracle_backend.py: bit-identical keys, identical winners incl. ties on the score
    (-> smallest (shift, song, offset)), negative scores, negative offsets and ranks without a candidate."""
    from oracle_backend import OracleIndex
    from pfann_amd.database import DeviceIndex
    idx = DeviceIndex(16, 0)
    ob = OracleIndex(16)
    rng = np.random.default_rng(5)
    G, nQ = 5, 300
    res = np.zeros((G, nQ), dtype=DeviceIndex.RESULT_DTYPE)
    res["song"] = rng.integers(-1, 200000, (G, nQ))
    res["offset"] = rng.integers(-40, 5000, (G, nQ))
    res["shift"] = rng.integers(0, 4, (G, nQ))
    res["score"] = rng.standard_normal((G, nQ)) * 0.3
    res["score"][:, :60] = np.round(res["score"][:, :60], 1)           # many exact score ties across ranks
    res["score"][1, 100] = -0.0
    res["score"][2, 100] = 0.0
    res["song"][:, 7] = -1                                               # nobody has a candidate
    keys = []
    for g in range(G):
        dev = torch_cuda.as_tensor(np.frombuffer(res[g].tobytes(), np.uint8).reshape(nQ, 24).copy()).cuda()
        kg = idx.pack_winner_keys(dev).cpu()
        assert np.array_equal(kg.numpy(), ob.pack_winner_keys(res[g]).numpy()), "key bits differ on rank %d" % g
        keys.append(kg)
    allk = torch_cuda.stack(keys)
    got, want = idx.pick_winner(allk.cuda()), ob.pick_winner(allk)
    for f in ("song", "offset", "shift"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["score"], want["score"]) and got["song"][7] == -1 and got["score"][7] == -np.inf
    # a query REFUSED on one rank (song == -2: candidate buffer sizing error) is not "no candidate": the pick hands the
    # -2 on and the host raises, whatever the other ranks found
    from pfann_amd.lib import PfannError
    res2 = res.copy()
    res2["song"][3, 11] = -2
    keys2 = [idx.pack_winner_keys(torch_cuda.as_tensor(np.frombuffer(res2[g].tobytes(), np.uint8).reshape(nQ, 24).copy()).cuda()).cpu()
             for g in range(G)]
    assert np.array_equal(keys2[3].numpy(), ob.pack_winner_keys(res2[3]).numpy())
    assert ob.pick_winner(torch_cuda.stack(keys2))["song"][11] == -2
    with pytest.raises(PfannError):
        idx.pick_winner(torch_cuda.stack(keys2).cuda())
    # and against the definition: highest score, ties -> smallest (shift, song, offset)
    for j in range(nQ):
        c = [(-(res["score"][g, j] + 0.0), res["shift"][g, j], res["song"][g, j], res["offset"][g, j]) for g in range(G) if res["song"][g, j] >= 0]
        if c:
            b = min(c)
            assert (got["shift"][j], got["song"][j], got["offset"][j]) == b[1:], j


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_two_phase_sharded_search_is_the_exact_global_topk(torch_cuda, storage):
    """pfann_search_bound on every shard -> MAX over the shards -> pfann_search_topk_bounded -> pfann_topk_merge equals
    the search over the unsplit db (labels and scores), each shard's bound is a true lower bound of its k-th best, and no
    shard emits a row below the reduced bound."""
    torch = torch_cuda
    from pfann_amd.database import DeviceIndex
    d, n, nq, k = 128, 150000, 2100, 100
    db = synth.unit_rows(71, "t/2p", n, d)
    # songs: runs of 40 similar consecutive rows, so that true matches cluster inside one shard
    for s0 in range(0, n, 40):
        db[s0:s0 + 40] = db[s0] + 0.6 * db[s0:s0 + 40]
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = synth.unit_rows(72, "t/2pq", nq, d)
    q[::2] = db[(np.arange(len(q[::2])) * 7919) % n] + 0.5 * q[::2]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    db_t = torch.from_numpy(db.astype(np.float32)).cuda()
    q_t = torch.from_numpy(q.astype(np.float32)).cuda()
    cuts = [0, 50000, 90000, n]                       # three uneven shards
    pos_all = np.array([0, n], np.int64)
    whole = DeviceIndex(d, 0, storage=storage)
    whole.load(db_t, pos_all, 0)
    D0, I0 = whole.search(q_t, k)
    shards = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ix = DeviceIndex(d, 0, storage=storage)
        ix.load(db_t[lo:hi].contiguous(), np.array([lo, hi], np.int64), lo)
        shards.append(ix)
    for ix in shards:                                             # m = k values per row: a shard's own k-th best is bounded
        c = ix.search_bound(q_t, k, k)
        assert c.shape == (nq, k) and bool(torch.isfinite(c).all())         # this shape takes the sampled-threshold path
        own = ix.reduce_bound(c[None], k)
        Dl, _ = ix.search(q_t, k)                                 # complete local list (also clears the pending state)
        assert bool((own <= Dl[:, k - 1] + 1e-6).all()), "a shard's bound exceeds its true k-th best"
    m = min(k, 2 * k // len(shards) + 8)                          # what dist.py exchanges
    cands = torch.stack([ix.search_bound(q_t, k, m) for ix in shards])       # one pending bound per handle
    L = whole.reduce_bound(cands, k)
    assert bool((L <= D0[:, k - 1] + 1e-6).all()), "the reduced bound exceeds the true global k-th best"
    # ... and it is tight: within the sampling loss of the global k-th best for most rows
    assert float((D0[:, k - 1] - L).median()) < 0.08
    parts = [ix.search_bounded(q_t, k, L) for ix in shards]
    S = torch.cat([p[0] for p in parts], 1)
    Lb = torch.cat([p[1] for p in parts], 1)
    Dm, Im = whole.merge_topk(S, Lb, k)
    assert torch.equal(Im, I0)
    assert float((Dm - D0).abs().max()) == 0.0
    for Dp, Ip in parts:                                          # nothing below the reduced bound (minus the margin) is emitted
        ok = Ip >= 0
        assert bool((Dp[ok] >= (L[:, None].expand_as(Dp))[ok] - 3e-3).all())
    # a bounded call without a pending bound is a plain search
    D1, I1 = shards[0].search_bounded(q_t, k, torch.full((nq,), float("-inf"), device="cuda"))
    D2, I2 = shards[0].search(q_t, k)
    assert torch.equal(I1, I2) and torch.equal(D1, D2)
    # a PENDING bound that bounds nothing (-inf): the second phase runs with the shard's own threshold, i.e. with hundreds of
    # survivors per row -- more than the wave-per-row select tier takes (round 5): every row goes through the list it leaves
    # to the workgroup selects, and the answer is still the plain search's, score bits included
    shards[0].search_bound(q_t, k, m)
    D3, I3 = shards[0].search_bounded(q_t, k, torch.full((nq,), float("-inf"), device="cuda"))
    assert torch.equal(I3, I2) and torch.equal(D3, D2)
    # ... and a tight one (the shard's own exact k-th best): a few dozen survivors per row, the wave tier's own case
    shards[0].search_bound(q_t, k, m)
    D4, I4 = shards[0].search_bounded(q_t, k, D2[:, k - 1].contiguous())
    assert torch.equal(torch.sort(I4, 1).values, torch.sort(I2, 1).values) and torch.equal(D4, D2)


@pytest.mark.parametrize("file_sr,seconds,n_ch", [(44100, 125.3, 1), (44100, 61.0, 2), (16000, 7.3, 2), (11025, 0.4, 1),
                                                  (48000, 60.0, 1), (22050, 119.02, 1)])
def test_resample_to_mono_vs_oracle(torch_cuda, file_sr, seconds, n_ch):
    """a1 at a non-native rate (musicdata.py:28-65): the device resampler + mono conversion against oracle/resample.py --
    lengths, the minute-wise seams (one and two full pieces, a tail of exactly one second, a file shorter than the filter)
    and the fake-stereo rule after resampling.  fp32 sums of up to 721 products in different orders: 2e-6."""
    from oracle import segmenter
    from pfann_amd.engine import Engine
    params = cfg("tiny")
    eng = Engine(params, 0)
    n = int(file_sr * seconds)
    rng = np.random.RandomState(file_sr % 1000 + n_ch)
    t = np.arange(n) / file_sr
    x = 0.4 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t + 1.0) + 0.1 * rng.randn(n)
    pcm = np.clip(np.round(x * 32767), -32768, 32767).astype(np.int16)[:, None]
    if n_ch == 2:
        pcm = np.concatenate([pcm, -pcm if file_sr == 16000 else (pcm // 2)], 1)      # 16 kHz case: opposite-phase stereo
    # the filtering arithmetic, to fp32 rounding: the oracle filters with the very table the kernel was given ...
    from pfann_amd import resample as presample
    want = segmenter.pcm_to_mono(pcm, file_sr, 8000, resample_table=presample.filter_table(file_sr, 8000)[0])
    got = eng.pcm16_to_mono(pcm, sample_rate=file_sr).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-6
    # ... and the whole path against the oracle's OWN table (float64 from the definition): the product builds its table in
    # fp32 op by op like julius, tests/test_host.py bounds the per-tap gap by 2e-5
    assert np.abs(got - segmenter.pcm_to_mono(pcm, file_sr, 8000)).max() < 2e-5
    if n_ch == 2 and file_sr == 16000:
        assert np.abs(got).max() > 0.3                                           # the flipped channel did not cancel
    assert np.array_equal(eng.pcm16_to_mono(pcm, sample_rate=8000).cpu().numpy(), segmenter.pcm_to_mono(pcm))


@pytest.mark.parametrize("cfgname", ["default", "seg", "tiny"])
def test_plan_batch_makes_fingerprints_independent_of_the_batch(cfgname):
    """pfann_set_plan_batch (include/pfann_amd.h; ADVICE r3: "one segment embeds to different bits depending on batch
    size"): by default the GEMM tile size, the split-K path and the small-batch head are chosen per call, so the last
    bits of a fingerprint depend on the batch it is computed in -- bounded here by 2e-5 between B <= 64 and B > 64; with
    the plan pinned (what the drop-in tools do) a segment has the SAME bits alone, in a small batch, at any position of
    a large one, and in a batch that is chunked by max_batch."""
    import torch
    from pfann_amd.engine import Engine
    params = cfg(cfgname)
    eng = Engine(params, 0, max_batch=256)
    eng.load_state_dict(synth.make_state_dict(params, seed=5))
    dev = eng.device
    pcm = synth.make_songs_torch(list(range(12)), 30.0, device=dev)
    wav = eng.pcm16_to_mono(pcm.reshape(-1))
    L = pcm.shape[1]
    starts = (torch.arange(12, device=dev)[:, None] * L + torch.arange(59, device=dev)[None, :] * 4000).reshape(-1)[:600]
    # ---- default: per-call variants, different bits, same fingerprints to rounding
    full = eng.embed_windows(wav, starts[:256])
    small = eng.embed_windows(wav, starts[:19])
    assert float((full[:19] - small).abs().max()) < 2e-5
    # ---- plan pinned: bit-identical however the windows are batched
    assert eng.set_plan_batch(256) == 256
    full = eng.embed_windows(wav, starts)                          # 600 windows = chunks of 256, 256, 88
    bits = full.view(torch.int32)
    for idx in (torch.arange(1, device=dev) + 577, torch.arange(19, device=dev) * 3, torch.arange(64, device=dev) + 200,
                torch.arange(65, device=dev) * 7 % 600, torch.arange(300, device=dev) * 2 % 600,
                torch.randperm(600, device=dev, generator=torch.Generator(device=dev).manual_seed(3))):
        e = eng.embed_windows(wav, starts[idx])
        assert bool((e.view(torch.int32) == bits[idx]).all()), "plan pinned, yet bits depend on the batch (B = %d)" % idx.shape[0]
    assert eng.set_plan_batch(0) == 0 and eng.set_plan_batch(5) == 65      # below 65 is raised: no small-batch variants
    # the pinned plan is still the same arithmetic to rounding
    assert float((eng.embed_windows(wav, starts[:19]) - full[:19]).abs().max()) < 2e-5


@pytest.mark.parametrize("G,m,k", [(2, 100, 100), (4, 58, 100), (8, 33, 100), (3, 7, 5), (8, 128, 128)])
def test_sharded_search_wave_reductions_vs_the_sorting_kernel(G, m, k):
    """pfann_bound_reduce / pfann_topk_merge_lists (one wavefront per query row, selection by bisection) against what they
    replace: the k-th largest through torch, and pfann_topk_merge (full bitonic sort per row) over the shard-major
    concatenation -- identical values, labels and ORDER, with ties inside and across lists, -inf / -1 padding, rows with
    fewer than k entries and rows with none."""
    import torch
    from pfann_amd.database import DeviceIndex
    ix = DeviceIndex(16, 0)
    dev = ix.device
    g = torch.Generator(device=dev)
    g.manual_seed(100 * G + m)
    nq = 777
    # ---- bound candidates [G, nq, m]: quantised values (many ties), -inf padding, some rows nearly empty
    c = (torch.randint(-50, 50, (G, nq, m), device=dev, generator=g).float() / 64.0)
    c[torch.rand((G, nq, m), device=dev, generator=g) < 0.2] = float("-inf")
    c[:, 5] = float("-inf")                                        # nothing present
    c[:, 6, 1:] = float("-inf")                                    # G values present
    lb = ix.reduce_bound(c, k)
    flat = c.permute(1, 0, 2).reshape(nq, G * m)
    want = torch.sort(flat, dim=1, descending=True).values[:, k - 1] if G * m >= k else torch.full((nq,), float("-inf"), device=dev)
    want = torch.where(torch.isfinite(want), want, torch.full_like(want, -3.4028234663852886e38))
    assert torch.equal(lb, want)
    # ---- shard lists [G, nq, kk]: sorted descending per list, labels unique, tails padded with (-FLT_MAX, -1)
    kk = k
    D = torch.sort(torch.randint(-30, 30, (G, nq, kk), device=dev, generator=g).float() / 32.0, dim=2, descending=True).values
    I = torch.arange(G * nq * kk, device=dev, dtype=torch.int64).reshape(G, nq, kk) * 3 + 1
    n_valid = torch.randint(0, kk + 1, (G, nq), device=dev, generator=g)
    n_valid[:, 3] = 0                                              # a row without any entry
    pad = torch.arange(kk, device=dev)[None, None, :] >= n_valid[:, :, None]
    D = torch.where(pad, torch.full_like(D, -3.4028234663852886e38), D).contiguous()
    I = torch.where(pad, torch.full_like(I, -1), I).contiguous()
    Dn, In = ix.merge_lists(D, I, k)
    S = D.permute(1, 0, 2).reshape(nq, G * kk).contiguous()
    L = I.permute(1, 0, 2).reshape(nq, G * kk).contiguous()
    Do, Io = ix.merge_topk(S, L, k)
    assert torch.equal(Dn, Do) and torch.equal(In, Io)
    assert bool((In[3] == -1).all())


def test_owned_songs_follow_the_callers_cut_at_a_rowless_boundary_song(torch_cuda):
    """song_pos = [0, 10, 10, 20] cut for two ranks by dist.shard_songs gives rank 0 the songs (0, 1) and
    rank 1 the ROWLESS song 1 with song 2 -- while the library, deriving the songs from the row range [0, 10), used to
    claim (0, 2) for rank 0: its owned score block was then one column wider than Database.song_range.  The cut is now
    stated once (pfann_db_set_owned_songs) and used by the owner-side matcher and the owned block alike."""
    torch = torch_cuda
    from pfann_amd import lib as L
    from pfann_amd.database import DeviceIndex
    from pfann_amd.dist import shard_songs
    pos = np.array([0, 10, 10, 20], np.int64)
    cut = shard_songs(pos, 2)
    assert cut == [(0, 1), (1, 3)]
    z = synth.unit_rows(3, "rows", 20, 128)
    shards = []
    for lo, hi in cut:
        ix = DeviceIndex(128, 0)
        ix.load(z[pos[lo]:pos[hi]], pos, int(pos[lo]))
        derived = ix.owned_songs()
        ix.load(z[pos[lo]:pos[hi]], pos, int(pos[lo]), song_range=(lo, hi))
        assert ix.owned_songs() == (lo, hi)
        shards.append((ix, derived))
    assert shards[0][1] == (0, 2)                                   # what the row range alone says: the ambiguity
    q = torch.as_tensor(z[12:17]).cuda()                            # song 2, offset 2
    I = torch.arange(10, 20, dtype=torch.int64).repeat(5, 1).cuda()
    for (ix, _), (lo, hi) in zip(shards, cut):
        res, ss = ix.match(q, I, [0], [5], 1, 0.0, 0, True, True, owned_block=True)
        assert ss.shape == (1, hi - lo, 2)
        if lo == 1:
            assert int(res[0]["song"]) == 2 and int(res[0]["offset"]) == 2 and abs(float(ss[0, 1, 0]) - 1.0) < 1e-6
            assert float(ss[0, 0].abs().sum()) == 0.0             # the rowless song's column
        else:
            assert int(res[0]["song"]) == -1
    with pytest.raises(L.PfannError):                               # a cut that does not span the shard's rows
        shards[0][0].load(z[0:10], pos, 0, song_range=(0, 3))


def test_match_phased_and_single_launch_forms_agree(torch_cuda):
    """The sequence matcher has two launch plans: up to 64 queries the three-launch form (candidates | scores spread over all
    CUs | argmax; round 6 raised its limit from 16), above it one workgroup per query.  The same 48 queries alone (phased) and
    as the head of a batch of 200 (single launch) must give the same (song, offset, shift) and bit-identical scores, and the
    same per-song score block."""
    torch = torch_cuda
    from pfann_amd.database import DeviceIndex
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    n_songs, seg, d, ql = 3000, 40, 128, 19
    db = torch.nn.functional.normalize(torch.randn((n_songs * seg, d), device="cuda", generator=g), dim=1)
    pos = np.arange(n_songs + 1, dtype=np.int64) * seg
    nQ = 200
    src = (torch.arange(nQ, device="cuda") * 577 + 3) % (n_songs * seg - 40)
    rows = (src[:, None] + torch.arange(ql, device="cuda")[None, :]).reshape(-1)
    q = torch.nn.functional.normalize(db[rows] + 0.6 * torch.randn((nQ * ql, d), device="cuda", generator=g), dim=1).contiguous()
    ix = DeviceIndex(d, 0)
    ix.load(db, pos, 0)
    _, I = ix.search(q, 100)
    qs, qn = np.arange(nQ, dtype=np.int64) * ql, np.full(nQ, ql, np.int32)
    res_all, ss_all = ix.match(q, I, qs, qn, 1, 0.0, 0, False, True, to_host=False)
    res_all, ss_all = res_all.clone(), ss_all.clone()
    for n_head in (17, 48, 64):
        res_h, ss_h = ix.match(q[: n_head * ql].contiguous(), I[: n_head * ql].contiguous(), qs[:n_head], qn[:n_head], 1, 0.0, 0,
                               False, True, to_host=False)
        assert torch.equal(res_h, res_all[:n_head]), "decisions of the phased form differ at %d queries" % n_head
        assert torch.equal(ss_h, ss_all[:n_head]), "per-song scores of the phased form differ at %d queries" % n_head


def test_melspec_is_bit_stable_beside_a_batched_search(torch_cuda):
    """Round 5 finding (profiles/r5/NOTES.md): with packed-fp32 VALU instructions in it, melspec_kernel returned wrong FFT
    bins for a few windows per launch whenever the batched fp16 scan ran on ANOTHER stream at the same time (found through
    PFANN_EXCHANGE_STREAM=1; two host threads driving an Engine and an index would have hit it too).  mel.hip is built
    without the SLP vectoriser since (pfann_amd/build.py); this keeps it that way: 4085 windows, six launches, each beside
    three searches of 4085 query rows on a side stream -- every bit of the log-mel must equal the quiet run's."""
    torch = torch_cuda
    from pfann_amd.database import DeviceIndex
    from pfann_amd.engine import Engine
    params = cfg("default")
    eng = Engine(params, 0, max_batch=4096)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    segs = torch.randn((4085, 8000), device="cuda", generator=g) * 0.1
    db = torch.nn.functional.normalize(torch.randn((120000, 128), device="cuda", generator=g), dim=1)
    q = torch.nn.functional.normalize(db[:4085] + 0.3 * torch.randn((4085, 128), device="cuda", generator=g), dim=1).contiguous()
    ix = DeviceIndex(128, 0)
    ix.load(db, np.array([0, 120000], np.int64), 0)
    quiet = eng.melspec(segs).clone()
    ix.search(q, 100)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(3):
                ix.search(q, 100)
        got = eng.melspec(segs)
        torch.cuda.synchronize()
        bad = int((got != quiet).reshape(4085, -1).any(dim=1).sum())
        assert bad == 0, "launch %d: %d windows of the log-mel differ from the quiet run (max %.3g)" % (
            rep, bad, float((got - quiet).abs().max()))


AGGRESSORS = {
    # name: (d, query rows, db rows) -> the scan-class kernels one search call of that shape launches (csrc/search.hip: search_topk)
    "d64_qres": (64, 4085, 120000),         # scan_f16_qres_kernel<4,true> (sampled pass) + scan_f16_qres_kernel<4> (full pass)
    "d128_generic": (128, 1000, 300000),    # < 1024 query rows: the survivor ladder on scan_f16_kernel<1> (sampled levels) and <4> (full pass)
    "d128_f16_storage": (128, 4085, 120000),  # fp16-only storage: scan_f16_qres_kernel<8,true,128> + <8,false,64>, s16 scores final
}


@pytest.mark.parametrize("aggressor", sorted(AGGRESSORS))
@pytest.mark.parametrize("victim", ["stft1024", "fft512"])
def test_melspec_is_bit_stable_beside_every_scan_class_kernel(torch_cuda, aggressor, victim):
    """Round 6: the guard above covers scan_f16_qres_kernel<8,true,128> + <8,false,64> (one d = 128 search
    call runs both) beside the default log-mel.  The same watch for the other instantiations of the fp16-MFMA scan -- d = 64
    (KS = 4), the generic scan_f16_kernel<1> / <4> of the survivor ladder, fp16-only storage -- and for the second FFT size
    of melspec_kernel (stft_n = 512: a different radix plan): every bit of the log-mel equal to the quiet run's."""
    torch = torch_cuda
    from pfann_amd.database import DeviceIndex
    from pfann_amd.engine import Engine
    params = cfg("default")
    if victim == "fft512":
        params.update(stft_n=512, stft_hop=128, n_mels=96)
    d, nq, n = AGGRESSORS[aggressor]
    eng = Engine(params, 0, max_batch=4096)
    g = torch.Generator(device="cuda")
    g.manual_seed(15)
    segs = torch.randn((4085, 8000), device="cuda", generator=g) * 0.1
    db = torch.nn.functional.normalize(torch.randn((n, d), device="cuda", generator=g), dim=1)
    q = torch.nn.functional.normalize(db[:nq] + 0.3 * torch.randn((nq, d), device="cuda", generator=g), dim=1).contiguous()
    ix = DeviceIndex(d, 0, storage="f16") if aggressor == "d128_f16_storage" else DeviceIndex(d, 0)
    ix.load(db, np.array([0, n], np.int64), 0)
    quiet = eng.melspec(segs).clone()
    ix.search(q, 100)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(3):
                ix.search(q, 100)
        got = eng.melspec(segs)
        torch.cuda.synchronize()
        bad = int((got != quiet).reshape(4085, -1).any(dim=1).sum())
        assert bad == 0, "%s beside %s, launch %d: %d windows of the log-mel differ from the quiet run (max %.3g)" % (
            victim, aggressor, rep, bad, float((got - quiet).abs().max()))


def test_other_kernels_are_bit_stable_beside_a_batched_search(torch_cuda):
    """The same question for the rest of the path (profiles/r5/NOTES.md section 6): encoder, sequence matcher and the
    small-batch search on the caller's stream while a batched fp16 search runs on a side stream -- outputs equal to the quiet
    run's, bit for bit.  (They were never seen to move; this keeps watch.)"""
    torch = torch_cuda
    from pfann_amd.database import DeviceIndex
    from pfann_amd.engine import Engine
    params = cfg("default")
    sd = synth.make_state_dict(params, seed=11)
    eng = Engine(params, 0, max_batch=1024)
    eng.load_state_dict(sd)
    g = torch.Generator(device="cuda")
    g.manual_seed(6)
    mel = torch.randn((1024, 256, 32), device="cuda", generator=g)
    n_songs, seg = 3000, 40
    db = torch.nn.functional.normalize(torch.randn((n_songs * seg, 128), device="cuda", generator=g), dim=1)
    pos = np.arange(n_songs + 1, dtype=np.int64) * seg
    q = torch.nn.functional.normalize(db[:4085] + 0.3 * torch.randn((4085, 128), device="cuda", generator=g), dim=1).contiguous()
    ix, other = DeviceIndex(128, 0), DeviceIndex(128, 0)
    ix.load(db, pos, 0)
    other.load(db, pos, 0)                      # (a second handle: the side stream's search must not share a workspace)
    nq = 215
    qs, ql = np.arange(nq, dtype=np.int64) * 19, np.full(nq, 19, np.int32)
    _, I = ix.search(q, 100)
    quiet_emb = eng.encode(mel).clone()
    quiet_res, quiet_ss = ix.match(q, I, qs, ql, 1, 0.0, 0, False, True, to_host=False)
    quiet_res, quiet_ss = quiet_res.clone(), quiet_ss.clone()
    quiet_D, quiet_I = ix.search(q[:19].contiguous(), 100)
    quiet_D, quiet_I = quiet_D.clone(), quiet_I.clone()
    other.search(q, 100)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(4):
                other.search(q, 100)
        emb = eng.encode(mel)
        res, ss = ix.match(q, I, qs, ql, 1, 0.0, 0, False, True, to_host=False)
        D1, I1 = ix.search(q[:19].contiguous(), 100)
        torch.cuda.synchronize()
        assert torch.equal(emb, quiet_emb), "encoder output moved beside a search (launch %d)" % rep
        assert torch.equal(res, quiet_res) and torch.equal(ss, quiet_ss), "matcher output moved beside a search (launch %d)" % rep
        assert torch.equal(D1, quiet_D) and torch.equal(I1, quiet_I), "small-batch search moved beside a search (launch %d)" % rep
