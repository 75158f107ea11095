"""The oracle (oracle/) against the golden vectors generated from the reference's own code
(tests/golden/make_golden.py) and against the probed outputs SURVEY.md §8c records."""
import json
import os

import numpy as np
import pytest

import make_golden as mg
from oracle import encoder, melspec, native, search, segmenter, seqscore
from pfann_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["default", "seg", "n640d64", "tiny", "nafstyle", "elu_full", "strides_pow2", "strides_np2"])
def test_encoder_matches_reference(name):
    z = np.load(os.path.join(G, "encoder_%s.npz" % name))
    params = json.loads(str(z["params"]))
    _, _, _, F, T = synth.model_dims(params)
    sd = synth.make_state_dict(params, seed=123)
    x = mg.encoder_inputs(F, T)
    taps = []
    emb = encoder.encode(x, sd, params, norm=True, taps=taps)
    raw = encoder.encode(x, sd, params, norm=False)
    assert np.abs(emb - z["emb"]).max() < 2e-6
    assert np.abs(raw - z["raw"]).max() < 2e-5 * max(1.0, np.abs(z["raw"]).max())
    sums = np.array([[t.astype(np.float64).sum(), np.abs(t.astype(np.float64)).sum()] for t in taps])
    assert np.allclose(sums, z["tap_sums"], rtol=1e-5, atol=1e-2)


def test_segmenter_matches_reference(tmp_path):
    z = np.load(os.path.join(G, "segmenter.npz"))
    params = json.load(open(os.path.join(os.path.dirname(G), "..", "configs", "default.json")))
    inputs = mg.segmenter_inputs()
    for name, pcm in inputs.items():
        synth.write_wav(str(tmp_path / (name + ".wav")), pcm)
    (tmp_path / "notwav.wav").write_bytes(b"this is not a wave file")
    for fsm in (1, 2):
        p = json.loads(json.dumps(params))
        p["indexer"]["frame_shift_mul"] = fsm
        for name in list(inputs) + ["missing", "notwav"]:
            w = segmenter.load_segments(str(tmp_path / (name + ".wav")), p)
            key = "%s_fsm%d" % (name, fsm)
            assert tuple(z[key + "_shape"]) == w.shape, key
            if w.shape[0]:
                # bit-identical rows, and checksums over every row
                assert np.array_equal(w[z[key + "_rows"]], z[key + "_vals"]), key
                sums = np.stack([w.astype(np.float64).sum(1), np.abs(w.astype(np.float64)).sum(1)], 1)
                assert np.allclose(sums, z[key + "_sums"], rtol=0, atol=1e-9), key


DB_CASES = ["clean_hit", "negative_offset", "past_end", "k_gt_ntotal", "duplicate_songs",
            "nonpositive_best", "no_candidates", "frame_shift_mul2", "empty_db", "random_noisy"]


@pytest.mark.parametrize("name", DB_CASES)
def test_seqscore_python_path_matches_reference(name):
    z = np.load(os.path.join(G, "database.npz"))
    db, q, labels = z[name + "_db"], z[name + "_q"], z[name + "_labels"]
    pos = seqscore.song_pos_from_key(z[name + "_key"])
    fsm = int(z[name + "_fsm"])
    score, (song, sec), ss = seqscore.query_embeddings_base(q, labels, db, pos, 0.5, fsm)
    assert song == int(z[name + "_song"])
    assert sec == float(z[name + "_sec"])
    assert score == float(z[name + "_score"]) or abs(score - float(z[name + "_score"])) < 1e-6
    assert np.allclose(ss, z[name + "_song_score"], atol=1e-6)
    # the labels the reference searched with are the exact flat top-k
    if name not in ("no_candidates", "nonpositive_best") and db.shape[0]:
        D, I = search.flat_ip_topk(q, db, labels.shape[1])
        assert np.array_equal(I, labels)
        D2, I2 = native.flat_ip_topk(q, db, labels.shape[1])
        assert np.array_equal(I2, labels)
        assert np.abs(D - D2).max() < 1e-6


@pytest.mark.parametrize("name", DB_CASES)
def test_seqscore_c_path(name):
    """C restatement of cpp/seqscore.cpp: agrees with the Python path wherever SURVEY.md
    §8c says the two reference paths agree, and shows the documented divergences."""
    z = np.load(os.path.join(G, "database.npz"))
    db, q, labels = z[name + "_db"], z[name + "_q"], z[name + "_labels"]
    if db.shape[0] == 0:
        pytest.skip("cpp path has no empty-db branch (database.py:126-127 is python-only)")
    pos = seqscore.song_pos_from_key(z[name + "_key"])
    fsm = int(z[name + "_fsm"])
    best, ss = native.seq_score(db, pos, q, labels, fsm, 0.0)
    hop = 0.5
    if name == "no_candidates":
        assert best == -1 and not ss.any()
        return
    assert best == int(z[name + "_song"])
    if name == "nonpositive_best":
        # song_scores never records a non-positive score; caller reads back 0.0 / 0.0
        assert not ss.any()
        return
    # caller-side scaling of database.py:190-193
    assert abs(ss[best, 0] - float(z[name + "_score"])) < 1e-6
    assert ss[best, 1] * hop / fsm == float(z[name + "_sec"])
    ss2 = ss.copy()
    ss2[:, 1] *= hop / fsm
    assert np.allclose(ss2, z[name + "_song_score"], atol=1e-6)


def test_seqscore_c_probed_values():
    """Known answers recorded from the compiled reference in SURVEY.md §8c:
    4 of 6 rows match with divisor 6 -> 0.6666666269 in fp32."""
    z = np.load(os.path.join(G, "database.npz"))
    n = "negative_offset"
    pos = seqscore.song_pos_from_key(z[n + "_key"])
    best, ss = native.seq_score(z[n + "_db"], pos, z[n + "_q"], z[n + "_labels"], 1, 0.0)
    assert best == 1 and ss[1, 1] == -2.0
    assert abs(float(ss[1, 0]) - 0.6666666269) < 1e-7


def test_seqscore_c_alpha():
    z = np.load(os.path.join(G, "database.npz"))
    n = "random_noisy"
    pos = seqscore.song_pos_from_key(z[n + "_key"])
    db, q, lab = z[n + "_db"], z[n + "_q"], z[n + "_labels"]
    best, ss = native.seq_score(db, pos, q, lab, 1, 2.0)
    assert best == 17
    off = int(ss[17, 1])
    ips = np.array([db[pos[17] + off + i] @ q[i] for i in range(19)], np.float32)
    assert abs(ss[17, 0] - np.exp(-2.0 * (1 - ips) ** 2).mean()) < 1e-5


def test_melspec_restatement_self_consistency():
    """a2 is parity-unpinned (torchaudio absent); check the fp32 torch.stft restatement
    against the independent float64 gather+rfft form and the documented bank shape."""
    params = json.load(open(os.path.join(os.path.dirname(G), "..", "configs", "default.json")))
    fb = melspec.mel_filterbank(8000, 1024, 256, 300, 4000).numpy()
    nz = fb > 0
    assert nz.sum() == 942 and nz.sum(0).max() == 7 and nz.sum(0).min() >= 1
    rows = np.nonzero(nz.sum(1))[0]
    assert rows[0] == 39 and rows[-1] == 511
    pcm = synth.make_song(3, seconds=3.0)
    segs = segmenter.segment(segmenter.pcm_to_mono(pcm[:, None]), 8000, 4000)
    m32 = melspec.melspec(segs, params)
    m64 = melspec.melspec_f64(segs, params)
    assert m32.shape == (5, 256, 32)
    assert np.abs(m32 - m64).max() < 5e-3
    assert np.abs(m32 - m64).mean() < 1e-4


def test_melspec_oracle_vs_independent_third_party():
    """a2 cannot be pinned against torchaudio (absent, version unpinned).  Second opinion from an
    unrelated implementation of the same documented semantics that IS installed here:
    transformers.audio_utils (numpy, fp64): HTK mel bank without normalisation, periodic-hann STFT
    with centre reflect padding, power 2, then ln(x + 1e-8).  The bank differs by the fp32-vs-fp64
    construction only (<= 3.9e-5); the log-mel agrees to 1e-4 in the audible bins."""
    au = pytest.importorskip("transformers.audio_utils")
    from oracle import melspec as om
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = json.load(open(os.path.join(repo, "configs", "default.json")))
    fb = np.asarray(om.mel_filterbank(p["sample_rate"], p["stft_n"], p["n_mels"], p["f_min"], p["f_max"]))
    fb2 = au.mel_filter_bank(num_frequency_bins=p["stft_n"] // 2 + 1, num_mel_filters=p["n_mels"],
                             min_frequency=p["f_min"], max_frequency=p["f_max"], sampling_rate=p["sample_rate"],
                             norm=None, mel_scale="htk")
    assert fb.shape == fb2.shape and np.abs(fb - fb2).max() < 5e-5
    w = au.window_function(p["stft_n"], "hann", periodic=True)
    for seed in (1, 2):
        x = synth.normal(seed, "t/au", 8000).astype(np.float32)
        x[2000:6000] += np.sin(np.arange(4000) * 2 * np.pi * (700 + 300 * seed) / 8000).astype(np.float32) * 3
        x -= x.mean()
        ref = om.melspec(x[None], p)[0]
        xn = (x / max(np.linalg.norm(x), 1e-12)).astype(np.float64)
        sp = au.spectrogram(xn, w, frame_length=p["stft_n"], hop_length=p["stft_hop"], fft_length=p["stft_n"],
                            power=2.0, center=True, pad_mode="reflect", onesided=True, mel_filters=fb2,
                            mel_floor=0.0, log_mel=None)
        lm = np.log(sp + 1e-8)
        assert ref.shape == lm.shape
        loud = ref > ref.max() - 5.0
        assert np.abs(ref - lm)[loud].max() < 3e-4, np.abs(ref - lm)[loud].max()
        assert np.abs(ref - lm).max() < 3e-3


# ------------------------------------------------------------------ a1 at other sample rates (parity unpinned: julius absent)
def test_resampler_restatement_properties():
    """No reference vector exists for julius.ResampleFrac (absent, unpinned): check what the published algorithm
    guarantees -- unit-sum phases (constants preserved), the documented output length, a 1 kHz tone surviving 44.1 -> 8 kHz
    to 1e-5 away from the edges, everything above the new Nyquist removed -- and the reference's minute-wise assembly."""
    from oracle import resample as R
    k, width = R.kernels(44100, 8000)
    assert k.shape == (80, 2 * 140 + 441) and width == 140
    assert float((k.sum(1) - 1).abs().max()) < 1e-6
    assert R.resample_frac(np.full((2, 44100), 0.25, np.float32), 44100, 8000).shape == (2, 8000)
    assert np.abs(R.resample_frac(np.full((1, 50000), 0.25, np.float32), 44100, 8000) - 0.25).max() < 1e-6
    assert R.resample_frac(np.zeros((1, 12345), np.float32), 44100, 8000).shape[1] == int(80 * 12345 / 441)
    x = np.zeros((1, 777), np.float32)
    assert R.resample_frac(x, 8000, 8000) is not None and R.resample_frac(x, 16000, 16000).shape == (1, 777)
    t = np.arange(44100 * 125) / 44100.0
    tone = np.sin(2 * np.pi * 1000 * t).astype(np.float32)[None]
    y = R.resample_chunked(tone, 44100, 8000)
    assert y.shape == (1, 125 * 8000)                                  # two full pieces + tail: seams at 59.5 s and 118.5 s
    ref = np.sin(2 * np.pi * 1000 * np.arange(y.shape[1]) / 8000.0)
    assert np.abs(y[0] - ref)[4000:-4000].max() < 1e-5                 # incl. both seams
    hiss = np.sin(2 * np.pi * 6000 * t[:44100 * 3]).astype(np.float32)[None]     # above the 4 kHz Nyquist
    assert np.abs(R.resample_frac(hiss, 44100, 8000))[0, 400:-400].max() < 2e-3
    # piece plan: 59 s stride, half-second strips, lengths add up
    plan = R.chunk_plan(44100 * 125, 44100, 8000)
    assert plan[0] == (0, 2646000, 0, 476000) and plan[1] == (2601900, 2646000, 4000, 472000)
    assert sum(p[3] for p in plan) == 125 * 8000
    assert R.chunk_plan(1000, 16000, 8000) == [(0, 1000, 0, 500)]
