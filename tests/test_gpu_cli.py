"""End-to-end drop-in check of the two CLIs on the GPU: builder.py <list> <db> <cfg>, then
matcher.py <qlist> <db> <result>, against the oracle pipeline run on the same files
(BASELINE.json configs[0]: 10-song synthetic db, plumbing + decisions)."""
import csv
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from pfann_amd import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_embed(path, params, sd):
    """The oracle pipeline for one file: segmenter -> mel -> encoder (unit-norm rows)."""
    from oracle import encoder as oe
    from oracle import melspec as om
    from oracle import segmenter as osg
    segs = osg.load_segments(path, params)
    return oe.encode(om.melspec(segs, params), sd, params)


@pytest.mark.parametrize("cfgname", ["tiny", "default"])
def test_builder_and_matcher_cli_vs_oracle(tmp_path, cfgname):
    import torch
    from oracle import encoder as oe
    from oracle import melspec as om
    from oracle import search as osr
    from oracle import segmenter as osg
    from oracle import seqscore as osq

    params = json.load(open(os.path.join(REPO, "configs", cfgname + ".json")))
    d, k = params["model"]["d"], params["indexer"]["top_k"]
    sd = synth.make_state_dict(params, seed=321)
    mdir = tmp_path / "model"
    mdir.mkdir()
    torch.save({n: torch.from_numpy(v) for n, v in sd.items()}, str(mdir / "model.pt"))
    shutil.copy(os.path.join(REPO, "configs", cfgname + ".json"), str(mdir / "configs.json"))

    # ---- music: 10 songs of different lengths, one stereo, one unreadable
    n_songs = 10
    music = []
    songs = {}
    for s in range(n_songs):
        path = str(tmp_path / ("song%02d.wav" % s))
        if s == 4:
            open(path, "wb").write(b"garbage")           # load error -> 0-segment song
        else:
            pcm = synth.make_song(100 + s, seconds=8.0 + s)
            if s == 6:
                pcm = np.stack([pcm, pcm // 2], 1)
            # song 8 is stored at 16 kHz (its samples read twice as fast): the builder resamples it on the device,
            # the oracle pipeline through oracle/resample.py
            synth.write_wav(path, pcm, sr=16000 if s == 8 else 8000)
            songs[s] = pcm
        music.append(path)
    mlist = tmp_path / "music.txt"
    mlist.write_text("".join(p + "\n" for p in music))

    # ---- queries: clean + noisy crops, a too-short one, a missing file
    queries, truth = [], []
    for j in range(8):
        s = [0, 1, 2, 3, 5, 6, 7, 9][j]
        src = songs[s] if songs[s].ndim == 1 else songs[s][:, 0]
        q, off = synth.make_query(src, j, 4.0, snr_db=6.0 if j % 2 else 30.0)
        path = str(tmp_path / ("q%02d.wav" % j))
        synth.write_wav(path, q)
        queries.append(path)
        truth.append((s, off))
    short = str(tmp_path / "qshort.wav")
    synth.write_wav(short, songs[2][16000:16000 + 3000])
    queries.append(short)
    queries.append(str(tmp_path / "qmissing.wav"))
    qlist = tmp_path / "queries.txt"
    qlist.write_text("".join(p + "\n" for p in queries))

    db = str(tmp_path / "db")
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, os.path.join(REPO, "builder.py"), str(mlist), db, str(mdir)],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    result = str(tmp_path / "result.txt")
    r = subprocess.run([sys.executable, os.path.join(REPO, "matcher.py"), str(qlist), db, result],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]

    # ---- db directory layout (builder.py:71,136-148)
    for f in ("embeddings", "landmarkValue", "landmarkKey", "songList.txt", "configs.json", "model.pt"):
        assert os.path.exists(os.path.join(db, f)), f
    key = np.fromfile(os.path.join(db, "landmarkKey"), dtype=np.int32)
    emb = np.fromfile(os.path.join(db, "embeddings"), dtype=np.float32).reshape(-1, d)
    assert open(os.path.join(db, "songList.txt")).read() == mlist.read_text()

    # ---- oracle pipeline on the same files
    ref_emb, ref_key = [], []
    for p in music:
        segs = osg.load_segments(p, params)
        ref_key.append(segs.shape[0])
        if segs.shape[0]:
            ref_emb.append(oe.encode(om.melspec(segs, params), sd, params))
    ref_emb = np.concatenate(ref_emb)
    assert np.array_equal(key, np.asarray(ref_key, np.int32)) and key[4] == 0
    assert emb.shape == ref_emb.shape
    assert np.abs(emb - ref_emb).max() < 1e-4
    pos = osq.song_pos_from_key(ref_key)

    tsv = [ln.rstrip("\n").split("\t") for ln in open(result, encoding="utf8")]
    raw_detail = open(os.path.splitext(result)[0] + "_detail.csv", "rb").read()
    assert raw_detail.startswith(b"query,answer,score,time,part_scores\r\n")       # csv module line endings
    detail = list(csv.reader(open(os.path.splitext(result)[0] + "_detail.csv", newline="")))[1:]
    scores = np.fromfile(result + ".bin", dtype=np.float32).reshape(len(queries), n_songs, 2)
    assert len(tsv) == len(detail) == len(queries)
    hits = 0
    for j, qp in enumerate(queries):
        segs = osg.load_segments(qp, params)
        assert tsv[j][0] == qp == detail[j][0]
        if segs.shape[0] == 0:                                      # matcher.py:94-107
            assert tsv[j][1] == "error" and detail[j][1:] == ["error", "-inf", "0"]
            assert not scores[j].any()
            continue
        e = oe.encode(om.melspec(segs, params), sd, params)
        D, I = osr.flat_ip_topk(e, ref_emb, k)
        sc, (song, sec), ss = osq.query_embeddings_base(e, I, ref_emb, pos, params["hop_size"], 1)
        assert tsv[j][1] == music[song] == detail[j][1], j
        assert float(detail[j][3]) == sec, j
        assert abs(float(detail[j][2]) - sc) < 2e-5, j
        assert np.allclose(scores[j], ss, atol=2e-5), j
        assert np.array_equal(scores[j][:, 1], ss[:, 1])
        if j < len(truth):
            hits += (song == truth[j][0] and abs(sec - truth[j][1]) <= 0.5)
    if cfgname == "default":
        assert hits >= 6          # random-weight encoder still identifies clean crops


def test_extractemb_matchemb_seam_and_accuracy(tmp_path):
    """extractemb -> matchemb reproduces matcher's outputs byte for byte,
    and the accuracy evaluator reads them; synthetic dataset + SNR protocol of genquery/genall."""
    import torch
    params = json.load(open(os.path.join(REPO, "configs", "tiny.json")))
    sd = synth.make_state_dict(params, seed=11)
    mdir = tmp_path / "model"
    mdir.mkdir()
    torch.save({n: torch.from_numpy(v) for n, v in sd.items()}, str(mdir / "model.pt"))
    shutil.copy(os.path.join(REPO, "configs", "tiny.json"), str(mdir / "configs.json"))
    env = dict(os.environ, PYTHONPATH=REPO)
    data = str(tmp_path / "data")

    def run(*cmd):
        r = subprocess.run([sys.executable] + list(cmd), capture_output=True, text=True, env=env,
                           cwd=str(tmp_path), timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return r.stdout

    run(os.path.join(REPO, "tools", "gen_synth_dataset.py"), data, "--songs", "8", "--queries", "12",
        "--seconds", "5", "--song-seconds", "12", "--snr", "20", "0")
    open(os.path.join(data, "query_snr20", "list.txt"), "a").write(os.path.join(data, "nope.wav") + "\n")
    db = str(tmp_path / "db")
    run(os.path.join(REPO, "builder.py"), os.path.join(data, "music.txt"), db, str(mdir))
    for snr in ("20", "0"):
        qd = os.path.join(data, "query_snr" + snr)
        r1, r2 = str(tmp_path / ("m%s.txt" % snr)), str(tmp_path / ("e%s.txt" % snr))
        run(os.path.join(REPO, "matcher.py"), os.path.join(qd, "list.txt"), db, r1)
        run(os.path.join(REPO, "extractemb.py"), os.path.join(qd, "list.txt"), db, str(tmp_path / ("emb" + snr)))
        run(os.path.join(REPO, "matchemb.py"), str(tmp_path / ("emb" + snr)), db, r2)
        for suffix in ("", ".bin"):
            assert open(r1 + suffix, "rb").read() == open(r2 + suffix, "rb").read(), (snr, suffix)
        d1 = open(os.path.splitext(r1)[0] + "_detail.csv", "rb").read()
        d2 = open(os.path.splitext(r2)[0] + "_detail.csv", "rb").read()
        assert d1 == d2
        qi = np.fromfile(os.path.join(str(tmp_path / ("emb" + snr)), "query_index"), dtype=np.int64).reshape(-1, 2)
        assert qi.shape[0] == (13 if snr == "20" else 12) and (qi[:12, 1] == 9).all()
        # the seam's files against the ORACLE (not against the product itself): query_index is the running
        # (start, count) table of extractemb.py:66-84, query_embeddings the oracle pipeline's unit-norm rows
        assert np.array_equal(qi[:12, 0], np.arange(12) * 9)
        qe = np.fromfile(os.path.join(str(tmp_path / ("emb" + snr)), "query_embeddings"), dtype=np.float32).reshape(-1, 16)
        assert qe.shape[0] == 12 * 9
        qfiles = [ln.strip() for ln in open(os.path.join(qd, "list.txt"))][:12]
        ref = np.concatenate([oracle_embed(f, params, sd) for f in qfiles])
        assert np.abs(qe - ref).max() < 1e-4, np.abs(qe - ref).max()
        assert np.abs(np.linalg.norm(qe, axis=1) - 1).max() < 1e-5
        # ... and matchemb's decisions against the oracle's search + sequence matcher on those oracle rows
        from oracle import search as osr
        from oracle import seqscore as osq
        dbe = np.fromfile(os.path.join(db, "embeddings"), dtype=np.float32).reshape(-1, 16)
        pos = osq.song_pos_from_key(np.fromfile(os.path.join(db, "landmarkKey"), dtype=np.int32))
        det = list(csv.reader(open(os.path.splitext(r2)[0] + "_detail.csv", newline="")))[1:]
        music = [ln.strip() for ln in open(os.path.join(data, "music.txt"))]
        for j in range(12):
            qj = qe[j * 9:(j + 1) * 9]
            _, I = osr.flat_ip_topk(qj, dbe, params["indexer"]["top_k"])
            sc, (song, sec), _ = osq.query_embeddings_base(qj, I, dbe, pos, params["hop_size"], 1)
            assert det[j][1] == music[song] and float(det[j][3]) == sec, j
            assert abs(float(det[j][2]) - sc) < 2e-5, j
        if snr == "20":
            assert qi[12, 1] == 0                                   # unreadable file: (pos, 0)
            lines = [ln for ln in open(os.path.splitext(r1)[0] + "_detail.csv", newline="")]
            assert lines[-1].split(",")[1:3] == ["error", "-inf"]
            # evaluator over the 12 real queries
            good = str(tmp_path / "good_detail.csv")
            open(good, "w", newline="").write("".join(lines[:-1]))
            out = run(os.path.join(REPO, "tools", "accuracy.py"), os.path.join(qd, "expected.csv"), good)
            assert "song correct" in out and "near match correct" in out and "exact match correct" in out
            acc = float(out.strip().splitlines()[-1].split()[-1])
            assert acc >= 50.0          # tiny random-weight encoder at 20 dB: well above chance (12.5 %)


@pytest.mark.parametrize("variant", ["frame_shift_mul2", "native_score_alpha", "use_float16"])
def test_matcher_cli_variants_vs_oracle(tmp_path, variant):
    """The matcher's variants END TO END through the CLI (they were library-level only):
      frame_shift_mul=2 in configs.json (query windows every half hop, per-shift candidate sets, time =
          (t - shift/fsm) * hop: database.py:129-148)                       vs oracle/seqscore.py
      score_alpha=2.0 with the reference's native path switched on (PFANN_CPP_ACCELERATE=1 = database.py:12
          cpp_accelerate; exp(-alpha (1-ip)^2) scores, score/time read back from the per-song block, times scaled
          by hop/fsm in float32: database.py:166-195, seqscore.cpp:100-110), together with frame_shift_mul=2
                                                                            vs oracle/seqscore_c.c
      "use_float16" in the indexer block (fp16-only storage, faiss' useFloat16, database.py:101-104)
                                                                            vs oracle flat_ip_topk_f16 + fp16-rounded rows
    The oracle searches and matches the CLI's OWN query fingerprints (extractemb.py writes the bits matcher.py used:
    both pin the kernel plan), so decisions and times must agree exactly and scores to rounding; the fingerprints
    themselves are checked against the oracle encoder."""
    import torch
    from oracle import native
    from oracle import search as osr
    from oracle import seqscore as osq
    params = json.load(open(os.path.join(REPO, "configs", "tiny.json")))
    fsm = 1
    env = dict(os.environ, PYTHONPATH=REPO)
    if variant in ("frame_shift_mul2", "native_score_alpha"):
        params["indexer"]["frame_shift_mul"] = fsm = 2
    if variant == "native_score_alpha":
        params["indexer"]["score_alpha"] = 2.0
        env["PFANN_CPP_ACCELERATE"] = "1"
    if variant == "use_float16":
        params["indexer"]["use_float16"] = True
    d, k, hop_s = params["model"]["d"], params["indexer"]["top_k"], params["hop_size"]
    sd = synth.make_state_dict(params, seed=77)
    mdir = tmp_path / "model"
    mdir.mkdir()
    torch.save({n: torch.from_numpy(v) for n, v in sd.items()}, str(mdir / "model.pt"))
    json.dump(params, open(str(mdir / "configs.json"), "w"))
    music, songs = [], []
    for s in range(9):
        path = str(tmp_path / ("song%d.wav" % s))
        songs.append(synth.make_song(300 + s, seconds=9.0 + s))
        synth.write_wav(path, songs[-1])
        music.append(path)
    music.append(music[2])                                        # a duplicated song: ties -> the lower id
    (tmp_path / "music.txt").write_text("".join(p + "\n" for p in music))
    queries = []
    for j in range(12):
        q, _ = synth.make_query(songs[j % 9], 40 + j, 4.0 + (j % 3), snr_db=[30.0, 10.0, 3.0][j % 3])
        path = str(tmp_path / ("q%02d.wav" % j))
        synth.write_wav(path, q)
        queries.append(path)
    queries.insert(5, str(tmp_path / "nope.wav"))
    (tmp_path / "queries.txt").write_text("".join(p + "\n" for p in queries))

    def run(*cmd):
        r = subprocess.run([sys.executable] + list(cmd), capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    db, result, edir = str(tmp_path / "db"), str(tmp_path / "result.txt"), str(tmp_path / "emb")
    run(os.path.join(REPO, "builder.py"), str(tmp_path / "music.txt"), db, str(mdir))
    run(os.path.join(REPO, "matcher.py"), str(tmp_path / "queries.txt"), db, result)
    run(os.path.join(REPO, "extractemb.py"), str(tmp_path / "queries.txt"), db, edir)
    assert json.load(open(os.path.join(db, "configs.json")))["indexer"].get("frame_shift_mul", 1) == fsm
    dbe = np.fromfile(os.path.join(db, "embeddings"), np.float32).reshape(-1, d)
    key = np.fromfile(os.path.join(db, "landmarkKey"), np.int32)
    pos = osq.song_pos_from_key(key)
    assert key[0] == (9 * 8000 - 8000) // 4000 + 1                 # the builder hops whole frames whatever fsm says (builder.py:64)
    qe = np.fromfile(os.path.join(edir, "query_embeddings"), np.float32).reshape(-1, d)
    qi = np.fromfile(os.path.join(edir, "query_index"), np.int64).reshape(-1, 2)
    detail = list(csv.reader(open(os.path.splitext(result)[0] + "_detail.csv", newline="")))[1:]
    blocks = np.fromfile(result + ".bin", np.float32).reshape(len(queries), len(music), 2)
    worst_emb = 0.0
    for j, qp in enumerate(queries):
        if qi[j, 1] == 0:
            assert detail[j][1:] == ["error", "-inf", "0"] and not blocks[j].any()
            continue
        e = qe[qi[j, 0]: qi[j, 0] + qi[j, 1]]
        ref = oracle_embed(qp, params, sd)
        assert ref.shape == e.shape
        worst_emb = max(worst_emb, float(np.abs(ref - e).max()))
        if variant == "use_float16":
            _, I = osr.flat_ip_topk_f16(e, dbe, k)
            rows = dbe.astype(np.float16).astype(np.float32)
            tol = 2e-4                                               # the kernel accumulates the fp16 rows in fp32
        else:
            _, I = osr.flat_ip_topk(e, dbe, k)
            rows, tol = dbe, 2e-6
        if variant == "native_score_alpha":
            best, ss = native.seq_score(rows, pos, e, I, fsm, 2.0)
            sc, song, sec = float(ss[best, 0]), best, float(ss[best, 1]) * hop_s / fsm
            ss[:, 1] *= hop_s / fsm
        else:
            sc, (song, sec), ss = osq.query_embeddings_base(e, I, rows, pos, hop_s, fsm)
        assert detail[j][1] == music[song], (variant, j, detail[j], song)
        assert float(detail[j][3]) == sec, (variant, j, detail[j], sec)
        assert abs(float(detail[j][2]) - sc) < tol, (variant, j, detail[j], sc)
        assert np.array_equal(blocks[j][:, 1], ss[:, 1]), (variant, j)
        assert np.abs(blocks[j][:, 0] - ss[:, 0]).max() < tol
        assert song != 9                                            # the duplicate of song 2 never wins the tie
    assert worst_emb < 1e-4


@pytest.mark.parametrize("config", ["default", "seg", "n640d64"])
def test_oracle_builds_its_own_database_from_the_same_files(config):
    """The fully independent form of end-to-end parity (the 2,000- and 10,000-song records of the same
    tool are under profiles/r5/ and -- round 6: the two depthwise models `seg.json` / `n640d64.json` and `default` at -6 dB --
    profiles/r6/): `builder.py` + `matcher.py` as subprocesses on 96 songs / 48 ten-second SNR-0 queries,
    against an oracle that reads the same WAV files with its own reader, embeds every song ON THE HOST into its own
    database, and answers the queries against THAT -- nothing shared but the files and the weights.  The product's files
    must agree: `landmarkKey` exactly, `embeddings` within 1e-4 (3e-5 expected: both sides build the mel bank the way
    torchaudio does), every (answer, time) of the TSV / `_detail.csv`, scores within 1e-5."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import decision_parity_oracle_db as dp
    workers = max(4, min(32, (os.cpu_count() or 8) // 8))
    out = dp.run(96, 48, 0.0, workers=workers, log=lambda *a: print(*a, file=sys.stderr, flush=True), config=config)
    assert "skipped" not in out, out
    os.makedirs(os.path.join(REPO, "gpurun_out", "r6"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "r6", "decision_parity_oracle_db_test_%s.json" % config), "w"), indent=1)
    assert out["landmarkKey_equal"] and out["embeddings_rows"] == 96 * 59 and out["tsv_and_detail_csv_agree"]
    assert out["embeddings_max_abs_diff"] < 1e-4, out["embeddings_max_abs_diff"]
    assert out["bugs"] == 0, out["flips"]
    assert out["identical_song_and_time"] + len(out["flips"]) == 48 and out["identical_song_and_time"] >= 47
    assert out["max_score_abs_diff_where_decisions_agree"] < 1e-5
    assert out["top1_hit_rate_product"] == out["top1_hit_rate_oracle"] or out["flips"]
