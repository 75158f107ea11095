"""Every binding INTEGRATION.md shows a maintainer of the reference is EXECUTED here, verbatim: the marked code blocks
are cut out of the document and run in a namespace that holds only what the reference's code has in scope at that
line, then their results are checked against the reference goldens / the oracle.

  seam-load, seam-db, seam-search, seam-call : the ctypes rebinding of database.py:12-32,166-195 (cpp/seqscore seam)
  ops-engine, ops-raw                        : Engine and the raw C-ABI form of the mel / encoder operators
  ops-imports, ops-matcher-loop              : pfann_amd.{model,melspec,musicdata,database} driven exactly as
                                               matcher.py:48-136 drives the reference's classes
"""
import json
import os
import re
import types

import numpy as np
import pytest

import make_golden as mg
from pfann_amd import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(os.path.dirname(__file__), "golden")


def snippets():
    text = open(os.path.join(REPO, "INTEGRATION.md"), encoding="utf8").read()
    out = {}
    for m in re.finditer(r"<!-- snippet: ([a-z0-9-]+) -->\s*```python\n(.*?)```", text, re.S):
        out[m.group(1)] = m.group(2)
    return out


def test_every_marked_snippet_is_covered_here():
    src = open(__file__, encoding="utf8").read()
    names = snippets()
    assert len(names) >= 8
    for n in names:
        assert '"%s"' % n in src, "INTEGRATION.md snippet %s has no test executing it" % n


def cfg(name):
    return json.load(open(os.path.join(REPO, "configs", name + ".json")))


# --------------------------------------------------------------------------- 1 + 2: the native seam, as written
SEAM_CASES = ["clean_hit", "negative_offset", "past_end", "k_gt_ntotal", "duplicate_songs", "frame_shift_mul2", "random_noisy"]


@pytest.mark.parametrize("name", SEAM_CASES)
def test_database_py_rebinding_as_written(tmp_path, monkeypatch, name):
    """INTEGRATION.md section 1 + 2 executed in the order a patched database.py would run them, on the reference's own golden
    cases (tests/golden/database.npz = outputs of the reference's Database.query_embeddings_base).  The native path's
    result equals the C restatement of cpp/seqscore.cpp, and -- on the cases where the reference's two paths agree --
    the golden (score, song, seconds)."""
    from oracle import native, search as osr, seqscore as osq
    z = np.load(os.path.join(G, "database.npz"))
    emb = np.ascontiguousarray(z[name + "_db"], np.float32)
    d = emb.shape[1]
    pos = osq.song_pos_from_key(z[name + "_key"])
    dir_for_db = str(tmp_path)
    emb.tofile(os.path.join(dir_for_db, "embeddings"))
    query = np.ascontiguousarray(z[name + "_q"], np.float32)
    fsm = int(z[name + "_fsm"])
    k = int(z[name + "_labels"].shape[1])
    self = types.SimpleNamespace(song_pos=pos, songList=["song%d" % i for i in range(pos.shape[0] - 1)], top_k=k,
                                 frame_shift_mul=fsm, score_alpha=0, hop_size=0.5)
    monkeypatch.chdir(REPO)                                    # the snippet loads 'pfann_amd/libpfann_amd.so'
    ns = {"np": np, "os": os, "self": self, "dir_for_db": dir_for_db, "d": d, "query": query}
    sn = snippets()
    exec(sn["seam-load"], ns)
    exec(sn["seam-db"], ns)
    exec(sn["seam-search"], ns)
    labels, distances = ns["labels"], ns["distances"]
    assert labels.dtype == np.int64 and labels.shape == (query.shape[0], k) and distances.dtype == np.float32
    Dr, Ir = osr.flat_ip_topk(query, emb, k)
    if fsm == 1 and name != "duplicate_songs":                 # duplicate rows tie: any of the tied labels is a valid top-k
        assert np.array_equal(np.sort(labels, 1), np.sort(Ir, 1))
    assert np.allclose(distances, Dr, atol=2e-6)
    # the golden's labels are what the reference searched with: score with exactly those
    ns["labels"] = np.ascontiguousarray(z[name + "_labels"], np.int64)
    exec(sn["seam-call"], ns)
    song_id, best, best_song_t, song_score = ns["song_id"], ns["best"], ns["best_song_t"], ns["song_score"]
    wbest, wss = native.seq_score(emb, pos, query, ns["labels"], fsm, 0.0)
    assert song_id == wbest
    assert np.array_equal(song_score[:, 1], wss[:, 1] * 0.5 / fsm)
    assert np.allclose(song_score[:, 0], wss[:, 0], atol=2e-6)
    # where cpp/seqscore and the python path agree, the golden decision
    assert song_id == int(z[name + "_song"])
    assert best_song_t[1] == float(z[name + "_sec"])
    assert abs(best - float(z[name + "_score"])) < 2e-6
    ns["mydll"].pfann_db_destroy.argtypes = [ns["c_void_p"]]
    ns["mydll"].pfann_db_destroy(self.amd_db)


def test_seam_is_safe_from_two_host_threads():
    """cpp/seqscore keeps no state, so the reference's seam may be called from several threads on one index; here the
    handle's scratch is guarded by a mutex: two threads hammering one handle with DIFFERENT queries always read back
    their own results."""
    import ctypes
    import threading
    from oracle import native, seqscore as osq
    from pfann_amd import lib as L
    from pfann_amd.database import DeviceIndex
    z = np.load(os.path.join(G, "database.npz"))
    db = np.ascontiguousarray(z["random_noisy_db"], np.float32)
    pos = osq.song_pos_from_key(z["random_noisy_key"])
    idx = DeviceIndex(db.shape[1], 0)
    idx.load(db, pos, 0)
    lib = L.load()
    q0 = np.ascontiguousarray(z["random_noisy_q"], np.float32)
    l0 = np.ascontiguousarray(z["random_noisy_labels"], np.int64)
    # a second, different query: rows of another song, labels from the oracle search
    from oracle import search as osr
    q1 = np.ascontiguousarray(db[int(pos[40]) + 2:int(pos[40]) + 9])
    _, l1 = osr.flat_ip_topk(q1, db, l0.shape[1])
    l1 = np.ascontiguousarray(l1, np.int64)
    want = [native.seq_score(db, pos, q, l, 1, 0.0) for q, l in ((q0, l0), (q1, l1))]
    assert want[0][0] != want[1][0]
    f32p, i64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
    bad = []

    def worker(which):
        q, l = (q0, l0) if which == 0 else (q1, l1)
        for _ in range(300):
            ss = np.zeros((pos.shape[0] - 1, 2), np.float32)
            b = lib.seq_score(idx.handle, pos.ctypes.data_as(i64p), pos.shape[0] - 1, q.ctypes.data_as(f32p), q.shape[0],
                              l.ctypes.data_as(i64p), l.shape[1], ss.ctypes.data_as(f32p), 1, 0.0)
            if b != want[which][0] or not np.array_equal(ss[:, 1], want[which][1][:, 1]) or \
                    not np.allclose(ss[:, 0], want[which][1][:, 0], atol=2e-6):
                bad.append((which, b))
    th = [threading.Thread(target=worker, args=(w,)) for w in (0, 1, 0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad[:5]


# ------------------------------------------------------------------------------- 3: the operator seams
def _model_dir(tmp_path, cfgname, seed=321):
    import shutil
    import torch
    params = cfg(cfgname)
    sd = synth.make_state_dict(params, seed=seed)
    mdir = tmp_path / ("model_" + cfgname)
    mdir.mkdir()
    torch.save({n: torch.from_numpy(v) for n, v in sd.items()}, str(mdir / "model.pt"))
    shutil.copy(os.path.join(REPO, "configs", cfgname + ".json"), str(mdir / "configs.json"))
    return params, sd, str(mdir)


def test_engine_snippet(tmp_path):
    """"ops-engine": Engine as section 3 shows it, every line against the oracle."""
    import torch
    from oracle import encoder as oe, melspec as om, segmenter as osg
    params, sd, mdir = _model_dir(tmp_path, "tiny")
    params["model_dir"] = mdir
    pcm = synth.make_song(7, seconds=5.0)
    wav = osg.pcm_to_mono(pcm[:, None])
    segs = osg.segment(wav, 8000, 4000)
    pcm44 = synth.make_song(8, seconds=2.0, sr=44100)
    ns = {"torch": torch, "os": os, "params": params, "batch": torch.from_numpy(segs), "wav": torch.from_numpy(wav),
          "hop": 4000, "pcm_int16": pcm44[:, None]}
    exec(snippets()["ops-engine"], ns)
    from pfann_amd.engine import mel_filterbank
    mel_ref = om.melspec(segs, params, mel_filterbank(8000, params["stft_n"], params["n_mels"], params["f_min"], params["f_max"]).numpy())
    emb_ref = oe.encode(om.melspec(segs, params), sd, params)       # embeddings: against the oracle's own float64 bank
    loud = mel_ref > mel_ref.max() - 11.5
    assert np.abs(ns["g"].cpu().numpy() - mel_ref)[loud].max() < 2e-3
    assert np.abs(ns["z"].cpu().numpy() - emb_ref).max() < 1e-4
    assert np.abs(ns["z_fused"].cpu().numpy() - emb_ref).max() < 1e-4
    from pfann_amd import resample as presample
    want44 = osg.pcm_to_mono(pcm44[:, None], 44100, 8000, resample_table=presample.filter_table(44100, 8000)[0])
    got44 = ns["wav44"].cpu().numpy()
    assert got44.shape == want44.shape and np.abs(got44 - want44).max() < 2e-6


def test_raw_c_abi_snippet():
    """"ops-raw": pfann_create / set_melbank / load_weight / melspec / encode through a bare ctypes.CDLL, default.json,
    against the reference's own golden embeddings (tests/golden/encoder_default.npz)."""
    import ctypes
    import torch
    from oracle import melspec as om
    from pfann_amd.engine import mel_filterbank
    from pfann_amd.lib import LIB_PATH
    params = cfg("default")
    z = np.load(os.path.join(G, "encoder_default.npz"))
    sd = synth.make_state_dict(params, seed=123)
    pcm = synth.make_song(5, seconds=3.0)
    from oracle import segmenter as osg
    segs_np = osg.segment(osg.pcm_to_mono(pcm[:, None]), 8000, 4000)
    B = segs_np.shape[0]
    ns = {"ctypes": ctypes, "np": np, "c_void_p": ctypes.c_void_p, "c_int": ctypes.c_int, "c_int64": ctypes.c_int64,
          "lib": ctypes.CDLL(LIB_PATH), "fb": mel_filterbank(8000, 1024, 256, 300, 4000).numpy(),
          "state_dict": {k: torch.from_numpy(v) for k, v in sd.items()},
          "segs": torch.from_numpy(segs_np).cuda(), "B": B,
          "mel_out": torch.empty((B, 256, 32), device="cuda"), "emb_out": torch.empty((B, 128), device="cuda")}
    exec(snippets()["ops-raw"], ns)
    torch.cuda.synchronize()
    from oracle import encoder as oe
    mel_ref = om.melspec(segs_np, params)
    assert np.abs(ns["emb_out"].cpu().numpy() - oe.encode(mel_ref, sd, params)).max() < 1e-4
    # and the golden input through the same raw context: the reference's own outputs
    x = torch.from_numpy(mg.encoder_inputs(256, 32)).cuda()
    out = torch.empty((x.shape[0], 128), device="cuda")
    assert ns["lib"].pfann_encode(ns["ctx"], x.data_ptr(), x.shape[0], out.data_ptr(), 1, None) == 0
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - z["emb"]).max() < 1e-4
    ns["lib"].pfann_destroy.argtypes = [ctypes.c_void_p]
    ns["lib"].pfann_destroy(ns["ctx"])


@pytest.mark.parametrize("cfgname", ["default", "seg", "n640d64"])
def test_fpnetwork_constructor_vs_reference_golden(cfgname):
    """FpNetwork(d, h, u, F, T, params['model']) exactly as builder.py:55 / matcher.py:60 construct it (no full config,
    no shared engine: the encoder-only context) against the reference's golden outputs, norm=True and norm=False; the
    torch.nn.Module-shaped calls the reference's scripts make on it (.to, .eval, .parameters) exist."""
    import torch
    from pfann_amd import lib as L
    from pfann_amd.model import FpNetwork
    z = np.load(os.path.join(G, "encoder_%s.npz" % cfgname))
    params = json.loads(str(z["params"]))
    d, h, u, F, T = synth.model_dims(params)
    model = FpNetwork(d, h, u, F, T, params["model"]).to(torch.device("cuda"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(params, seed=123).items()})
    model.eval()
    for param in model.parameters():
        param.requires_grad = False
    x = torch.from_numpy(mg.encoder_inputs(F, T)).cuda()
    assert np.abs(model(x).cpu().numpy() - z["emb"]).max() < 1e-4
    raw = model.forward(x, norm=False).cpu().numpy()
    assert np.abs(raw - z["raw"]).max() < 1e-4 * max(1.0, np.abs(z["raw"]).max())
    with pytest.raises(L.PfannError):                          # an encoder-only context refuses front-end work
        model.engine.melspec(torch.zeros(1, model.engine.seg_len))


@pytest.mark.parametrize("cfgname", ["default", "seg", "n640d64"])
def test_matcher_loop_on_the_mirror_classes(tmp_path, cfgname):
    """"ops-imports" + "ops-matcher-loop": the reference's matcher loop (matcher.py:48-136) on pfann_amd's classes, against
    a database directory written from the ORACLE's embeddings; answers equal the oracle pipeline's decisions
    (reference python path, tests/golden-pinned), scores within 1e-4 (embedding tolerance)."""
    import torch
    from oracle import encoder as oe, melspec as om, search as osr, segmenter as osg, seqscore as osq
    params, sd, mdir = _model_dir(tmp_path, cfgname)
    dmodel, k = params["model"]["d"], params["indexer"]["top_k"]
    n_songs = 6
    songs, music = {}, []
    for s in range(n_songs):
        path = str(tmp_path / ("song%d.wav" % s))
        songs[s] = synth.make_song(200 + s, seconds=7.0 + s)
        synth.write_wav(path, songs[s] if s != 2 else np.stack([songs[s], songs[s] // 3], 1))
        music.append(path)
    db_dir = tmp_path / "db"
    db_dir.mkdir()
    keys, embs = [], []
    for p in music:
        segs = osg.load_segments(p, dict(params, indexer=dict(params["indexer"], frame_shift_mul=1)))
        keys.append(segs.shape[0])
        embs.append(oe.encode(om.melspec(segs, params), sd, params))
    db_emb = np.concatenate(embs).astype(np.float32)
    db_emb.tofile(str(db_dir / "embeddings"))
    np.asarray(keys, np.int32).tofile(str(db_dir / "landmarkKey"))
    (db_dir / "songList.txt").write_text("".join(p + "\n" for p in music))
    torch.save({n: torch.from_numpy(v) for n, v in sd.items()}, str(db_dir / "model.pt"))
    json.dump(params, open(str(db_dir / "configs.json"), "w"))
    queries = []
    for j, s in enumerate([0, 1, 3, 4, 5]):
        q, _ = synth.make_query(songs[s], 50 + j, 4.0, snr_db=10.0)
        path = str(tmp_path / ("q%d.wav" % j))
        synth.write_wav(path, q)
        queries.append(path)
    queries.append(str(tmp_path / "missing.wav"))
    qlist = tmp_path / "q.txt"
    qlist.write_text("".join(p + "\n" for p in queries))

    ns = {"torch": torch, "os": os, "np": np, "params": params, "dir_for_db": str(db_dir), "file_list_for_query": str(qlist)}
    sn = snippets()
    exec(sn["ops-imports"], ns)
    exec(sn["ops-matcher-loop"], ns)
    answers = ns["answers"]
    assert len(answers) == len(queries)
    pos = osq.song_pos_from_key(keys)
    fsm = params["indexer"].get("frame_shift_mul", 1)
    for (name, ans, sco, tim), qp in zip(answers, queries):
        assert name == qp
        segs = osg.load_segments(qp, params)
        if segs.shape[0] == 0:
            assert ans == "error" and sco == -1e999 and tim == 0
            continue
        e = oe.encode(om.melspec(segs, params), sd, params)
        _, I = osr.flat_ip_topk(e, db_emb, k)
        w_sco, (w_song, w_tim), _ = osq.query_embeddings_base(e, I, db_emb, pos, params["hop_size"], fsm)
        assert ans == music[w_song] and tim == w_tim, (qp, ans, tim, w_song, w_tim)
        assert abs(sco - w_sco) < 1e-4


def test_database_under_ranks_snippet(tmp_path):
    """"ranks-database": INTEGRATION.md's Database-under-ranks script, as written, under torch.distributed.run with 2 ranks
    (gloo-staged collectives on this box's one GPU): every rank gets the single-process answer for a query cut out of
    the database, with the whole [n_songs, 2] score block."""
    import subprocess
    import sys
    import torch
    params = cfg("tiny")
    d = params["model"]["d"]
    key = [7, 0, 12, 9, 5, 11]
    pos = np.concatenate([[0], np.cumsum(key)])
    emb = synth.unit_rows(77, "ranks/db", int(pos[-1]), d).astype(np.float32)
    db_dir = tmp_path / "db"
    db_dir.mkdir()
    emb.tofile(str(db_dir / "embeddings"))
    np.asarray(key, np.int32).tofile(str(db_dir / "landmarkKey"))
    (db_dir / "songList.txt").write_text("".join("song%d.wav\n" % i for i in range(len(key))))
    json.dump(params, open(str(db_dir / "configs.json"), "w"))
    q = emb[pos[3] + 2: pos[3] + 7]
    np.save(str(tmp_path / "q.npy"), q)
    head = ("import os, sys, json\nimport numpy as np\nsys.path.insert(0, %r)\ndir_for_db = %r\n"
            "params = json.load(open(os.path.join(dir_for_db, 'configs.json')))\nembeddings = np.load(%r)\n"
            % (REPO, str(db_dir), str(tmp_path / "q.npy")))
    tail = ("\nassert ans == 3 and tim == 2 * params['hop_size'] and abs(sco - 1.0) < 1e-5, (ans, tim, sco)\n"
            "assert hi - lo < len(db.songList) and song_score[3, 0] > 0.999 and song_score[1].sum() == 0\nprint('RANK_OK')\n")
    script = tmp_path / "ranks_db.py"
    script.write_text(head + snippets()["ranks-database"] + tail)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PFANN_GPUS")}
    env.update(PFANN_DIST_BACKEND="gloo", PFANN_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29771", str(script)], capture_output=True, text=True, timeout=600,
                       env=env, cwd=str(tmp_path))
    assert r.returncode == 0 and r.stdout.count("RANK_OK") == 2, r.stdout[-2000:] + r.stderr[-3000:]
    assert torch.cuda.is_available()
