"""The song-sharded multi-GPU path behind the product's own entry points (reference
counterpart: matcher.py:65,136 + the latent index_cpu_to_all_gpus at database.py:101-104): builder.py / matcher.py /
extractemb.py / matchemb.py and `Database` run as N ranks (one process per GPU; here 2 ranks sharing this box's one
GPU over gloo, and 1 rank through RCCL) and write files BYTE-IDENTICAL to the single-process run."""
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
DROP = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PFANN_DIST_BACKEND", "PFANN_FORCE_DEVICE", "PFANN_GPUS", "PFANN_FORCE_SHARDED",
        "PFANN_EXCHANGE_STREAM")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in DROP}
    e.update(PYTHONPATH=REPO, MASTER_ADDR="127.0.0.1")
    e.update(kw)
    return e


def _run(cmd, cwd, env, timeout=900):
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=cwd, timeout=timeout)
    assert r.returncode == 0, " ".join(cmd) + "\n" + r.stdout[-3000:] + r.stderr[-4000:]
    return r.stdout


def _tool(name):
    return [sys.executable, os.path.join(REPO, name)]


def _two_ranks(**kw):                       # 2 ranks on this box's one GPU: the tool starts them itself (PFANN_GPUS)
    return _env(PFANN_GPUS="2", PFANN_DIST_BACKEND="gloo", PFANN_FORCE_DEVICE="0", **kw)


def _same_file(a, b, what):
    x, y = open(a, "rb").read(), open(b, "rb").read()
    assert len(x) == len(y), "%s: %d vs %d bytes" % (what, len(x), len(y))
    if x != y:
        diff = np.nonzero(np.frombuffer(x, np.uint8) != np.frombuffer(y, np.uint8))[0]
        raise AssertionError("%s differs in %d bytes, first at %d" % (what, diff.size, diff[0]))


def _model_dir(tmp, cfgname, seed, patch=None):
    import torch
    params = json.load(open(os.path.join(REPO, "configs", cfgname + ".json")))
    if patch:
        patch(params)
    sd = synth.make_state_dict_calibrated(params, seed=seed) if cfgname == "default" else synth.make_state_dict(params, seed=seed)
    mdir = os.path.join(tmp, "model")
    os.makedirs(mdir, exist_ok=True)
    torch.save({n: torch.from_numpy(np.asarray(v)) for n, v in sd.items()}, os.path.join(mdir, "model.pt"))
    json.dump(params, open(os.path.join(mdir, "configs.json"), "w"))
    return mdir, params


def _db_files_equal(a, b):
    for f in ("embeddings", "landmarkValue", "landmarkKey", "songList.txt", "configs.json", "model.pt"):
        _same_file(os.path.join(a, f), os.path.join(b, f), "db/" + f)


def _results_equal(a, b):
    _same_file(a, b, "result TSV")
    _same_file(a + ".bin", b + ".bin", "result .bin")
    _same_file(os.path.splitext(a)[0] + "_detail.csv", os.path.splitext(b)[0] + "_detail.csv", "_detail.csv")


def test_cfg1_set_two_ranks_byte_identical(tmp_path):
    """BASELINE config 1's shape (10 songs incl. an unreadable one, a stereo one, one at 16 kHz; clean + noisy crops, a
    too-short query, a missing file): builder / matcher / extractemb + matchemb under 2 ranks == the 1-process files."""
    tmp = str(tmp_path)
    mdir, params = _model_dir(tmp, "default", 123)
    music, songs = [], {}
    for s in range(10):
        path = os.path.join(tmp, "song%02d.wav" % s)
        if s == 4:
            open(path, "wb").write(b"garbage")
        else:
            pcm = synth.make_song(100 + s, seconds=8.0 + s)
            if s == 6:
                pcm = np.stack([pcm, pcm // 2], 1)
            synth.write_wav(path, pcm, sr=16000 if s == 8 else 8000)
            songs[s] = pcm
        music.append(path)
    mlist = os.path.join(tmp, "music.txt")
    open(mlist, "w").write("".join(p + "\n" for p in music))
    queries = []
    for j in range(8):
        s = [0, 1, 2, 3, 5, 6, 7, 9][j]
        src = songs[s] if songs[s].ndim == 1 else songs[s][:, 0]
        q, _ = synth.make_query(src, j, 4.0, snr_db=6.0 if j % 2 else 30.0)
        path = os.path.join(tmp, "q%02d.wav" % j)
        synth.write_wav(path, q)
        queries.append(path)
    short = os.path.join(tmp, "qshort.wav")
    synth.write_wav(short, songs[2][16000:16000 + 3000])
    queries += [short, os.path.join(tmp, "qmissing.wav")]
    qlist = os.path.join(tmp, "queries.txt")
    open(qlist, "w").write("".join(p + "\n" for p in queries))

    db1, db2 = os.path.join(tmp, "db1"), os.path.join(tmp, "db2")
    _run(_tool("builder.py") + [mlist, db1, mdir], tmp, _env())
    _run(_tool("builder.py") + [mlist, db2, mdir], tmp, _two_ranks())
    _db_files_equal(db1, db2)
    key = np.fromfile(os.path.join(db1, "landmarkKey"), np.int32)
    assert key[4] == 0 and key.sum() > 100
    r1, r2, r3 = (os.path.join(tmp, n) for n in ("r1.txt", "r2.txt", "r3.txt"))
    _run(_tool("matcher.py") + [qlist, db1, r1], tmp, _env())
    out = _run(_tool("matcher.py") + [qlist, db1, r2], tmp, _two_ranks())
    assert out.count("total query time") == 1                 # rank 0 alone speaks
    _results_equal(r1, r2)
    rows = [ln.rstrip("\n").split("\t") for ln in open(r2)]
    assert rows[-1][1] == "error" and rows[-2][1] in music and len(rows) == 10     # short query: padded to one segment
    # the other two tools, 2 ranks: extractemb's files equal the 1-process ones, matchemb reproduces the matcher
    e1, e2 = os.path.join(tmp, "e1"), os.path.join(tmp, "e2")
    _run(_tool("extractemb.py") + [qlist, db1, e1], tmp, _env())
    _run(_tool("extractemb.py") + [qlist, db1, e2], tmp, _two_ranks())
    for f in ("query_embeddings", "query_index", "queryList.txt", "configs.json"):
        _same_file(os.path.join(e1, f), os.path.join(e2, f), "extractemb/" + f)
    _run(_tool("matchemb.py") + [e2, db1, r3], tmp, _two_ranks())
    _results_equal(r1, r3)


def _write_set(tmp, n_songs, n_queries, bad_songs=(), bad_queries=()):
    """n_songs seeded 30 s songs + n_queries 10 s SNR-0 crops as WAV files (synthesised on the GPU)."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import cli_bench
    dev = torch.device("cuda", 0)
    sdir, qdir = os.path.join(tmp, "songs"), os.path.join(tmp, "queries")
    os.makedirs(sdir), os.makedirs(qdir)
    music = [os.path.join(sdir, "s%05d.wav" % i) for i in range(n_songs)]
    for c0 in range(0, n_songs, 200):
        ids = list(range(c0, min(c0 + 200, n_songs)))
        cli_bench.write_wavs(music[c0:c0 + len(ids)], synth.make_songs_torch(ids, 30.0, device=dev).cpu().numpy())
    for b in bad_songs:
        open(music[b], "wb").write(b"RIFFxxxxWAVEjunk")
    q_song = [int((j * 7919 + 13) % n_songs) for j in range(n_queries)]
    queries = [os.path.join(qdir, "q%05d.wav" % j) for j in range(n_queries)]
    for c0 in range(0, n_queries, 200):
        c1 = min(c0 + 200, n_queries)
        qp, _ = synth.make_queries_torch(synth.make_songs_torch(q_song[c0:c1], 30.0, device=dev), list(range(c0, c1)), 10.0, 0.0)
        cli_bench.write_wavs(queries[c0:c1], qp.cpu().numpy())
    for b in bad_queries:
        os.remove(queries[b])
    mlist, qlist = os.path.join(tmp, "music.txt"), os.path.join(tmp, "queries.txt")
    open(mlist, "w").write("".join(p + "\n" for p in music))
    open(qlist, "w").write("".join(p + "\n" for p in queries))
    torch.cuda.synchronize()
    return mlist, qlist, music, q_song


def test_600_songs_two_ranks_byte_identical_and_rccl_world1(tmp_path):
    """600 songs (35 k segments; two unreadable files = 0-segment songs, one at each end of a shard), 300 queries (two
    missing = error rows): several rounds of launch groups per tool (PFANN_MAX_BATCH=4096).  2 ranks (gloo, one GPU) and
    1 rank through RCCL with PFANN_FORCE_SHARDED=1 (every collective of the sharded path really executes on the nccl
    backend) against the plain single-process run: every output file byte for byte."""
    tmp = str(tmp_path)
    mdir, params = _model_dir(tmp, "default", 123, patch=lambda p: p["indexer"].update(index_factory="Flat"))
    mlist, qlist, music, q_song = _write_set(tmp, 600, 300, bad_songs=(0, 299), bad_queries=(7, 299))
    common = dict(PFANN_MAX_BATCH="4096")
    db1, db2 = os.path.join(tmp, "db1"), os.path.join(tmp, "db2")
    _run(_tool("builder.py") + [mlist, db1, mdir], tmp, _env(**common))
    _run(_tool("builder.py") + [mlist, db2, mdir], tmp, _two_ranks(**common))
    _db_files_equal(db1, db2)
    key = np.fromfile(os.path.join(db1, "landmarkKey"), np.int32)
    assert key[0] == 0 and key[299] == 0 and key.sum() == 598 * 59
    # another grouping of the same list (smaller launch groups, so a different cut) still gives the same bytes ONLY
    # because the plan is pinned to the group size: same PFANN_MAX_BATCH everywhere here
    r1, r2, r3 = (os.path.join(tmp, n) for n in ("r1.txt", "r2.txt", "r3.txt"))
    _run(_tool("matcher.py") + [qlist, db1, r1], tmp, _env(**common))
    _run(_tool("matcher.py") + [qlist, db1, r2], tmp, _two_ranks(**common))
    _results_equal(r1, r2)
    rows = [ln.rstrip("\n").split("\t") for ln in open(r1)]
    assert rows[7][1] == "error" and rows[299][1] == "error"
    hit = sum(1 for j, (_, ans) in enumerate(rows) if ans == music[q_song[j]])
    assert hit >= 0.9 * 298, hit
    blocks = np.fromfile(r1 + ".bin", np.float32).reshape(300, 600, 2)
    assert not blocks[7].any() and not blocks[299].any() and blocks[0].any()
    # one rank, RCCL, the whole exchange protocol forced
    env = _env(PFANN_FORCE_SHARDED="1", **common)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29761", os.path.join(REPO, "matcher.py"), qlist, db1, r3]
    _run(cmd, tmp, env)
    _results_equal(r1, r3)
    # PFANN_EXCHANGE_STREAM=1 (round 5): search, collectives, matcher and winner pick of a launch group on a stream of their
    # own (the next group's encoder then runs beside them) -- flag on == flag off, under 2 gloo ranks and through RCCL
    r4, r5 = os.path.join(tmp, "r4.txt"), os.path.join(tmp, "r5.txt")
    _run(_tool("matcher.py") + [qlist, db1, r4], tmp, _two_ranks(PFANN_EXCHANGE_STREAM="1", **common))
    _results_equal(r1, r4)
    _run(cmd[:-1] + [r5], tmp, dict(env, PFANN_EXCHANGE_STREAM="1"))
    _results_equal(r1, r5)
    # Database's per-query contract under ranks: the reference's tuple, whole [n_songs, 2] block, on every rank
    worker = os.path.join(tmp, "dbworker.py")
    open(worker, "w").write(DB_WORKER)
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", "29763", worker, REPO, db1, r1 + ".bin"], tmp,
               _env(PFANN_DIST_BACKEND="gloo", PFANN_FORCE_DEVICE="0"))
    assert out.count("DB_RANK_OK") == 2


DB_WORKER = r'''
import os, sys, json
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from pfann_amd.database import Database
from pfann_amd.dist import init_ranks, finish_ranks
ranks = init_ranks()
db_dir = sys.argv[2]
params = json.load(open(os.path.join(db_dir, "configs.json")))
db = Database(db_dir, params["indexer"], params["hop_size"], d=params["model"]["d"], ranks=ranks)
assert db.index.ntotal < 598 * 59 and db.song_range[1] - db.song_range[0] < 600       # a shard, not the database
emb = np.fromfile(os.path.join(db_dir, "embeddings"), np.float32).reshape(-1, 128)
key = np.fromfile(os.path.join(db_dir, "landmarkKey"), np.int32)
pos = np.concatenate([[0], np.cumsum(key)])
for song in (1, 298, 300, 599):                      # songs of both shards, next to the 0-segment songs
    q = emb[pos[song] + 5: pos[song] + 24]
    score, (sid, sec), ss = db.query_embeddings(q)
    assert sid == song and sec == 2.5 and abs(score - 1.0) < 1e-5, (song, sid, sec, score)
    assert ss.shape == (600, 2) and abs(ss[song, 0] - 1.0) < 1e-5 and ss[song, 1] == 2.5
    assert ss[0].sum() == 0 and ss[299].sum() == 0
print("DB_RANK_OK")
finish_ranks(ranks)
'''


def test_edge_lists_and_variants_under_two_ranks(tmp_path):
    """Lists the round cutting must not trip over -- empty, one file, fewer files than ranks, nothing but unreadable
    files -- and the matcher variants (fp16-only storage + frame_shift_mul = 2, tiny d = 16 model) under 2 ranks: every
    output byte-identical to the single-process run; a one-song database built by 2 ranks likewise."""
    tmp = str(tmp_path)

    def patch(p):
        p["indexer"].update(frame_shift_mul=2, use_float16=True)
    mdir, params = _model_dir(tmp, "tiny", 9, patch=patch)
    music = []
    for s in range(5):
        path = os.path.join(tmp, "m%d.wav" % s)
        synth.write_wav(path, synth.make_song(500 + s, seconds=7.0 + 2 * s))
        music.append(path)
    lists = {"all": music, "one": music[:1]}
    for name, files in lists.items():
        open(os.path.join(tmp, name + ".txt"), "w").write("".join(p + "\n" for p in files))
    dbs = {}
    for name in lists:
        for tag, env in (("a", _env()), ("b", _two_ranks())):
            dbs[name + tag] = os.path.join(tmp, "db_" + name + tag)
            _run(_tool("builder.py") + [os.path.join(tmp, name + ".txt"), dbs[name + tag], mdir], tmp, env)
        _db_files_equal(dbs[name + "a"], dbs[name + "b"])
    qs = []
    for j in range(7):
        q, _ = synth.make_query(synth.make_song(500 + j % 5, seconds=7.0 + 2 * (j % 5)), 60 + j, 3.0 + j % 3, snr_db=12.0)
        path = os.path.join(tmp, "q%d.wav" % j)
        synth.write_wav(path, q)
        qs.append(path)
    cases = {"empty": [], "single": qs[:1], "missing_only": [os.path.join(tmp, "no1.wav"), os.path.join(tmp, "no2.wav")],
             "seven": qs[:3] + [os.path.join(tmp, "no3.wav")] + qs[3:]}
    for name, files in cases.items():
        ql = os.path.join(tmp, "ql_%s.txt" % name)
        open(ql, "w").write("".join(p + "\n" for p in files))
        r1, r2 = os.path.join(tmp, "r1_%s.txt" % name), os.path.join(tmp, "r2_%s.txt" % name)
        _run(_tool("matcher.py") + [ql, dbs["alla"], r1], tmp, _env())
        _run(_tool("matcher.py") + [ql, dbs["alla"], r2], tmp, _two_ranks())
        _results_equal(r1, r2)
        assert os.path.getsize(r1 + ".bin") == len(files) * 5 * 8
        assert len(open(r1).read().splitlines()) == len(files)
    # a tiny score-block budget: every launch group is split into several sub-launches, which are COLLECTIVE under ranks
    # -- the cut must be the same on every rank although the shards are 2 and 3 songs wide (13 pairs: 4 queries per launch)
    ql = os.path.join(tmp, "ql_seven.txt")
    r3, r4 = os.path.join(tmp, "r3_seven.txt"), os.path.join(tmp, "r4_seven.txt")
    _run(_tool("matcher.py") + [ql, dbs["alla"], r3], tmp, _env(PFANN_SCORE_BLOCK_MB="0.0001"))
    _run(_tool("matcher.py") + [ql, dbs["alla"], r4], tmp, _two_ranks(PFANN_SCORE_BLOCK_MB="0.0001"), timeout=300)
    _results_equal(os.path.join(tmp, "r1_seven.txt"), r3)
    _results_equal(r3, r4)
    rows = [ln.rstrip("\n").split("\t")[1] for ln in open(os.path.join(tmp, "r2_seven.txt"))]
    assert rows[3] == "error" and sum(1 for j, a in enumerate(rows) if a in music) == 7
    # the one-song database (its single song lives on one rank, the other shard is empty)
    r1, r2 = os.path.join(tmp, "o1.txt"), os.path.join(tmp, "o2.txt")
    ql = os.path.join(tmp, "ql_seven.txt")
    _run(_tool("matcher.py") + [ql, dbs["onea"], r1], tmp, _env())
    _run(_tool("matcher.py") + [ql, dbs["onea"], r2], tmp, _two_ranks())
    _results_equal(r1, r2)


def test_eight_ranks_on_one_gpu_byte_identical(tmp_path):
    """World size 8 -- BASELINE config 4's "sharded 8 ways" -- as far as a one-GPU box allows: eight ranks of builder.py
    and matcher.py share this GPU (gloo-staged collectives, PFANN_MAX_BATCH=2048 so that eight workspaces fit), which
    exercises rounds of eight launch groups, the equal-share tail, eight-way song shards, the all-to-all / merge / key
    pick with G = 8 and eight writers of one `.bin` -- against the single-process files, byte for byte."""
    tmp = str(tmp_path)
    mdir, params = _model_dir(tmp, "default", 123, patch=lambda p: p["indexer"].update(index_factory="Flat"))
    mlist, qlist, music, q_song = _write_set(tmp, 300, 120, bad_songs=(150,), bad_queries=(60,))
    common = dict(PFANN_MAX_BATCH="2048")
    eight = _env(PFANN_GPUS="8", PFANN_DIST_BACKEND="gloo", PFANN_FORCE_DEVICE="0", **common)
    db1, db8 = os.path.join(tmp, "db1"), os.path.join(tmp, "db8")
    _run(_tool("builder.py") + [mlist, db1, mdir], tmp, _env(**common))
    _run(_tool("builder.py") + [mlist, db8, mdir], tmp, eight, timeout=1500)
    _db_files_equal(db1, db8)
    r1, r8 = os.path.join(tmp, "r1.txt"), os.path.join(tmp, "r8.txt")
    _run(_tool("matcher.py") + [qlist, db1, r1], tmp, _env(**common))
    _run(_tool("matcher.py") + [qlist, db8, r8], tmp, eight, timeout=1500)
    _results_equal(r1, r8)
    rows = [ln.rstrip("\n").split("\t") for ln in open(r8)]
    assert rows[60][1] == "error" and sum(1 for j, (_, a) in enumerate(rows) if a == music[q_song[j]]) >= 100
