"""Generate tests/golden/*.npz by running the REFERENCE's own Python in this container.

Run once from the repo root (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_golden.py

What is executed verbatim from /root/reference:
  * model.py            (FpNetwork)                         -> encoder_*.npz
  * datautil/musicdata.py (MusicDataset)                    -> segmenter.npz
  * database.py         (Database.query_embeddings_base)    -> database.npz

Two third-party modules those files import are not installed (no network):
  * `julius`  - musicdata.py:29 builds julius.ResampleFrac(sr_in, sr_out).  All fixture
    WAVs are already at the model rate, where a fractional resampler is the identity,
    so an identity module is injected.  Non-8 kHz input stays out of scope / unpinned.
  * `faiss`   - database.py:88-99,121,155 use read_index / search / reconstruct on the
    index.  An exact flat inner-product index (the definition of IndexFlatIP) is injected.
    These fixtures therefore pin database.py's OWN logic (candidate generation, averaging,
    strict-> tie-breaks, the -1 / empty-db quirks), not faiss.
datautil/melspec.py cannot be imported (torchaudio absent): the mel stage has no
reference-generated fixture and is declared "parity unpinned" (DESIGN.md).

Inputs are regenerated from integer seeds by pfann_amd.synth at test time; only expected
outputs (and small inputs) are stored.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from pfann_amd import synth  # noqa: E402


# ----------------------------------------------------------------- third-party shims
def _install_julius():
    m = types.ModuleType("julius")

    class ResampleFrac:
        def __init__(self, old_sr, new_sr):
            assert old_sr == new_sr, "fixtures are generated at the model rate only"

        def __call__(self, x):
            return x
    m.ResampleFrac = ResampleFrac
    sys.modules["julius"] = m


class _FlatIP:
    """Exact inner-product flat index: the definition of faiss.IndexFlatIP."""

    def __init__(self, xb):
        self.xb = np.ascontiguousarray(xb, dtype=np.float32)
        self.d = self.xb.shape[1]
        self.ntotal = self.xb.shape[0]

    def search(self, q, k):
        D = np.full((q.shape[0], k), -np.finfo(np.float32).max, np.float32)
        I = np.full((q.shape[0], k), -1, np.int64)
        if self.ntotal:
            s = q @ self.xb.T
            o = np.argsort(-s, axis=1, kind="stable")[:, :k]
            D[:, :o.shape[1]] = np.take_along_axis(s, o, 1)
            I[:, :o.shape[1]] = o
        return D, I

    def reconstruct(self, i, out=None):
        if out is None:
            return self.xb[i].copy()
        out[:] = self.xb[i]
        return out


def _install_faiss(store):
    m = types.ModuleType("faiss")
    for name in ("Index", "IndexBinary", "IndexPreTransform", "IndexIVF"):
        setattr(m, name, type(name, (), {}))
    m.IndexFlat = _FlatIP
    m.downcast_index = lambda x: x
    m.read_index = lambda path: store[path]
    sys.modules["faiss"] = m


# ------------------------------------------------------------------------- encoder
def encoder_inputs(F, T, B=3):
    """Deterministic log-mel-like inputs (regenerated in tests): range [-17, 2]."""
    u = synth.uniform01(777, "golden/mel", B * F * T).reshape(B, F, T)
    x = (-17.0 + 19.0 * u).astype(np.float32)
    x[B - 1] = synth.uniform01(778, "golden/mock", F * T).reshape(F, T)  # mock_data.py-like [0,1)
    return x


ENCODER_CASES = {
    "default": "configs/default.json",
    "seg": "configs/seg.json",
    "n640d64": "configs/n640d64.json",
    "tiny": "configs/tiny.json",
}


def naf_style_params():
    p = json.load(open(os.path.join(REPO, "configs/tiny.json")))
    p["model"].update(conv_activation="ELU", relu_after_bn=False, fuller=False)
    return p


def elu_full_params():
    """full conv2 (fused-LayerNorm capable) with ELU and activation-before-LN."""
    p = json.load(open(os.path.join(REPO, "configs/tiny.json")))
    p["model"].update(conv_activation="ELU", relu_after_bn=False, fuller=True)
    return p


def strides_pow2_params():
    """Per-block strides (model.py:83-85) as tools/convert_naf_to_pfann.py:82-110 writes them for a converted
    model: ELU, activation before LN, full conv2; a stride-1 conv1 in block 1 (pad (1,1) with T > 1) and
    early size-1 axes.  Every Fo*To stays a power of two, so the LayerNorm-fused GEMM path applies."""
    p = json.load(open(os.path.join(REPO, "configs/tiny.json")))
    st = [2, 1, 2, 2, 2, 2, 1, 1]
    p["model"].update(conv_activation="ELU", relu_after_bn=False, fuller=True,
                      strides=[[[1, st[i]], [2, 1]] for i in range(8)])
    return p


def strides_np2_params():
    """Stride 3 along F in block 0 (256 -> 86 -> 43 -> ...): Fo*To is NOT a power of two, which the fused
    path does not support -- the library must fall back to the separate-LayerNorm kernels and say so."""
    p = json.load(open(os.path.join(REPO, "configs/tiny.json")))
    sf = [3, 2, 2, 2, 2, 2, 2, 2]
    st = [2, 2, 2, 2, 2, 1, 1, 1]
    p["model"].update(fuller=True, strides=[[[1, st[i]], [sf[i], 1]] for i in range(8)])
    return p


def gen_encoder():
    from model import FpNetwork
    cases = {k: json.load(open(os.path.join(REPO, v))) for k, v in ENCODER_CASES.items()}
    cases["nafstyle"] = naf_style_params()
    cases["elu_full"] = elu_full_params()
    cases["strides_pow2"] = strides_pow2_params()
    cases["strides_np2"] = strides_np2_params()
    for name, params in cases.items():
        d, h, u, F, T = synth.model_dims(params)
        net = FpNetwork(d, h, u, F, T, params["model"])
        sd = synth.make_state_dict(params, seed=123)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        net.eval()
        x = torch.from_numpy(encoder_inputs(F, T))
        taps = []
        hooks = []
        for blk in net.f.convs:
            mods = (blk.relu1, blk.relu2) if blk.relu_after_bn else (blk.ln1, blk.ln2)
            for mod in mods:
                hooks.append(mod.register_forward_hook(lambda m, i, o: taps.append(o.detach().double())))
        with torch.no_grad():
            emb = net(x).numpy()
            for hk in hooks:
                hk.remove()
            raw = net.forward(x, norm=False).numpy()
        sums = np.array([[t.sum().item(), t.abs().sum().item()] for t in taps])
        np.savez(os.path.join(OUT, "encoder_%s.npz" % name), params=json.dumps(params),
                 emb=emb, raw=raw, tap_sums=sums)
        print("encoder", name, emb.shape, np.linalg.norm(emb, axis=1))


# ----------------------------------------------------------------------- segmenter
def segmenter_inputs():
    """name -> int16 [n] or [n,2]; regenerated identically in tests."""
    n70 = 70 * 8000
    base = synth.make_song(7, seconds=70.0)
    other = synth.make_song(8, seconds=12.0)
    stereo = np.stack([other, (other.astype(np.int32) * 3 // 4 + synth.make_song(9, 12.0) // 4).astype(np.int16)], 1)
    fake = np.stack([other, (-other.astype(np.int32)).clip(-32768, 32767).astype(np.int16)], 1)
    return {"mono70": base[:n70], "stereo12": stereo, "fakestereo12": fake,
            "short": other[:3000], "exact1s": other[:8000], "len_8001": other[:8001]}


def gen_segmenter():
    _install_julius()
    from datautil.musicdata import MusicDataset
    params = json.load(open(os.path.join(REPO, "configs/default.json")))
    out = {}
    with tempfile.TemporaryDirectory() as td:
        names = []
        for name, pcm in segmenter_inputs().items():
            synth.write_wav(os.path.join(td, name + ".wav"), pcm)
            names.append(os.path.join(td, name + ".wav"))
        names.append(os.path.join(td, "missing.wav"))
        with open(os.path.join(td, "notwav.wav"), "wb") as f:
            f.write(b"this is not a wave file")
        names.append(os.path.join(td, "notwav.wav"))
        lst = os.path.join(td, "list.txt")
        open(lst, "w").write("\n".join(names) + "\n")
        for fsm in (1, 2):
            p = json.loads(json.dumps(params))
            p["indexer"]["frame_shift_mul"] = fsm
            ds = MusicDataset(lst, p)
            for i in range(len(ds)):
                _, path, wav = ds[i]
                key = "%s_fsm%d" % (os.path.splitext(os.path.basename(path))[0], fsm)
                w = wav.numpy()
                out[key + "_shape"] = np.array(w.shape)
                if w.shape[0]:
                    pick = sorted(set([0, w.shape[0] // 2, w.shape[0] - 1, min(119, w.shape[0] - 1)]))
                    out[key + "_rows"] = np.array(pick)
                    out[key + "_vals"] = w[pick]
                    out[key + "_sums"] = np.stack([w.astype(np.float64).sum(1), np.abs(w.astype(np.float64)).sum(1)], 1)
                print("segmenter", key, w.shape)
    np.savez_compressed(os.path.join(OUT, "segmenter.npz"), **out)


# ------------------------------------------------------------------------ database
def database_cases():
    """Toy db: d=16, songs of 10/8/0/12 segments, hop 0.5 s."""
    d = 16
    key = [10, 8, 0, 12]
    db = synth.unit_rows(5, "golden/db", sum(key), d)
    pos = np.pad(np.cumsum(key), (1, 0))
    cases = {}
    cases["clean_hit"] = dict(db=db, key=key, q=db[pos[3] + 4: pos[3] + 9], k=5)
    qa = np.concatenate([synth.unit_rows(6, "golden/pre", 2, d), db[pos[1]: pos[1] + 4]])
    cases["negative_offset"] = dict(db=db, key=key, q=qa, k=5)
    qb = np.concatenate([db[pos[0] + 7: pos[0] + 10], synth.unit_rows(7, "golden/post", 3, d)])
    cases["past_end"] = dict(db=db, key=key, q=qb, k=5)
    cases["k_gt_ntotal"] = dict(db=db, key=key, q=db[pos[3] + 1: pos[3] + 4], k=40)
    dup = np.concatenate([db[:10], db[:10], db[18:]])
    cases["duplicate_songs"] = dict(db=dup, key=[10, 10, 0, 12], q=dup[2:6], k=6)
    cases["nonpositive_best"] = dict(db=db, key=key, q=-db[pos[3] + 4: pos[3] + 5], k=1,
                                     force_labels=int(pos[3] + 4))
    cases["no_candidates"] = dict(db=db, key=key, q=db[3:6], k=4, force_labels=-1)
    q2 = db[pos[3] + 1: pos[3] + 7].copy()
    q2 = np.repeat(q2, 2, axis=0)[:11] * 0.9 + 0.1 * synth.unit_rows(8, "golden/fsm", 11, d)
    q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    cases["frame_shift_mul2"] = dict(db=db, key=key, q=q2.astype(np.float32), k=5, fsm=2)
    cases["empty_db"] = dict(db=np.zeros((0, d), np.float32), key=[0, 0], q=db[:3], k=3)
    # a larger random case: 40 songs, noisy 19-row query, k=100, d=128
    key2 = [int(x) for x in (20 + 40 * synth.uniform01(9, "golden/key", 40))]
    key2[5] = 0
    db2 = synth.unit_rows(10, "golden/db2", sum(key2), 128)
    pos2 = np.pad(np.cumsum(key2), (1, 0))
    qn = db2[pos2[17] + 3: pos2[17] + 22] + 0.8 * synth.unit_rows(11, "golden/n", 19, 128)
    qn /= np.linalg.norm(qn, axis=1, keepdims=True)
    cases["random_noisy"] = dict(db=db2, key=key2, q=qn.astype(np.float32), k=100)
    return cases


def gen_database():
    store = {}
    _install_faiss(store)
    import database as refdb
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, c in database_cases().items():
            dd = os.path.join(td, name)
            os.makedirs(dd)
            key = np.asarray(c["key"], np.int32)
            key.tofile(os.path.join(dd, "landmarkKey"))
            open(os.path.join(dd, "songList.txt"), "w").write("".join("song%d.wav\n" % i for i in range(len(key))))
            idx = _FlatIP(np.ascontiguousarray(c["db"], np.float32).reshape(-1, c["q"].shape[1]))
            store[os.path.join(dd, "landmarkValue")] = idx
            fsm = c.get("fsm", 1)
            dbo = refdb.Database(dd, {"top_k": c["k"], "frame_shift_mul": fsm}, 0.5)
            q = np.ascontiguousarray(c["q"], np.float32)
            if "force_labels" in c:
                idx.search = lambda qq, k, _v=c["force_labels"]: (
                    np.zeros((qq.shape[0], k), np.float32), np.full((qq.shape[0], k), _v, np.int64))
            D, I = idx.search(q, c["k"])
            score, (song, sec), ss = dbo.query_embeddings_base(q)
            out[name + "_db"] = idx.xb
            out[name + "_key"] = key
            out[name + "_q"] = q
            out[name + "_labels"] = I
            out[name + "_fsm"] = np.array(fsm)
            out[name + "_score"] = np.array(score, np.float64)
            out[name + "_song"] = np.array(song)
            out[name + "_sec"] = np.array(sec, np.float64)
            out[name + "_song_score"] = ss
            print("database", name, score, song, sec)
    np.savez_compressed(os.path.join(OUT, "database.npz"), **out)


# ----------------------------------------------------------------------- evaluator
def accuracy_cases():
    """Two seeded (expected.csv, *_detail.csv) pairs in the formats genquery.py:139-160 and matcher.py:84,158-163
    write, with every branch of tools/accuracy.py:24-45: wrong song, right song off by 0.2 / 0.25 / 0.3 / 0.5 /
    0.75 s, negative times, 'error' rows, paths whose basenames carry the match (accuracy.py:14,28-31)."""
    cases = {}
    for ci, (n, seed) in enumerate([(40, 31), (257, 32)]):
        u = synth.uniform01(seed, "golden/acc", 4 * n).reshape(4, n)
        gt, pr = [["query", "answer", "time", "snr"]], [["query", "answer", "score", "time", "part_scores"]]
        for j in range(n):
            song = int(u[0, j] * 23)
            tm = round(float(u[1, j]) * 20.0, 3)
            q = "/data/q%d/query%04d.wav" % (ci, j)
            gt.append([q, "/music/a/song%03d.wav" % song, repr(tm), "0"])
            kind = int(u[2, j] * 9)
            ans = "/other/prefix/song%03d.wav" % (song if kind != 0 else (song + 1) % 23)
            dt = [0.0, 0.0, 0.2, -0.25, 0.3, -0.5, 0.75, 0.0, -0.1][kind]
            if kind == 7:
                pr.append([os.path.basename(q), "error", "-inf", "0"])
            else:
                pr.append(["rel/" + os.path.basename(q), ans, repr(float(u[3, j])), repr(tm + dt)])
        cases["acc%d" % ci] = (gt, pr)
    return cases


def _csv_text(rows):
    import csv
    import io
    buf = io.StringIO()
    csv.writer(buf).writerows(rows)
    return buf.getvalue()


def gen_accuracy():
    """Runs the reference's tools/accuracy.py (a script: argparse at import) on the seeded pairs and stores its
    three printed lines.  Only the csv INPUTS (regenerated from seeds in the test) and the printed OUTPUT are data
    of this repo; the script itself is executed from /root/reference."""
    import contextlib
    import io
    import runpy
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, (gt, pr) in accuracy_cases().items():
            g, p = os.path.join(td, name + "_expected.csv"), os.path.join(td, name + "_detail.csv")
            open(g, "w", newline="").write(_csv_text(gt))
            open(p, "w", newline="").write(_csv_text(pr))
            argv, sys.argv = sys.argv, ["accuracy.py", g, p]
            buf = io.StringIO()
            try:
                with contextlib.redirect_stdout(buf):
                    runpy.run_path(os.path.join(REF, "tools", "accuracy.py"), run_name="__main__")
            finally:
                sys.argv = argv
            out[name] = buf.getvalue()
            print("accuracy", name, repr(buf.getvalue()))
    json.dump(out, open(os.path.join(OUT, "accuracy.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs("logs", exist_ok=True)
    which = sys.argv[1:] or ["encoder", "segmenter", "database", "accuracy"]
    if "encoder" in which:
        gen_encoder()
    if "segmenter" in which:
        gen_segmenter()
    if "database" in which:
        gen_database()
    if "accuracy" in which:
        gen_accuracy()
