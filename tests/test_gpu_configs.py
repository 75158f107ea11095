"""BASELINE.json configs 2-5 at their stated sizes on one MI355X (config 1 = tests/test_gpu_cli.py).

  cfg 2  fma_medium_train scale: 10 k songs -> 590 k x 128 db, 2000 x 10 s queries at SNR 0
  cfg 3  fma_medium scale: 25 k songs -> 1.475 M rows, 2000 queries at each SNR in {-6 .. 8} dB
         (genall.sh:1-4, testall.sh:74-80), hit-rates through tools/accuracy.py, table written out
  cfg 4  large scale: 100 k songs -> 5.9 M rows (3.0 GB: the > 2 GB windows) on one GPU, and the sharded
         protocol at 1/8-of-that shard size (2 ranks x 12.5 k songs on this GPU) against the single-index result
  cfg 5  configs/n640d64.json (d = 64, depthwise) with fp16-only storage against the exact fp32 path

No datasets exist here: songs and queries come from the seeded torch generators of pfann_amd/synth.py (on the
device), weights are the seeded state_dict with the calibrated output bias.  At full size the checks are
size-independent properties of the search (labels in range, scores descending, reported score = q . db[label], no
sampled row beats the k-th reported score, ...) plus, on a sample of >= 32 queries, the whole path against the CPU
oracle: embeddings within 1e-4 and identical (song, offset) decisions."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from pfann_amd import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out", "r5")
SEG, QSEG, HOP = 59, 19, 4000

import gpu_workloads as gw          # noqa: E402 -- (tests/ is on sys.path: conftest.py) engines and databases built once per session


def make_queries(eng, n_songs, nq, snr, qid0=0):
    """-> (int16 device PCM [nq, 80000], source song per query, offsets in s, embeddings device [nq*19, d])"""
    import torch
    dev = eng.device
    q_song = [int((j * 7919 + 13) % n_songs) for j in range(qid0, qid0 + nq)]
    pcms, offs, embs = [], [], []
    for c0 in range(0, nq, 256):
        ids = q_song[c0:c0 + 256]
        qp, qo = synth.make_queries_torch(synth.make_songs_torch(ids, 30.0, device=dev),
                                          list(range(qid0 + c0, qid0 + c0 + len(ids))), 10.0, snr)
        pcms.append(qp)
        offs.append(qo)
        starts = (torch.arange(len(ids), device=dev)[:, None] * qp.shape[1] + torch.arange(QSEG, device=dev)[None, :] * HOP)
        embs.append(eng.embed_windows(eng.pcm16_to_mono(qp.reshape(-1)), starts.reshape(-1)))
    return torch.cat(pcms), np.asarray(q_song), torch.cat(offs).cpu().numpy(), torch.cat(embs)


def search_properties(torch, index, db, q, D, I, k, exact_scores=True, tol=2e-6):
    """Size-independent properties of a top-k answer, checked on the device for every query row."""
    n = db.shape[0]
    assert I.shape == (q.shape[0], k) and int(I.min()) >= 0 and int(I.max()) < n, "labels out of range"
    assert bool((D[:, 1:] <= D[:, :-1]).all()), "scores not descending"
    assert bool((torch.sort(I, dim=1).values[:, 1:] != torch.sort(I, dim=1).values[:, :-1]).all()), "duplicate labels"
    if exact_scores:
        worst = 0.0
        for r0 in range(0, q.shape[0], 4096):                       # D[r, i] == q[r] . db[I[r, i]]
            rows = db[I[r0:r0 + 4096].reshape(-1)].reshape(-1, k, db.shape[1])
            chk = torch.einsum("qkd,qd->qk", rows, q[r0:r0 + 4096])
            worst = max(worst, float((chk - D[r0:r0 + 4096]).abs().max()))
        assert worst < tol, "reported scores differ from q . db[label] by %g" % worst
    # exactness witness: no row of a 64 k-row random sample may beat the k-th reported score unless it is reported
    g = torch.Generator(device=db.device)
    g.manual_seed(5)
    samp = torch.randint(0, n, (65536,), device=db.device, generator=g)
    xs = db[samp]
    misses = 0
    for r0 in range(0, q.shape[0], 2048):
        s = q[r0:r0 + 2048] @ xs.T
        beat = s > (D[r0:r0 + 2048, k - 1:k] + tol)
        if bool(beat.any()):
            rr, cc = beat.nonzero(as_tuple=True)
            reported = (I[r0 + rr] == samp[cc][:, None]).any(dim=1)
            misses += int((~reported).sum())
    assert misses == 0, "%d sampled rows beat the k-th reported score without being in the list" % misses


def hit_rates(q_song, q_off, res, hop_s=0.5):
    ok = res["song"] == q_song
    err = np.abs(res["offset"] * hop_s - q_off)
    return float(ok.mean()), float((ok & (err <= 0.5)).mean()), float((ok & (err <= 0.25)).mean())


def _save(name, obj):
    os.makedirs(OUT, exist_ok=True)
    json.dump(obj, open(os.path.join(OUT, name), "w"), indent=1)


def _run_queries(index, emb, k, nq):
    qstart = np.arange(nq, dtype=np.int64) * QSEG
    qlen = np.full(nq, QSEG, np.int32)
    D, I = index.search(emb, k)
    res, _ = index.match(emb, I, qstart, qlen)
    return D, I, res


def test_config2_10k_songs_queries_snr0():
    """the session's config-2 workload (gpu_workloads.cfg2_state: database, all 2000 queries, GPU decisions AND the CPU
    oracle's over the first gw.CFG2_QUERIES of them -- tests/test_gpu_decision_parity.py asserts on those): the search's size-independent
    properties, the one-query regime against the batched answer, hit-rates."""
    import torch
    t0 = time.time()
    rec, st = gw.cfg2_state()
    params, k, nq = st["params"], st["params"]["indexer"]["top_k"], 2000
    db, index, emb, res = st["shard"], st["index"], st["emb"], st["res"]
    assert db.shape[0] == 590000 and rec["db_rows"] == 590000 and rec["gpu_queries"] == nq and rec["queries"] == gw.CFG2_QUERIES
    D, I = index.search(emb, k)
    assert abs(float(emb.norm(dim=1).mean()) - 1) < 1e-5
    search_properties(torch, index, db, emb, D, I, k)
    # one query at a time (the HBM-bound small-batch kernels) must give the batched answer
    for j in (0, 777, 1999):
        D1, I1 = index.search(emb[j * QSEG:(j + 1) * QSEG].contiguous(), k)
        assert torch.equal(torch.sort(I1, 1).values, torch.sort(I[j * QSEG:(j + 1) * QSEG], 1).values)
        assert float((D1 - D[j * QSEG:(j + 1) * QSEG]).abs().max()) < 2e-6
    _, q_off = synth.make_queries_torch(synth.make_songs_torch(st["q_song"][:256].tolist(), 30.0, device=db.device), list(range(256)), 10.0, 0.0)
    hr = hit_rates(st["q_song"][:256], q_off.cpu().numpy(), res[:256])
    _save("cfg2.json", {"db_rows": int(db.shape[0]), "queries": nq, "snr": 0, "song/near/exact (first 256 queries)": hr,
                        "oracle_queries": rec["queries"], "max_emb_err_vs_oracle": rec["max_embedding_abs_diff"],
                        "top1_hit_rate_all_queries": rec["top1_hit_rate_gpu_all_queries"], "seconds": time.time() - t0})
    print("cfg2: %d rows, hit-rate %.4f, emb err %.2e, %.1fs" % (db.shape[0], rec["top1_hit_rate_gpu"], rec["max_embedding_abs_diff"], time.time() - t0))
    assert rec["top1_hit_rate_gpu"] > 0.5 and hr[0] > 0.5


def test_config3_25k_songs_snr_sweep(tmp_path):
    import csv
    import torch
    from pfann_amd.database import DeviceIndex
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import accuracy
    t0 = time.time()
    params, sd, eng = gw.engine("default")
    d, k, n_songs, nq = 128, params["indexer"]["top_k"], 25000, 2000
    db, pos = gw.database(n_songs)
    index = DeviceIndex(d, 0)
    index.load(db, pos, 0)
    samples = []
    table = {}
    for si, snr in enumerate([-6, -4, -2, 0, 2, 4, 6, 8]):
        q_pcm, q_song, q_off, emb = make_queries(eng, n_songs, nq, float(snr), qid0=si * nq)
        D, I, res = _run_queries(index, emb, k, nq)
        search_properties(torch, index, db, emb, D, I, k)
        if snr in (-6, 0, 8):                                       # oracle decisions on 12 queries of three SNRs = 36
            js = list(range(0, nq, nq // 12))[:12]                   # (collected here, run in ONE pool of oracle processes below)
            samples.append((q_pcm[js].cpu().numpy(), res[js], np.concatenate([emb[j * QSEG:(j + 1) * QSEG].cpu().numpy() for j in js])))
        # the reference's protocol: expected.csv + *_detail.csv through tools/accuracy.py (accuracy.py:34-45)
        gt, pr = tmp_path / ("expected_%d.csv" % snr), tmp_path / ("result_%d_detail.csv" % snr)
        with open(gt, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["query", "answer", "time", "snr"])
            w.writerows([["q%05d.wav" % j, "song%06d.wav" % q_song[j], repr(float(q_off[j])), snr] for j in range(nq)])
        with open(pr, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["query", "answer", "score", "time", "part_scores"])
            w.writerows([["q%05d.wav" % j, "song%06d.wav" % res[j]["song"], res[j]["score"], res[j]["offset"] * 0.5]
                         for j in range(nq)])
        r = accuracy.evaluate(str(gt), str(pr))
        assert r["total"] == nq
        table[str(snr)] = {"song": r["song"] / nq, "near": r["near"] / nq, "exact": r["exact"] / nq}
        assert (r["song"] / nq, r["near"] / nq, r["exact"] / nq) == pytest.approx(hit_rates(q_song, q_off, res), abs=1e-12)
        print("cfg3 snr %+d dB: song %.4f near %.4f exact %.4f" % (snr, r["song"] / nq, r["near"] / nq, r["exact"] / nq))
    gw.oracle_sample(params, sd, db.cpu().numpy(), pos, np.concatenate([a for a, _, _ in samples]), range(36),
                     np.concatenate([b for _, b, _ in samples]), np.concatenate([c for _, _, c in samples]), k)
    _save("cfg3_snr_sweep.json", {"db_rows": int(db.shape[0]), "queries_per_snr": nq, "hit_rates": table,
                                  "seconds": time.time() - t0})
    rates = [table[str(s)]["song"] for s in (-6, -4, -2, 0, 2, 4, 6, 8)]
    assert rates[-1] >= rates[0] and rates[-1] > 0.5               # cleaner queries are not harder


def test_config4_100k_songs_single_gpu():
    import torch
    from pfann_amd.database import DeviceIndex
    t0 = time.time()
    params, sd, eng = gw.engine("default")
    d, k, n_songs, nq = 128, params["indexer"]["top_k"], 100000, 512
    db, pos = gw.database(n_songs)
    assert db.shape[0] == 5900000 and db.numel() * 4 > (2 << 30)      # beyond one 2 GB buffer window
    index = DeviceIndex(d, 0)
    index.load(db, pos, 0)
    q_pcm, q_song, q_off, emb = make_queries(eng, n_songs, nq, 0.0)
    D, I, res = _run_queries(index, emb, k, nq)
    search_properties(torch, index, db, emb, D, I, k)
    for j in (0, 300):                                               # small-batch kernels on the 3 GB shard
        D1, I1 = index.search(emb[j * QSEG:(j + 1) * QSEG].contiguous(), k)
        assert torch.equal(torch.sort(I1, 1).values, torch.sort(I[j * QSEG:(j + 1) * QSEG], 1).values)
    sample = list(range(0, nq, nq // 32))[:32]
    worst = gw.oracle_sample(params, sd, db.cpu().numpy(), pos, q_pcm.cpu().numpy(), sample, res, emb.cpu().numpy(), k)
    hr = hit_rates(q_song, q_off, res)
    _save("cfg4_single.json", {"db_rows": int(db.shape[0]), "queries": nq, "song/near/exact": hr,
                               "max_emb_err_vs_oracle": worst, "seconds": time.time() - t0})
    print("cfg4: %d rows, hit-rates %.4f %.4f %.4f, %.1fs" % ((db.shape[0],) + hr + (time.time() - t0,)))


def test_config4_sharded_protocol_at_eighth_shard_size(tmp_path):
    """Two ranks on this GPU, 12,500 songs (737,500 rows = 1/8 of the 100 k-song db) per shard: all-to-all of per-shard
    top-k by query slice, merge, all-gather, owner-side rerank, winner pick -- against one index over the same 25 k songs."""
    common = ["--steps", "1", "--warmup", "0", "--queries", "256", "--db-songs", "25000", "--no-cpu-baseline", "--no-cli", "--no-prof",
              "--no-alt", "--max-batch", "4864", "--scaling", "strong"]
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + common + ["--dump-decisions", one],
                       capture_output=True, text=True, timeout=1200, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, PFANN_DIST_BACKEND="gloo", PFANN_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29741", os.path.join(REPO, "bench.py"),
                        "--gpus", "2"] + common + ["--dump-decisions", two],
                       capture_output=True, text=True, timeout=1200, cwd=REPO, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    a, b = np.load(one), np.load(two)
    assert a.shape == (256, 3) and np.array_equal(a[:, :2], b[:, :2])
    assert np.abs(a[:, 2] - b[:, 2]).max() < 1e-6


def test_config5_d64_fp16_storage():
    import torch
    from oracle import search as osr
    from pfann_amd.database import DeviceIndex
    t0 = time.time()
    params, sd, eng = gw.engine("n640d64", 4096)
    d, k, n_songs, nq = 64, params["indexer"]["top_k"], 10000, 2000
    assert eng.set_fused_layernorm(True) is True                     # depthwise model on the fused path (conv_dw_ln_kernel)
    db, pos = gw.database(n_songs, "n640d64", 4096)
    q_pcm, q_song, q_off, emb = make_queries(eng, n_songs, nq, 0.0)
    i32 = DeviceIndex(d, 0)
    i32.load(db, pos, 0)
    D32, I32, res32 = _run_queries(i32, emb, k, nq)
    search_properties(torch, i32, db, emb, D32, I32, k)
    sample = list(range(0, nq, nq // 32))[:32]
    db_host = db.cpu().numpy()
    worst = gw.oracle_sample(params, sd, db_host, pos, q_pcm.cpu().numpy(), sample, res32, emb.cpu().numpy(), k)
    del i32
    i16 = DeviceIndex(d, 0, storage="f16")
    i16.load(db, pos, 0)
    assert i16.lib.pfann_db_bytes(i16.handle) == db.shape[0] * d * 2
    D16, I16, res16 = _run_queries(i16, emb, k, nq)
    db16 = db.half().float()
    q16 = emb.half().float()
    search_properties(torch, i16, db16, q16, D16, I16, k, tol=3e-6)     # s16 = fl16(q) . fl16(x), fp32 accumulation
    e_host = emb.cpu().numpy()
    for j in sample[:8]:                                             # and the fp16 oracle on a few whole queries
        Dr, Ir = osr.flat_ip_topk_f16(e_host[j * QSEG:(j + 1) * QSEG], db_host, k)
        got = I16[j * QSEG:(j + 1) * QSEG].cpu().numpy()
        for r in range(QSEG):
            for lab in set(got[r].tolist()) ^ set(Ir[r].tolist()):
                s16 = float(db_host[lab].astype(np.float16).astype(np.float64) @ e_host[j * QSEG + r].astype(np.float16).astype(np.float64))
                assert abs(s16 - Dr[r, k - 1]) < 3e-6, "fp16 top-k differs from the fp16 oracle beyond a k-th tie"
    same = float(((res16["song"] == res32["song"]) & (res16["offset"] == res32["offset"])).mean())
    overlap = float((torch.sort(I16, 1).values == torch.sort(I32, 1).values).float().mean())
    h32, h16 = hit_rates(q_song, q_off, res32), hit_rates(q_song, q_off, res16)
    _save("cfg5_fp16.json", {"db_rows": int(db.shape[0]), "d": d, "queries": nq, "identical_decisions_fp16_vs_fp32": same,
                             "topk_label_overlap": overlap, "hit_rates_fp32": h32, "hit_rates_fp16": h16,
                             "max_emb_err_vs_oracle": worst, "seconds": time.time() - t0})
    print("cfg5: fp16 vs fp32 identical decisions %.4f, top-k slot overlap %.4f, song hit-rate fp32 %.4f fp16 %.4f, %.1fs"
          % (same, overlap, h32[0], h16[0], time.time() - t0))
    assert same >= 0.97, "fp16 storage changed %.2f %% of the decisions" % (100 * (1 - same))
    assert abs(h16[0] - h32[0]) <= 0.02
