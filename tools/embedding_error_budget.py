#!/usr/bin/env python
"""Who spends the 1e-4 fingerprint budget (north_star "within 1e-4"; model.py:54-73).

The GPU path and the fp32 CPU oracle both round.  This tool puts a third point next to them: the SAME op sequence
carried out in float64 (oracle/melspec.melspec_f64 -> oracle/encoder.encode(dtype=float64)) on every segment of a
query population, and reports per set

    |GPU - f64|        the product's distance from the exact value of the reference's formula
    |oracle32 - f64|   torch-CPU fp32's own distance from it (MKL-DNN convs + fp32 LayerNorm)
    |GPU - oracle32|   what the parity tests see (the sum of the two, at worst)
    |GPU - f64pb|      ... from float64 evaluated with the PRODUCT's mel bank: the GPU's arithmetic error proper
    |f64pb - f64|      what the two statements of the (unpinned) mel bank -- fp32 torch ops the way torchaudio builds it vs
                       float64 from the definition, <= 3.8e-5 apart per weight -- do to a fingerprint by themselves

as max / p99.9 / p99 / median of the per-segment maxima, for three GPU legs that differ in ONE ingredient each:
    default      five-block F(2,2) loader on the stride-2 layers + LayerNorm statistics as E[z^2]-mean^2 partials
    no_w22       PFANN_NO_W22=1: plain six-block kernel everywhere (attributes the (e0-e1)*W0 differences)
    unfused      pfann_set_fused_layernorm(0): LayerNorm as its own two-pass kernel (attributes the fused statistics)
and, for the `--worst` segments with the largest |GPU - f64|, the same three distances at each of the 16 sub-layer
activations (pfann_debug_activation), relative to the activation's own RMS.  Each GPU leg is a subprocess (the
environment switches are read once per process).  The oracle runs in worker processes (tools/oracle_pool.py).

    python tools/embedding_error_budget.py --config default --queries 2000 --out profiles/r5/embedding_error_budget_default.json
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (REPO, os.path.join(REPO, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
QSEG, HOP = 19, 4000


def weights(params):
    from pfann_amd import synth
    try:
        return synth.make_state_dict_calibrated(params, seed=123), True
    except KeyError:
        return synth.make_state_dict(params, seed=123), False


def query_pcm(n_queries, snr, n_songs=10000, device="cuda"):
    """The queries tools/decision_parity.py uses (BASELINE config 2's: 10 s crops of the synthetic songs at `snr` dB)."""
    import torch
    from pfann_amd import synth
    q_song = [int((j * 7919 + 13) % n_songs) for j in range(n_queries)]
    pcms = []
    for c0 in range(0, n_queries, 256):
        ids = q_song[c0:c0 + 256]
        qp, _ = synth.make_queries_torch(synth.make_songs_torch(ids, 30.0, device=device), list(range(c0, c0 + len(ids))), 10.0, snr)
        pcms.append(qp)
    return torch.cat(pcms)


def song_pcm(n_queries, device="cuda"):
    """--material songs: `n_queries` ten-second pieces (19 windows each, the same shape as a query) cut from the CLEAN
    synthetic songs at 0 / 10 / 20 s -- database-side material: tonal, no added noise, quiet mel bins at the log's floor."""
    import torch
    from pfann_amd import synth
    out = []
    for c0 in range(0, (n_queries + 2) // 3, 256):
        ids = list(range(c0, min(c0 + 256, (n_queries + 2) // 3)))
        pcm = synth.make_songs_torch(ids, 30.0, device=device)                  # [songs, 240000] int16
        out.append(pcm.reshape(len(ids) * 3, 80000))
    return torch.cat(out)[:n_queries]


def gpu_leg(config, pcm_path, out_path, plan, unfused=False, taps_path=None):
    """One GPU leg in THIS process: fingerprints of every window of every query in q_pcm.npy (launch groups of `plan`
    windows, plan pinned -- the way the CLIs run), and the 16 sub-layer activations + log-mel of the windows listed in
    taps_path."""
    import torch
    from pfann_amd.engine import Engine
    params = json.load(open(os.path.join(REPO, "configs", config + ".json")))
    sd, _ = weights(params)
    eng = Engine(params, 0, max_batch=plan)
    eng.load_state_dict(sd)
    eng.set_plan_batch(plan)
    if unfused:
        assert not eng.set_fused_layernorm(False)
    dev = eng.device
    q_pcm = torch.from_numpy(np.load(pcm_path)).to(dev)
    nq, L = q_pcm.shape
    per = max(1, plan // QSEG)
    embs = []
    for c0 in range(0, nq if not taps_path else 0, per):          # (the attribution call wants the taps only)
        qp = q_pcm[c0:c0 + per]
        starts = (torch.arange(qp.shape[0], device=dev)[:, None] * L + torch.arange(QSEG, device=dev)[None, :] * HOP).reshape(-1)
        embs.append(eng.embed_windows(eng.pcm16_to_mono(qp.reshape(-1)), starts))
    out = {"emb": torch.cat(embs).cpu().numpy()} if embs else {}
    if taps_path:
        seg = np.load(taps_path)                      # global window numbers: query * QSEG + t
        starts = torch.as_tensor((seg // QSEG) * L + (seg % QSEG) * HOP, device=dev)
        wav = eng.pcm16_to_mono(q_pcm.reshape(-1))
        eng.debug_keep(True)
        taps, es = [[] for _ in range(16)], []
        for c0 in range(0, len(seg), 8):              # (pfann_debug_keep retains the first 8 samples of a call)
            es.append(eng.embed_windows(wav, starts[c0:c0 + 8].contiguous()))
            torch.cuda.synchronize()
            for i in range(16):
                taps[i].append(eng.debug_activation(i, min(8, len(seg) - c0)))
        eng.debug_keep(False)
        for i in range(16):
            out["tap%d" % i] = np.concatenate(taps[i])
        out["tap_emb"] = torch.cat(es).cpu().numpy()
        # the front-end by itself: windows cut the way the kernel cuts them, through the operator
        idx = starts[:, None] + torch.arange(eng.seg_len, device=dev)[None, :]
        segs = wav[idx]
        out["tap_mel"] = eng.melspec(segs - segs.mean(dim=1, keepdim=True)).cpu().numpy()
    np.savez(out_path, **out)


def f64_on_gpu(params, sd, q_pcm, bank, chunk=512, log=print):
    """The float64 yardstick for every window of q_pcm, evaluated by torch's float64 kernels on cuda:0 (oracle/melspec.
    melspec_f64_torch -> oracle/encoder.encode(dtype=float64, device="cuda")), with the oracle's own mel bank and with
    `bank` (the product's) -> (emb64, emb64_bank) float64 [windows, d].  The segmenter stays the oracle's numpy one."""
    import torch
    from oracle import encoder as oe
    from oracle import melspec as om
    from oracle import segmenter as osg
    e64, e64b = [], []
    for c0 in range(0, q_pcm.shape[0], max(1, chunk // QSEG)):
        segs = np.concatenate([osg.segment(osg.pcm_to_mono(np.asarray(q_pcm[j])[:, None]), 8000, HOP)
                               for j in range(c0, min(c0 + max(1, chunk // QSEG), q_pcm.shape[0]))])
        e64.append(oe.encode(om.melspec_f64_torch(segs, params, device="cuda"), sd, params, dtype=np.float64, device="cuda"))
        e64b.append(oe.encode(om.melspec_f64_torch(segs, params, bank=bank, device="cuda"), sd, params, dtype=np.float64, device="cuda"))
    torch.cuda.empty_cache()
    return np.concatenate(e64), np.concatenate(e64b)


def _finite(diff):
    return diff[~np.isnan(diff).any(axis=1)]


def _stats(diff):
    """diff [n, d] -> per-segment maxima summarised."""
    m = np.abs(diff).max(axis=1)
    return {"max": float(m.max()), "p99.9": float(np.quantile(m, 0.999)), "p99": float(np.quantile(m, 0.99)),
            "median": float(np.median(m)), "argmax_segment": int(m.argmax())}


def run(config="default", n_queries=2000, snr=0.0, workers=32, plan=9728, worst=20, legs=("default", "no_w22", "unfused"),
        log=print, keep=False, partial_out=None, oracle32_every=4, f64_where="gpu", inline_default_leg=False, material="queries"):
    """oracle32_every: the fp32 CPU oracle (the expensive side: ~300 windows/s on a whole host) embeds every n-th query;
    the distances that involve it are taken over those windows, everything else over all of them.  f64_where "gpu": the
    float64 yardstick by torch's float64 kernels on the GPU, cross-checked against the host evaluation on 4 queries;
    "cpu": all of it on the host (the oracle pool).  inline_default_leg: the `default` GPU leg in this process."""
    import torch
    import oracle_pool
    params = json.load(open(os.path.join(REPO, "configs", config + ".json")))
    sd, calibrated = weights(params)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="pfann_budget_", dir=base)
    t0 = time.time()
    q_pcm = (query_pcm(n_queries, snr) if material == "queries" else song_pcm(n_queries)).cpu().numpy()
    pcm_path = os.path.join(work, "q_pcm.npy")
    np.save(pcm_path, q_pcm)
    del_cache = getattr(torch.cuda, "empty_cache", None)
    if del_cache:
        del_cache()

    def leg(name, taps=None):
        env = dict(os.environ, PYTHONPATH=REPO)
        env.pop("PFANN_NO_W22", None)
        if name == "no_w22":
            env["PFANN_NO_W22"] = "1"
        out = os.path.join(work, "gpu_%s%s.npz" % (name, "_taps" if taps else ""))
        cmd = [sys.executable, os.path.abspath(__file__), "--gpu-leg", name, "--config", config, "--pcm", pcm_path, "--leg-out", out,
               "--plan", str(plan)] + (["--taps", taps] if taps else [])
        subprocess.run(cmd, env=env, check=True)
        return dict(np.load(out))

    # the mel bank the product hands its kernel (pfann_amd/engine.py: built the way torchaudio builds it, fp32 torch ops); the
    # oracle's own is written filter by filter in float64 (oracle/melspec.py).  Neither is pinned (torchaudio is absent):
    # the float64 evaluation is therefore done with BOTH, so that arithmetic error and bank statement are told apart
    from oracle import melspec as om
    from pfann_amd.engine import mel_filterbank
    bank = om.mel_filterbank_torchaudio(params["sample_rate"], params["stft_n"], params["n_mels"], params["f_min"], params["f_max"],
                                        params.get("naf_mode", False)).numpy()           # the oracle's restatement of torchaudio's ...
    assert np.array_equal(bank, mel_filterbank(params["sample_rate"], params["stft_n"], params["n_mels"], params["f_min"],
                                               params["f_max"], params.get("naf_mode", False)).numpy())   # ... IS the product's table
    # the fp32 oracle (host, worker processes) starts first and runs beside the GPU work
    import threading
    pool_out = {}
    sub = np.arange(0, n_queries, max(1, int(oracle32_every)))
    cpu_bank = bank if f64_where == "cpu" else None
    th = threading.Thread(target=lambda: pool_out.update(oracle_pool.run_embed(params, sd, q_pcm[sub], workers=workers, bank=cpu_bank,
                                                                               f64=f64_where == "cpu")))
    th.start()
    gpu = {}
    for name in legs:
        if name == "default" and inline_default_leg and "PFANN_NO_W22" not in os.environ:
            out_path = os.path.join(work, "gpu_default.npz")
            gpu_leg(config, pcm_path, out_path, plan)
            gpu[name] = np.load(out_path)["emb"]
        else:
            gpu[name] = leg(name)["emb"]
    log("embedding_error_budget[%s]: %d GPU legs done at %.1f s" % (config, len(legs), time.time() - t0))
    f64_check = None
    if f64_where == "gpu":
        e64, e64b = f64_on_gpu(params, sd, q_pcm, bank, log=log)
        log("embedding_error_budget[%s]: float64 yardstick on the GPU done at %.1f s" % (config, time.time() - t0))
        # ... and the same on the host for 4 queries: the two evaluations of one formula must agree far below anything measured
        chk = oracle_pool.run_taps(params, sd, q_pcm, np.arange(0, 4 * QSEG), bank=bank)
        f64_check = {"windows": 4 * QSEG, "max_abs_diff_gpu_f64_vs_host_f64": float(np.abs(chk["emb64"] - e64[:4 * QSEG]).max()),
                     "max_abs_diff_product_bank": float(np.abs(chk["emb64_bank"] - e64b[:4 * QSEG]).max())}
        assert f64_check["max_abs_diff_gpu_f64_vs_host_f64"] < 1e-10 and f64_check["max_abs_diff_product_bank"] < 1e-10, f64_check
    th.join()
    w_sub = (sub[:, None] * QSEG + np.arange(QSEG)[None, :]).reshape(-1)         # windows the fp32 oracle embedded
    e32 = pool_out["emb32"]
    if f64_where == "cpu":
        e64s, e64bs = pool_out["emb64"], pool_out["emb64_bank"]
        e64 = np.full((n_queries * QSEG, e32.shape[1]), np.nan)
        e64b = e64.copy()
        e64[w_sub], e64b[w_sub] = e64s, e64bs
        if len(sub) != n_queries:
            gpu = {name: np.where(np.isnan(e64[:, :1]), np.nan, g) for name, g in gpu.items()}
    log("embedding_error_budget[%s]: oracle pool %.1f s on %d processes" % (config, pool_out["wall_s"], pool_out["workers"]))
    nseg = e64.shape[0]
    res = {"config": config, "material": "10 s queries at %g dB SNR" % snr if material == "queries" else "clean synthetic songs (database side)",
           "queries": n_queries, "segments": int(nseg), "snr_db": snr if material == "queries" else None, "plan_batch": plan,
           "calibrated_head": calibrated, "tolerance": 1e-4,
           "f64": "oracle/melspec.melspec_f64 -> oracle/encoder.encode(dtype=float64): the reference's op sequence in double on the "
                  "float32 weights, with the oracle's own float64-built mel bank",
           "f64_product_bank": "the same in double with the mel bank the product's kernel is given (fp32 torch-op construction, "
                               "torchaudio's way): what the GPU's ARITHMETIC is judged against",
           "float64_evaluated": "torch float64 kernels on the GPU (im2col + dgemm; not the product's kernels)" if f64_where == "gpu" else "host",
           "float64_gpu_vs_host_check": f64_check,
           "oracle32_windows": int(len(w_sub)), "oracle32_every_nth_query": int(oracle32_every),
           "oracle32_vs_f64": _stats(e32 - e64[w_sub]),
           "mel_bank_statement_gap_f64_product_bank_vs_f64": _stats(_finite(e64b - e64)),
           "legs": {}}
    for name in legs:
        res["legs"][name] = {"gpu_vs_f64_product_bank": _stats(_finite(gpu[name] - e64b)), "gpu_vs_f64": _stats(_finite(gpu[name] - e64)),
                             "gpu_vs_oracle32": _stats(gpu[name][w_sub] - e32)}
    res["legs_differ_bitwise"] = {a + "_vs_" + b: int((gpu[a] != gpu[b]).any(axis=1).sum())
                                  for a in legs for b in legs if a < b}
    res["wall_s"] = round(time.time() - t0, 1)
    res["oracle_wall_s"] = round(pool_out["wall_s"], 1)
    if partial_out:                                   # the population figures are safe before the attribution stage starts
        os.makedirs(os.path.dirname(os.path.abspath(partial_out)), exist_ok=True)
        json.dump(res, open(partial_out, "w"), indent=1)
    # ---- per sub-layer, the windows where the default leg's ARITHMETIC is furthest from float64 ----
    if worst:
        m = np.nan_to_num(np.abs(gpu[legs[0]] - e64b).max(axis=1), nan=-1.0)
        seg = np.sort(np.argsort(m)[-worst:])
        taps_path = os.path.join(work, "worst.npy")
        np.save(taps_path, seg)
        ot = oracle_pool.run_taps(params, sd, q_pcm, seg, bank=bank)
        layers = []
        gl = {name: leg(name, taps_path) for name in legs}
        res["worst_segments"] = {"segments": seg.tolist(), "gpu_vs_f64_product_bank_each": m[seg].tolist(),
                                 "log_mel": {"oracle32_vs_f64": float(np.abs(ot["mel32"] - ot["mel64"]).max()),
                                             "f64_product_bank_vs_f64": float(np.abs(ot["mel64_bank"] - ot["mel64"]).max()),
                                             **{name + "_vs_f64_product_bank": float(np.abs(gl[name]["tap_mel"] - ot["mel64_bank"]).max())
                                                for name in legs}}}
        for i in range(16):
            t64, t64b = ot["tap64_%d" % i], ot["tap64_bank_%d" % i]
            row = {"sub_layer": "%d.%s" % (i // 2, "conv1" if i % 2 == 0 else "conv2"), "shape": list(t64.shape[1:]),
                   "rms": float(np.sqrt(np.mean(t64 ** 2))),
                   "oracle32_vs_f64": float(np.abs(ot["tap32_%d" % i] - t64).max()),
                   "f64_product_bank_vs_f64": float(np.abs(t64b - t64).max())}
            for name in legs:
                row[name + "_vs_f64_product_bank"] = float(np.abs(gl[name]["tap%d" % i].reshape(t64.shape) - t64b).max())
            layers.append(row)
        res["worst_segments"]["per_sub_layer_max_abs"] = layers
        res["worst_segments"]["embedding"] = {"oracle32_vs_f64": float(np.abs(ot["emb32"] - ot["emb64"]).max()),
                                              "f64_encoder_on_the_fp32_log_mel_vs_f64": float(np.abs(ot["emb64_mel32"] - ot["emb64"]).max()),
                                              "f64_product_bank_vs_f64": float(np.abs(ot["emb64_bank"] - ot["emb64"]).max()),
                                              **{name + "_vs_f64_product_bank": float(np.abs(gl[name]["tap_emb"] - ot["emb64_bank"]).max())
                                                 for name in legs}}
    res["wall_s"] = round(time.time() - t0, 1)
    res["oracle_wall_s"] = round(pool_out["wall_s"], 1)
    if not keep:
        import shutil
        shutil.rmtree(work, ignore_errors=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="default")
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--snr", type=float, default=0.0)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--plan", type=int, default=9728)
    ap.add_argument("--worst", type=int, default=20)
    ap.add_argument("--legs", default="default,no_w22,unfused")
    ap.add_argument("--material", default="queries", choices=["queries", "songs"])
    ap.add_argument("--oracle32-every", type=int, default=4, help="the fp32 CPU oracle embeds every n-th query (1: all)")
    ap.add_argument("--f64", default="gpu", choices=["gpu", "cpu"], help="where the float64 yardstick is evaluated")
    ap.add_argument("--out", default=None)
    ap.add_argument("--gpu-leg", default=None, help="(internal) run one GPU leg in this process")
    ap.add_argument("--pcm", default=None)
    ap.add_argument("--leg-out", default=None)
    ap.add_argument("--taps", default=None)
    a = ap.parse_args()
    if a.gpu_leg:
        gpu_leg(a.config, a.pcm, a.leg_out, a.plan, unfused=a.gpu_leg == "unfused", taps_path=a.taps)
        sys.exit(0)
    r = run(a.config, a.queries, a.snr, a.workers, a.plan, a.worst, tuple(a.legs.split(",")),
            log=lambda *x: print(*x, file=sys.stderr, flush=True), partial_out=a.out, oracle32_every=a.oracle32_every, f64_where=a.f64,
            material=a.material)
    print(json.dumps({k: v for k, v in r.items() if k != "worst_segments"}, indent=1))
    if "worst_segments" in r:
        for row in r["worst_segments"]["per_sub_layer_max_abs"]:
            print(row)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(r, open(a.out, "w"), indent=1)
