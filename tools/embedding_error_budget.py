#!/usr/bin/env python
"""Who spends the 1e-4 fingerprint budget (north_star "within 1e-4"; model.py:54-73).

The GPU path and the fp32 CPU oracle both round.  This tool puts a third point next to them: the SAME op sequence
carried out in float64 (oracle/melspec.melspec_f64 -> oracle/encoder.encode(dtype=float64)) on every segment of a
query population, and reports per set

    |GPU - f64|        the product's distance from the exact value of the reference's formula
    |oracle32 - f64|   torch-CPU fp32's own distance from it (MKL-DNN convs + fp32 LayerNorm)
    |GPU - oracle32|   what the parity tests see (the sum of the two, at worst)

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
    for c0 in range(0, nq, per):
        qp = q_pcm[c0:c0 + per]
        starts = (torch.arange(qp.shape[0], device=dev)[:, None] * L + torch.arange(QSEG, device=dev)[None, :] * HOP).reshape(-1)
        embs.append(eng.embed_windows(eng.pcm16_to_mono(qp.reshape(-1)), starts))
    out = {"emb": torch.cat(embs).cpu().numpy()}
    if taps_path:
        seg = np.load(taps_path)                      # global window numbers: query * QSEG + t
        starts = torch.as_tensor((seg // QSEG) * L + (seg % QSEG) * HOP, device=dev)
        wav = eng.pcm16_to_mono(q_pcm.reshape(-1))
        eng.debug_keep(True)
        e = eng.embed_windows(wav, starts)
        torch.cuda.synchronize()
        for i in range(16):
            out["tap%d" % i] = eng.debug_activation(i, len(seg))
        eng.debug_keep(False)
        out["tap_emb"] = e.cpu().numpy()
        # the front-end by itself: windows cut the way the kernel cuts them, through the operator
        idx = starts[:, None] + torch.arange(eng.seg_len, device=dev)[None, :]
        segs = wav[idx]
        out["tap_mel"] = eng.melspec(segs - segs.mean(dim=1, keepdim=True)).cpu().numpy()
    np.savez(out_path, **out)


def _stats(diff):
    """diff [n, d] -> per-segment maxima summarised."""
    m = np.abs(diff).max(axis=1)
    return {"max": float(m.max()), "p99.9": float(np.quantile(m, 0.999)), "p99": float(np.quantile(m, 0.99)),
            "median": float(np.median(m)), "argmax_segment": int(m.argmax())}


def run(config="default", n_queries=2000, snr=0.0, workers=32, plan=9728, worst=20, legs=("default", "no_w22", "unfused"),
        log=print, keep=False):
    import torch
    import oracle_pool
    params = json.load(open(os.path.join(REPO, "configs", config + ".json")))
    sd, calibrated = weights(params)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="pfann_budget_", dir=base)
    t0 = time.time()
    q_pcm = query_pcm(n_queries, snr).cpu().numpy()
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

    # the oracle pool starts first and runs beside the GPU legs (CPU only)
    import threading
    pool_out = {}
    th = threading.Thread(target=lambda: pool_out.update(oracle_pool.run_embed(params, sd, q_pcm, workers=workers)))
    th.start()
    gpu = {name: leg(name)["emb"] for name in legs}
    log("embedding_error_budget[%s]: %d GPU legs done at %.1f s" % (config, len(legs), time.time() - t0))
    th.join()
    e32, e64, e64m32 = pool_out["emb32"], pool_out["emb64"], pool_out["emb64_mel32"]
    log("embedding_error_budget[%s]: oracle pool %.1f s on %d processes" % (config, pool_out["wall_s"], pool_out["workers"]))
    nseg = e64.shape[0]
    res = {"config": config, "queries": n_queries, "segments": int(nseg), "snr_db": snr, "plan_batch": plan,
           "calibrated_head": calibrated, "tolerance": 1e-4,
           "exact": "oracle/melspec.melspec_f64 -> oracle/encoder.encode(dtype=float64): the reference's op sequence in double on "
                    "the float32 weights",
           "oracle32_vs_f64": _stats(e32 - e64),
           "f64_on_fp32_mel_vs_f64": _stats(e64m32 - e64),
           "legs": {}}
    for name in legs:
        res["legs"][name] = {"gpu_vs_f64": _stats(gpu[name] - e64), "gpu_vs_oracle32": _stats(gpu[name] - e32)}
    res["legs_differ_bitwise"] = {a + "_vs_" + b: int((gpu[a] != gpu[b]).any(axis=1).sum())
                                  for a in legs for b in legs if a < b}
    # ---- per sub-layer, the worst windows of the default leg ----
    if worst:
        m = np.abs(gpu[legs[0]] - e64).max(axis=1)
        seg = np.sort(np.argsort(m)[-worst:])
        taps_path = os.path.join(work, "worst.npy")
        np.save(taps_path, seg)
        ot = oracle_pool.run_taps(params, sd, q_pcm, seg)
        layers = []
        gl = {name: leg(name, taps_path) for name in legs}
        mel64 = ot["mel64"]
        res["worst_segments"] = {"segments": seg.tolist(), "gpu_vs_f64_each": m[seg].tolist(),
                                 "log_mel": {"oracle32_vs_f64": float(np.abs(ot["mel32"] - mel64).max()),
                                             **{name: float(np.abs(gl[name]["tap_mel"] - mel64).max()) for name in legs}}}
        for i in range(16):
            t64 = ot["tap64_%d" % i]
            rms = float(np.sqrt(np.mean(t64 ** 2)))
            row = {"sub_layer": "%d.%s" % (i // 2, "conv1" if i % 2 == 0 else "conv2"), "shape": list(t64.shape[1:]), "rms": rms,
                   "oracle32_vs_f64": float(np.abs(ot["tap32_%d" % i] - t64).max())}
            for name in legs:
                row[name + "_vs_f64"] = float(np.abs(gl[name]["tap%d" % i].reshape(t64.shape) - t64).max())
            layers.append(row)
        res["worst_segments"]["per_sub_layer_max_abs"] = layers
        res["worst_segments"]["embedding"] = {"oracle32_vs_f64": float(np.abs(ot["emb32"] - ot["emb64"]).max()),
                                              **{name: float(np.abs(gl[name]["tap_emb"] - ot["emb64"]).max()) for name in legs}}
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
            log=lambda *x: print(*x, file=sys.stderr, flush=True))
    print(json.dumps({k: v for k, v in r.items() if k != "worst_segments"}, indent=1))
    if "worst_segments" in r:
        for row in r["worst_segments"]["per_sub_layer_max_abs"]:
            print(row)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(r, open(a.out, "w"), indent=1)
