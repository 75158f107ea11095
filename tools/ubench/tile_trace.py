"""Per-CU timeline of conv_gemm_ln_w22_kernel from in-kernel timestamps (tuning aid, round 6).

Needs a library built with -DPFANN_TILE_TRACE in place of pfann_amd/libpfann_amd.so:
    PFANN_HIPCC_FLAGS=-DPFANN_TILE_TRACE python -m pfann_amd.build --force
    python tools/ubench/tile_trace.py [windows=4864] [rows_per_sample=1024] [out.json]
Every workgroup leaves (start, first sub-step, end of the K loop, end) at 100 MHz plus HW_ID / XCC_ID.  From them, per CU:
how long 2 / 1 / 0 workgroups are resident, how long 2 / 1 / 0 of them are inside their K loops, the gap between a
workgroup's end and the start of its successor in the same slot, and the phase lengths."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from pfann_amd import lib as plib, synth                      # noqa: E402
from pfann_amd.engine import Engine                           # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4864
RPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
OUT = sys.argv[3] if len(sys.argv) > 3 else None
params = json.load(open("configs/default.json"))
eng = Engine(params, 0, max_batch=B)
eng.load_state_dict(synth.make_state_dict(params, seed=123))
lib = plib.load()
lib.pfann_debug_set_tile_trace.restype = ctypes.c_int
lib.pfann_debug_set_tile_trace.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int]
pcm = synth.make_songs_torch(list(range(64)), 30.0, device="cuda").reshape(-1)
wav = eng.pcm16_to_mono(pcm)
starts = (torch.arange(B, device="cuda") * 1571) % (wav.shape[0] - 8000)
for _ in range(2):
    eng.embed_windows(wav, starts)
cap = 1 << 19
buf = torch.zeros((cap, 12), dtype=torch.int64, device="cuda")
assert lib.pfann_debug_set_tile_trace(buf.data_ptr(), cap, RPS) == 0
eng.embed_windows(wav, starts)
torch.cuda.synchronize()
assert lib.pfann_debug_set_tile_trace(None, 0, 0) == 0
t = buf.cpu().numpy()
t = t[t[:, 0] != 0]
n = t.shape[0]
hw, xcc = t[:, 4], t[:, 5] & 0xF
cu_id, sh_id, se_id = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
wave_id, simd_id = hw & 0xF, (hw >> 4) & 0x3
cu = ((xcc * 8 + se_id) * 2 + sh_id) * 16 + cu_id
t0 = t[:, 0].min()
st, lb, le, en = [(t[:, i] - t0).astype(np.float64) * 10.0 for i in range(4)]     # ns
print("traced %d workgroups of the rows=%d layer on %d CUs (%d windows); launch span %.3f ms" % (n, RPS, len(np.unique(cu)), B, (en.max()) / 1e6))
print("wave slots of thread 0's wave:", dict(zip(*np.unique(wave_id, return_counts=True))))
PH = ["kk 0..2: fragment reads, requests, MFMAs", "stash: LayerNorm transform + LDS refill (+ its waits)", "kk 3 MFMAs", "wait for the weight tile (vmcnt)", "barrier", "wait for the requested rows (vmcnt) in front of the stash"]
ph = t[:, 6:12].astype(np.float64)
print("K-loop phases of wave 0, shader cycles per tile (mean):")
for i, nm in enumerate(PH):
    print("   %-56s %9.0f  (%.1f %%)" % (nm, ph[:, i].mean(), 100 * ph[:, i].mean() / ph.sum(axis=1).mean()))
print("phase means (ns): prologue %.0f  K loop %.0f  epilogue %.0f  whole %.0f" % ((lb - st).mean(), (le - lb).mean(), (en - le).mean(), (en - st).mean()))
# steady state window: between the 10th and 90th percentile of start times
w0, w1 = np.percentile(st, 10), np.percentile(st, 90)
res = {"resident": np.zeros(4), "in_loop": np.zeros(4)}
gaps, lag = [], []
for c in np.unique(cu):
    m = cu == c
    ev = []
    for a, b_, c_, d in zip(st[m], lb[m], le[m], en[m]):
        ev += [(a, 0, 1), (d, 0, -1), (b_, 1, 1), (c_, 1, -1)]
    ev.sort()
    cnt = [0, 0]
    prev = None
    for x, kind, dlt in ev:
        if prev is not None and x > w0 and prev < w1:
            dur = min(x, w1) - max(prev, w0)
            if dur > 0:
                res["resident"][min(cnt[0], 3)] += dur
                res["in_loop"][min(cnt[1], 3)] += dur
        cnt[kind] += dlt
        prev = x
    # successor gap: for each end, the next start on this CU after it (greedy, slot-agnostic)
    s_sorted = np.sort(st[m])
    for d in np.sort(en[m]):
        j = np.searchsorted(s_sorted, d)
        if j < len(s_sorted) and w0 < d < w1:
            gaps.append(s_sorted[j] - d)
    # lag between the two residents' loop ends
    les = np.sort(le[m])
    les = les[(les > w0) & (les < w1)]
    lag += list(np.diff(les))
tot = res["resident"].sum()
out = {"windows": B, "rows_per_sample": RPS, "workgroups": int(n), "cus": int(len(np.unique(cu))),
       "phase_ns": {"prologue": float((lb - st).mean()), "k_loop": float((le - lb).mean()), "epilogue": float((en - le).mean()),
                    "whole": float((en - st).mean())},
       "k_loop_phase_cycles_wave0": {nm: float(ph[:, i].mean()) for i, nm in enumerate(PH)},
       "resident_fraction": {str(i): float(res["resident"][i] / tot) for i in range(4)},
       "in_loop_fraction": {str(i): float(res["in_loop"][i] / tot) for i in range(4)},
       "successor_gap_ns": {"median": float(np.median(gaps)), "mean": float(np.mean(gaps)), "p90": float(np.percentile(gaps, 90))},
       "loop_end_lag_ns": {"median": float(np.median(lag)), "p10": float(np.percentile(lag, 10)), "p90": float(np.percentile(lag, 90))}}
print(json.dumps(out, indent=1))
if OUT:
    json.dump(out, open(OUT, "w"), indent=1)
