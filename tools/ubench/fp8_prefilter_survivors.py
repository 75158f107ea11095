"""Would an fp8 (e4m3) first stage pay for the batched scan?  ("port only if the survivor count stays
<= 2x today's".)  Measured on real fingerprints, independent of any kernel: the 1 M-segment synthetic database embedded by
the hot path, 512 ten-second SNR-0 queries.  For a sample of query rows: exact fp32 scores; rows and queries rounded to
fp16 / to e4m3 (per-vector power-of-two scaling into the format's range); the RIGOROUS margin of a pre-filter
    |q.x - q'.x'| <= |q'| |x - x'| + |q - q'| |x|          (Cauchy-Schwarz on the two rounding residuals, known per vector)
and the number of rows any exact pre-filter in that format must hand to the fp32 re-scoring: rows whose upper bound
s' + margin reaches the k-th largest lower bound s' - margin."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from pfann_amd import synth                      # noqa: E402
from pfann_amd.builder import embed_files         # noqa: E402
from pfann_amd.engine import Engine               # noqa: E402


def main():
    params = json.load(open(os.path.join(REPO, "configs", "default.json")))
    n_songs = int(sys.argv[1]) if len(sys.argv) > 1 else 16950
    eng = Engine(params, 0, max_batch=9728)
    eng.load_state_dict(synth.make_state_dict_calibrated(params, seed=123))
    dev = eng.device
    d, k = 128, 100

    class Pcm:
        def __init__(self, ids, pcm):
            self.files, self.pcm = ["s%d" % i for i in ids], pcm

        def load_pcm(self, i):
            return self.pcm[i]

        def __len__(self):
            return len(self.files)
    db = torch.empty((n_songs * 59, d), device=dev)
    for c0 in range(0, n_songs, 512):
        ids = list(range(c0, min(c0 + 512, n_songs)))
        pcm = synth.make_songs_torch(ids, 30.0, device=dev)
        for i, n_seg, e in embed_files(eng, Pcm(ids, pcm), 4000, batch_windows=9728):
            db[ids[i] * 59:(ids[i] + 1) * 59] = e
    q_song = [int((j * 7919 + 13) % n_songs) for j in range(64)]
    qp, _ = synth.make_queries_torch(synth.make_songs_torch(q_song, 30.0, device=dev), list(range(64)), 10.0, 0.0)
    starts = (torch.arange(64, device=dev)[:, None] * qp.shape[1] + torch.arange(19, device=dev)[None, :] * 4000).reshape(-1)
    q = eng.embed_windows(eng.pcm16_to_mono(qp.reshape(-1)), starts)            # 1216 query rows

    def rounded(x, fmt):
        if fmt == "f16":
            return x.half().float()
        # e4m3: scale every vector by a power of two so that its largest magnitude lands in [224, 448]
        mx = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
        sc = torch.exp2(torch.floor(torch.log2(448.0 / mx)))
        return (x * sc).to(torch.float8_e4m3fn).float() / sc
    out = {"db_rows": int(db.shape[0]), "query_rows": int(q.shape[0]), "k": k}
    s = q @ db.T                                                                 # exact scores [1216, N]
    kth = s.topk(k, dim=1).values[:, -1]
    out["kth_score_mean"] = float(kth.mean())
    for fmt in ("f16", "e4m3"):
        dbr, qr = rounded(db, fmt), rounded(q, fmt)
        rx = (db - dbr).norm(dim=1)                                              # |x - x'| per row
        rq = (q - qr).norm(dim=1)
        nx = db.norm(dim=1)
        nqr = qr.norm(dim=1)
        sr = qr @ dbr.T
        margin = nqr[:, None] * rx[None, :] + rq[:, None] * nx[None, :]          # rigorous, per (query row, db row)
        assert bool(((s - sr).abs() <= margin * 1.0001 + 1e-6).all())
        lower_kth = (sr - margin).topk(k, dim=1).values[:, -1]
        surv = ((sr + margin) >= lower_kth[:, None]).sum(dim=1).float()
        out[fmt] = {"margin_mean": float(margin.mean()), "margin_max": float(margin.max()),
                    "actual_error_max": float((s - sr).abs().max()),
                    "survivors_per_query_row_mean": float(surv.mean()), "survivors_median": float(surv.median()),
                    "survivors_max": float(surv.max()), "relative_residual_db_mean": float((rx / nx).mean())}
    out["e4m3_over_f16_survivors"] = out["e4m3"]["survivors_per_query_row_mean"] / out["f16"]["survivors_per_query_row_mean"]
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.join(REPO, "gpurun_out", "r4"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "r4", "fp8_prefilter_survivors.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
