"""pfann_set_plan_batch: are a segment's fingerprint bits independent of the batch it is computed in?
Embeds the same windows alone, in small batches at different positions, permuted, and inside a full launch group, with
the plan fixed, and compares bit patterns; then prices the plan (variants of a 9728-batch at small batches)."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from pfann_amd import synth                      # noqa: E402
from pfann_amd.engine import Engine               # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
    params = json.load(open(os.path.join(REPO, "configs", cfg + ".json")))
    MB = 9728
    eng = Engine(params, 0, max_batch=MB)
    eng.load_state_dict(synth.make_state_dict_calibrated(params, seed=123) if cfg == "default" else synth.make_state_dict(params, seed=123))
    dev = eng.device
    pcm = synth.make_songs_torch(list(range(170)), 30.0, device=dev)            # [170, 240000]
    wav = eng.pcm16_to_mono(pcm.reshape(-1))
    L = pcm.shape[1]
    starts_all = (torch.arange(170, device=dev)[:, None] * L + torch.arange(59, device=dev)[None, :] * 4000).reshape(-1)[:MB]
    n = starts_all.shape[0]
    out = {}
    for plan in (MB, 0):
        assert eng.set_plan_batch(plan) == plan
        full = eng.embed_windows(wav, starts_all)
        torch.cuda.synchronize()
        pick = torch.tensor([0, 1, 63, 64, 127, 128, 200, 1000, 4863, 4864, 9000, n - 1], device=dev)
        res = {}
        # alone
        worst = 0
        for i in pick.tolist():
            e = eng.embed_windows(wav, starts_all[i:i + 1])
            worst = max(worst, int((e.view(torch.int32) != full[i:i + 1].view(torch.int32)).sum()))
        res["alone_vs_full_bits_differ"] = worst
        # small batches at several sizes / positions
        for B in (2, 19, 64, 65, 130, 1216, 4864, 4865):
            idx = torch.arange(B, device=dev) * 2 % n
            e = eng.embed_windows(wav, starts_all[idx])
            res["B%d_vs_full_bits_differ" % B] = int((e.view(torch.int32) != full[idx].view(torch.int32)).sum())
            res["B%d_max_abs" % B] = float((e - full[idx]).abs().max())
        perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        e = eng.embed_windows(wav, starts_all[perm])
        res["permuted_full_bits_differ"] = int((e.view(torch.int32) != full[perm].view(torch.int32)).sum())
        # timing
        for B in (19, 152, 1216, 2432, 4864, 9728):
            st = starts_all[:B].contiguous()
            eng.embed_windows(wav, st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 20 if B < 2000 else 5
            for _ in range(reps):
                eng.embed_windows(wav, st)
            torch.cuda.synchronize()
            res["B%d_segments_per_s" % B] = round(B * reps / (time.perf_counter() - t0), 1)
        out["plan_%d" % plan] = res
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.join(REPO, "gpurun_out", "r4"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "r4", "plan_invariance_%s.json" % cfg), "w"), indent=1)


if __name__ == "__main__":
    main()
