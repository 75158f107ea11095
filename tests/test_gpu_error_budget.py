"""The 1e-4 fingerprint budget, triangulated: on >= 5,000 windows per config the GPU path must be no
further from the float64 evaluation of the reference's op sequence than twice what torch-CPU fp32 (the parity oracle) is
itself, with a floor of 3e-5 -- i.e. the product may not be the one that spends the tolerance.  The full-population
records (2000 queries per config, three GPU legs, per sub-layer attribution of the 20 worst windows) are
profiles/r5/embedding_error_budget_*.json, written by the same tool (tools/embedding_error_budget.py)."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config", ["default", "seg", "n640d64"])
def test_gpu_is_not_further_from_float64_than_the_fp32_oracle_allows(config):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import embedding_error_budget as eb
    workers = max(4, min(32, (os.cpu_count() or 8) // 8))
    out = eb.run(config, n_queries=264, snr=0.0, workers=workers, worst=0, legs=("default",),
                 log=lambda *a: print(*a, file=sys.stderr, flush=True))
    os.makedirs(os.path.join(REPO, "gpurun_out", "r5"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "r5", "embedding_error_budget_test_%s.json" % config), "w"), indent=1)
    assert out["segments"] >= 5000
    gpu, o32 = out["legs"]["default"]["gpu_vs_f64"]["max"], out["oracle32_vs_f64"]["max"]
    assert gpu <= max(2.0 * o32, 3e-5), (gpu, o32)
    assert out["legs"]["default"]["gpu_vs_oracle32"]["max"] < 1e-4
