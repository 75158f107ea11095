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
    # float64 for all 5,016 windows by torch's float64 kernels on the GPU (checked against the host's float64 on 76 of
    # them inside the tool); the fp32 host oracle -- the scarce resource of this suite -- on every 4th query
    out = eb.run(config, n_queries=264, snr=0.0, workers=workers, worst=0, legs=("default",), oracle32_every=4,
                 f64_where="gpu", inline_default_leg=True, log=lambda *a: print(*a, file=sys.stderr, flush=True))
    os.makedirs(os.path.join(REPO, "gpurun_out", "r5"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "r5", "embedding_error_budget_test_%s.json" % config), "w"), indent=1)
    assert out["segments"] >= 5000 and out["oracle32_windows"] >= 1250
    assert out["float64_gpu_vs_host_check"]["max_abs_diff_gpu_f64_vs_host_f64"] < 1e-10
    # the GPU's arithmetic (same mel bank on both sides) against float64, next to what torch-CPU fp32 itself loses
    gpu, o32 = out["legs"]["default"]["gpu_vs_f64_product_bank"]["max"], out["oracle32_vs_f64"]["max"]
    assert gpu <= max(2.0 * o32, 3e-5), (gpu, o32)
    # what the parity bar sees end to end (GPU with its fp32-built bank vs the fp32 oracle with its float64-built one),
    # and the share of it that is the two statements of the unpinned mel bank and nothing else (round 5's finding: the
    # bank gap of <= 3.8e-5 per weight moves fingerprints by more than all fp32 rounding of either side)
    assert out["legs"]["default"]["gpu_vs_oracle32"]["max"] < 1e-4
    assert out["mel_bank_statement_gap_f64_product_bank_vs_f64"]["max"] < 8e-5
