"""Full-population decision parity (north_star: "top-1 match / segment-offset decisions exactly on the
same queries"): ALL 2000 queries of BASELINE config 2 (10 k songs -> 590 k x 128 db, 10 s @ SNR 0) go through the CPU
oracle (tools/decision_parity.py: 24 oracle processes fed through tmpfs) next to the GPU path: fingerprints within
1e-4, (song, offset) identical, scores within 1e-5.  A flip is tolerated only when the oracle's own numbers show a tie
the reference itself would break by rounding (top-k boundary within 1e-5 / two alignments within 1e-6); anything
else is a bug.  Config 3's -6 dB and 8 dB sets (25 k songs) run the same tool; their records are committed under
profiles/r4/ (tools/decision_parity.py --songs 25000 --snr -6 | 8)."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cfg2_query_population_vs_oracle():
    import gpu_workloads as gw          # the session's config-2 workload: database, queries, GPU decisions, oracle decisions
    out, _ = gw.cfg2_state()
    os.makedirs(os.path.join(REPO, "gpurun_out", "r6"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "r6", "decision_parity_cfg2_snr0_test.json"), "w"), indent=1)
    nq = gw.CFG2_QUERIES
    assert out["queries"] == nq == 2000 and out["db_rows"] == 590000
    assert out["max_embedding_abs_diff"] < 1e-4, out["max_embedding_abs_diff"]
    assert out["max_score_abs_diff_where_decisions_agree"] < 1e-5
    assert out["bugs"] == 0, [f for f in out["flips"] if f["class"] == "bug"]
    assert out["identical_song_and_offset"] + len(out["flips"]) == nq
    assert out["identical_song_and_offset"] >= nq - 5, out["flips"]          # ties are rare events, not a population
    assert out["top1_hit_rate_gpu"] == out["top1_hit_rate_oracle"] or out["flips"]
