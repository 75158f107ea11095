"""bench.py contract on a reduced workload (fast): one JSON line with the required keys,
roofline + cpu_baseline objects, and GPU decisions identical to the oracle's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract_small():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--queries", "16", "--db-songs", "400", "--cpu-queries", "4", "--cli-songs", "40", "--cli-queries", "8"],
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["vs_baseline"] is None and out["dtype"] == "f32" and "workload" in out["config"]
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] <= 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    mp = cb["multi_process"]
    assert mp["value"] > 0 and mp["queries"] == 16 and mp["identical_song_and_offset_vs_gpu"] == "16/16", mp
    assert cb["value"] == max(mp["value"], cb["single_process"]["value"]) and cb["cores"] in (mp["cores"], cb["single_process"]["cores"])
    par = out["oracle_decision_parity"]
    assert par["identical_song_and_offset"] == par["queries"]
    assert out["top1_hit_rate"] > 0.5
    assert out["builder"]["value"] > 0 and out["hbm_resident"]["value"] > 0 and "H2D" in out["value_includes"]
    # the timed loop runs two batches deep (stated in the line); the one-at-a-time figure is kept beside it
    assert out["step_overlap"].startswith("two batches deep") and out["serial"]["value"] > 0
    assert set(cb["stages_s"]) == {"compute embedding", "search", "rerank"} and cb["python_rerank"]["queries"] >= 1
    assert cb["python_rerank"]["same_decision_as_c_path"].split("/")[0] == cb["python_rerank"]["same_decision_as_c_path"].split("/")[1]
    cli = out["cli"]
    assert cli["builder"]["segments"] == 40 * 59 and cli["matcher"]["segments"] == 8 * 19, cli
    assert cli["cli_builder_segments_per_s"] > 0 and cli["cli_matcher_segments_per_s"] > 0
    assert "compute embedding" in cli["builder"]["stages_s"] and "search" in cli["matcher"]["stages_s"]
    assert cli["matcher"]["top1_hit_rate"] >= 0.5
    assert out["ranks_seen"] == 1 and out["devices"] == [0]
    assert out["alt_modes"]["fp16_db"]["decisions_identical_to_fp32_db"].endswith("/16")
    assert len(out["alt_modes"]["encoder_only"]) == 3 and all(v["segments_per_s"] > 0 for v in out["alt_modes"]["encoder_only"].values())
    assert out["seq_score_seam"]["same_best_song"] is True and out["seq_score_seam"]["calls"] == 200
    assert out["seq_score_seam"]["gpu_call_us_median"] <= out["seq_score_seam"]["gpu_call_us_p95"]
    # round 6: the whole step per launch group of 1 / 4 / 16 queries (19 / 76 / 304 windows: the one-query plan, the
    # middle-of-the-curve plan with its split-K layers, and the query-stationary search at 76 and 304 rows) decides like the timed step
    bc = out["batch_curve"]["points"]
    assert [pt["queries"] for pt in bc] == [1, 4, 16], bc
    for pt in bc:
        same, of = pt["decisions_identical_to_the_timed_step"].split("/")
        assert same == of and pt["segments_per_s"] > 0 and set(pt["kernel_ms_by_stage"]) == {"encoder", "scan", "matcher"}, pt
    assert bc[-1]["fraction_of_largest_group_rate"] == 1.0


def test_live_traffic_of_the_conv_gemms_is_measured_under_rocprofv3(tmp_path):
    """bench.py's roofline.traffic leg (tools/live_traffic.py): two rocprofv3 --pmc passes of a child that pushes one launch
    group (9728 windows) through the encoder.  Every one of the 15 conv GEMM launches of both repetitions is
    counted in both passes, and the bytes per launch sit where the committed passes of the whole bench put them (3.76 GB; the
    algorithmic figure is 3.22 GB), i.e. the counters, their x2 correction and the per-launch averaging are the ones the
    committed profiles use."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import live_traffic
    W = 9728
    lt = live_traffic.measure(windows=W, keep_dir=str(tmp_path))
    assert lt["dispatches"] == 30 and lt["launches_per_group"] == 15 and lt["windows_per_group"] == W, lt
    assert 3.22e9 < lt["hbm_bytes_per_launch"] < 4.3e9, lt
    assert lt["fetch_size_kb_per_launch"] > 0 and lt["write_size_kb_per_launch"] > 0


def test_two_deep_and_serial_loops_decide_alike(tmp_path):
    """bench.py times its batches two deep (H2D of batch i+1 under batch i's kernels, decisions read one batch late);
    --serial runs them one at a time.  Same decisions either way, and the JSON says which loop ran."""
    import numpy as np
    common = ["--steps", "3", "--warmup", "1", "--db-songs", "600", "--queries", "24", "--no-cpu-baseline", "--no-cli",
              "--no-alt", "--max-batch", "512"]
    outs, lines = [], []
    for extra, name in ((["--serial"], "serial.npy"), ([], "deep.npy")):
        p = str(tmp_path / name)
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + common + extra + ["--dump-decisions", p],
                           capture_output=True, text=True, timeout=900, cwd=REPO)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(np.load(p))
        lines.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0]))
    assert np.array_equal(outs[0], outs[1])
    assert lines[0]["step_overlap"].startswith("none") and lines[0]["serial"] is None
    assert lines[1]["step_overlap"].startswith("two batches deep") and lines[1]["serial"]["value"] > 0
    assert lines[0]["steps"] == lines[1]["steps"] == 3


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_two_rank_sharded_path_matches_single_gpu(tmp_path, scaling):
    """N>1 on the real kernels: 2 ranks sharing this box's GPU (gloo-staged collectives, debugging
    aid) must reach exactly the single-GPU decisions.  strong: the 24 queries split over the ranks; weak (bench.py's
    default): every rank brings 12 of its own -- the same 24 queries either way."""
    import numpy as np
    common = ["--steps", "1", "--warmup", "0", "--db-songs", "600", "--no-cpu-baseline", "--no-cli", "--no-prof", "--max-batch", "512"]
    one = str(tmp_path / "one.npy")
    two = str(tmp_path / "two.npy")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + common + ["--queries", "24", "--dump-decisions", one],
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, PFANN_DIST_BACKEND="gloo", PFANN_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29733" if scaling == "strong" else "29735",
                        os.path.join(REPO, "bench.py"), "--gpus", "2", "--scaling", scaling,
                        "--queries", "24" if scaling == "strong" else "12"] + common + ["--dump-decisions", two],
                       capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    a, b = np.load(one), np.load(two)
    assert np.array_equal(a[:, :2], b[:, :2])
    assert np.abs(a[:, 2] - b[:, 2]).max() < 1e-6
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling
    assert out["config"]["queries_per_step"] == 24 and out["config"]["queries_per_step_per_gpu"] == 12


def test_two_rank_default_is_the_same_job_strong_scaling_with_scan_throughput_and_collectives(tmp_path):
    """The line the driver's SCALE runs will read: N > 1 defaults to STRONG scaling (the same job at
    every N), carries `scan_throughput` (db rows x query rows per second of scan-kernel time, slowest rank), the event time
    of every collective of the exchange protocol, the weak-scaling figure as the side key, and a CLI leg run by the
    tools' own 2 ranks.  2 ranks on this box's one GPU (gloo-staged collectives)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--queries", "48", "--db-songs", "600", "--max-batch", "512", "--cli-songs", "60",
                        "--cli-queries", "12"],
                       capture_output=True, text=True, timeout=1200, cwd=REPO,
                       env=_clean_env(PFANN_DIST_BACKEND="gloo", PFANN_FORCE_DEVICE="0", PFANN_MAX_BATCH="512"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["queries_per_step"] == 48
    assert out["config"]["queries_per_step_per_gpu"] == 24
    st = out["scan_throughput"]
    assert st["value"] > 0 and len(st["scan_ms_per_step_per_rank"]) == 2 and st["db_rows"] == 600 * 59
    assert st["query_rows_per_step"] == 48 * 19 and any(t.startswith("scan_topk") for t in st["tags"])
    col = out["collectives"]
    for name in ("emb_allgather", "bound_allgather", "all_to_all", "merge", "slice_allgather", "key_allgather"):
        assert col[name] > 0, name
    # round 5: where each rank's step goes (what the first curve on real xGMI is to be read with), max and min over ranks
    cp = out["critical_path"]
    assert len(cp["per_rank"]) == 2 and cp["exchange_stream"] is False and cp["step_ms"] == out["ms_per_step"]
    for side in ("max_over_ranks", "min_over_ranks"):
        assert set(cp[side]) == {"encoder", "scan", "collectives", "matcher", "unattributed"}
        assert cp[side]["encoder"] > 0 and cp[side]["scan"] > 0 and cp[side]["collectives"] > 0 and cp[side]["matcher"] > 0
    assert all(cp["max_over_ranks"][kk] >= cp["min_over_ranks"][kk] for kk in cp["max_over_ranks"])
    assert abs(sum(cp["per_rank"][0].values()) - cp["step_ms"]) < 0.01
    # ... and the N = 1 CPU baseline travels with the N > 1 line (timed at N = 1 only: no rank holds the whole db here)
    cb = out["cpu_baseline"]
    assert cb is not None and cb["measured_in_this_run"] is False and cb["value"] > 0 and cb["carried_from"]
    assert "gpus_shared_with_the_bench_ranks" in out["cli"]
    om = out["other_scaling_mode"]
    assert om["scaling"] == "weak" and om["queries_per_step"] == 12 and om["value"] > 0
    cli = out["cli"]
    assert cli["gpus"] == 2 and cli["builder"]["segments"] == 60 * 59 and cli["matcher"]["segments"] == 12 * 19, cli
    assert cli["matcher"]["top1_hit_rate"] >= 0.5


def _clean_env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PFANN_DIST_BACKEND", "PFANN_FORCE_DEVICE")}
    e.update(kw)
    return e


def test_plain_gpus2_on_a_one_gpu_box_fails_loudly_or_runs_two_rccl_ranks():
    """`python bench.py --gpus 2` never reports n_gpus 1: with fewer than 2 devices it refuses (non-zero, no JSON);
    with 2+ devices it really runs two RCCL ranks on two devices."""
    import torch
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--queries", "12", "--db-songs", "600", "--no-cpu-baseline", "--no-cli", "--no-prof", "--max-batch", "512"],
                       capture_output=True, text=True, timeout=900, cwd=REPO, env=_clean_env())
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and not lines, r.stdout[-1000:]
        assert "refusing" in r.stderr
    else:
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["backend"] == "nccl" and len(set(out["devices"])) == 2


def test_self_launched_two_ranks_report_two(tmp_path):
    """the launcher itself on the real kernels: plain `python bench.py --gpus 2` (gloo-staged collectives, both ranks on
    this box's one GPU) -> n_gpus 2, ranks_seen 2."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--queries", "12", "--db-songs", "600", "--no-cpu-baseline", "--no-cli", "--no-prof", "--max-batch", "512"],
                       capture_output=True, text=True, timeout=900, cwd=REPO,
                       env=_clean_env(PFANN_DIST_BACKEND="gloo", PFANN_FORCE_DEVICE="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["backend"] == "gloo" and out["devices"] == [0, 0]


def test_rccl_collectives_of_the_sharded_path_on_one_gpu(tmp_path):
    """RCCL really executes: one rank under torch.distributed.run with the default nccl backend and --force-sharded, so
    the bound all-gather, the all-to-all of the shard lists, the merged-slice all-gathers, the key all-gather, the ragged
    embedding gather, the barrier and the MAX all-reduce all go through RCCL; decisions equal the plain single-GPU run."""
    import numpy as np
    common = ["--steps", "1", "--warmup", "0", "--db-songs", "600", "--queries", "24", "--no-cpu-baseline", "--no-cli", "--no-prof",
              "--max-batch", "512"]
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "rccl.npy")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + common + ["--dump-decisions", one],
                       capture_output=True, text=True, timeout=900, cwd=REPO, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29741", os.path.join(REPO, "bench.py"),
                        "--gpus", "1", "--force-sharded"] + common + ["--dump-decisions", two],
                       capture_output=True, text=True, timeout=900, cwd=REPO, env=_clean_env(MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["backend"] == "nccl" and out["ranks_seen"] == 1 and out["n_gpus"] == 1
    a, b = np.load(one), np.load(two)
    assert np.array_equal(a[:, :2], b[:, :2]) and np.abs(a[:, 2] - b[:, 2]).max() < 1e-6
    # PFANN_EXCHANGE_STREAM=1 (round 5): the same exchange on a stream of its own, three batches two deep, so that batch
    # i+1's encoder really runs beside batch i's collectives -- the same bytes out as with the flag off
    xs_on = str(tmp_path / "xs_on.npy")
    for flag, path in (("1", xs_on),):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                            "--master-addr", "127.0.0.1", "--master-port", "29743", os.path.join(REPO, "bench.py"),
                            "--gpus", "1", "--force-sharded", "--steps", "3", "--warmup", "1", "--db-songs", "600", "--queries", "24",
                            "--no-cpu-baseline", "--no-cli", "--no-alt", "--max-batch", "512", "--dump-decisions", path],
                           capture_output=True, text=True, timeout=900, cwd=REPO,
                           env=_clean_env(MASTER_ADDR="127.0.0.1", PFANN_EXCHANGE_STREAM=flag))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert line["critical_path"]["exchange_stream"] is (flag == "1")
    assert np.array_equal(np.load(xs_on), b)                    # (b: the same exchange on the caller's stream, above)
