"""bench.py's rank launcher (CPU, gloo): `python bench.py --gpus N` must really run N ranks, and no path may print an
`n_gpus` that differs from --gpus.  --launch-check stops after the ranks have met in the process group."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PFANN_DIST_BACKEND")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=600, cwd=REPO, env=e)


def _json(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[0])


def test_plain_invocation_spawns_the_ranks_it_was_asked_for():
    r = _run(["--gpus", "2", "--launch-check"], PFANN_DIST_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json(r)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["backend"] == "gloo"
    assert len(out["devices"]) == 2 and out["all_reduce_of_ones"] == 2
    # stdout is the JSON line and NOTHING else: gloo's "[Gloo] Rank 0 is connected to ..." chatter goes to stderr
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout


def test_world_size_mismatch_is_fatal():
    r = _run(["--gpus", "2", "--launch-check"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "{" not in r.stdout
    r = _run(["--gpus", "1", "--launch-check"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "{" not in r.stdout


def test_rccl_job_without_enough_devices_refuses_instead_of_running_one_rank():
    """here: 0 devices visible; on the 1-GPU box the GPU test asks for 2."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(have + 1) if have else "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "{" not in r.stdout, r.stdout[-500:]
    assert "refusing" in r.stderr
