"""Rank launcher of the drop-in tools (no torch import here): `PFANN_GPUS=N python matcher.py ...` starts N ranks of the
very same command, one per GPU, and returns their exit status.  The ranks meet through torch.distributed's env://
rendezvous on 127.0.0.1 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set here), exactly what
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N` sets up -- which works just as well -- minus the elastic
agent: the launcher itself costs 50 ms instead of 2.5 s (it neither imports torch nor runs a rendezvous store)."""
import ctypes
import os
import signal
import socket
import subprocess
import sys
import time


def hip_device_count():
    """hipGetDeviceCount through torch's own HIP runtime, without importing torch; -1 when that cannot be had."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so") if spec else ""
        if not os.path.exists(hip):
            return -1
        lib = ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
        n = ctypes.c_int(0)
        rc = lib.hipGetDeviceCount(ctypes.byref(n))
        return n.value if rc == 0 else 0
    except (OSError, AttributeError, ImportError, ValueError):
        return -1


def self_launch_if_asked(argv):
    """-> exit status of the N ranks, or None when there is nothing to launch (no PFANN_GPUS, one rank asked for, or this
    process already is a rank).  Refuses to start fewer RCCL ranks than asked for (one device per rank)."""
    want = os.environ.get("PFANN_GPUS", "")
    if not want or "WORLD_SIZE" in os.environ:
        return None
    backend = os.environ.get("PFANN_DIST_BACKEND", "nccl")
    forced = os.environ.get("PFANN_FORCE_SHARDED", "0") not in ("0", "")
    have = None
    if want == "all" or (backend == "nccl" and "PFANN_FORCE_DEVICE" not in os.environ):
        have = hip_device_count()
        if have < 0:                                   # unknown torch layout: ask torch itself
            import torch
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    n = have if want == "all" else int(want)
    if n <= 1 and not forced:
        return None
    n = max(n, 1)
    if have is not None and want != "all" and have < n:
        print("PFANN_GPUS=%d: only %d HIP device(s) visible; an RCCL job needs one device per rank -- refusing to run fewer "
              "ranks than asked for" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(argv[0])] + list(argv[1:])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen(cmd, env=env))

    def stop(*_a):
        for p in procs:
            if p.poll() is None:
                p.terminate()
    old = {sg: signal.signal(sg, stop) for sg in (signal.SIGINT, signal.SIGTERM)}
    try:
        rc = 0
        alive = list(procs)
        t_stop = None
        while alive:
            for p in list(alive):
                r = p.poll()
                if r is not None:
                    alive.remove(p)
                    if r != 0 and rc == 0:             # one rank failed: the others would wait for it forever
                        rc = r
                        stop()
                        t_stop = time.time()
            if alive and t_stop is not None and time.time() - t_stop > 15.0:
                for p in alive:                        # a rank stuck in a collective may not act on SIGTERM
                    p.kill()
                t_stop = time.time()
            if alive:
                time.sleep(0.01)
        return rc
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
