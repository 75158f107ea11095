"""Start-up overlap for the drop-in tools (no torch import here): the root shims call start() BEFORE they import the
tool's module, so that -- while the interpreter spends its 0.7 s importing torch -- a thread has the HIP runtime
initialise this rank's device and load every code object of libpfann_amd.so (pfann_prewarm: ~0.45 s otherwise paid at
the first launch of each translation unit), and pulls the model file into the page cache.  fast_exit() ends a tool whose
files are closed without tearing down the 29 GB workspace buffer by buffer."""
import ctypes
import os
import sys
import threading

_thread = None
_engine = None          # (handle, bytes of its pfann_config, device) built by the start-up thread, or None
_pinned = []            # [(address, int16 elements)] pinned slabs allocated by the start-up thread for the file loader


def _device():
    return int(os.environ.get("PFANN_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def _engine_job(lib, job):
    """The encoder context with its weights, built without torch: configs.json -> pfann_config (config.py), model.pt ->
    numpy (ptfile.py) -> pfann_load_weight.  Anything unexpected (an unknown pickle, a missing tensor): no context, and
    the tool loads the model the ordinary way."""
    global _engine
    import json
    from . import lib as _l
    from .config import config_from_params
    from .ptfile import load_state_dict_numpy
    import numpy as np
    configs_json, model_pt, max_batch = job
    params = json.load(open(configs_json))
    cfg = config_from_params(params, max_batch)
    sd = load_state_dict_numpy(model_pt)
    lib.pfann_create.restype = ctypes.c_void_p
    lib.pfann_create.argtypes = [ctypes.POINTER(_l.Config), ctypes.c_int]
    lib.pfann_load_weight.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64]
    lib.pfann_weights_missing.argtypes = [ctypes.c_void_p]
    lib.pfann_destroy.argtypes = [ctypes.c_void_p]
    h = lib.pfann_create(ctypes.byref(cfg), _device())
    if not h:
        return
    ok = True
    for name, val in sd.items():
        arr = np.ascontiguousarray(val, dtype=np.float32)
        if lib.pfann_load_weight(h, name.encode(), arr.ctypes.data, arr.size) < 0:
            ok = False
            break
    if ok and lib.pfann_weights_missing(h) == 0:
        _engine = (h, bytes(cfg), _device())
    else:
        lib.pfann_destroy(h)


def slab_cap(n):
    """size class of a launch group's PCM slab (int16 elements): the next multiple of 8 M samples (builder._PinnedPool)"""
    return -(-int(n) // (1 << 23)) * (1 << 23)


def _pinned_job(hiplib, job, count):
    """`count` pinned slabs of the size a full launch group's PCM takes, allocated here -- while torch imports -- because a
    pinned allocation of 84 MB takes 10-40 ms under a lock of the HIP runtime that stalls every launch of the process:
    taken by the loader thread in the middle of the first groups, it cost a run 35-45 ms (PFANN_TIMELINE=1)."""
    import json
    configs_json, _model, max_batch = job
    params = json.load(open(configs_json))
    hop = int(params["sample_rate"] * params["hop_size"]) // max(1, int(params["indexer"].get("frame_shift_mul", 1)))
    seg = int(params["sample_rate"] * params["segment_size"])
    n = slab_cap(int(max_batch * hop * 1.06) + seg)
    hiplib.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hiplib.hipHostMalloc.restype = ctypes.c_int
    for _ in range(count):
        ptr = ctypes.c_void_p()
        if hiplib.hipHostMalloc(ctypes.byref(ptr), n * 2, 0) != 0 or not ptr.value:
            break
        _pinned.append((ptr.value, n))


def take_pinned():
    """-> [(address, int16 elements)] of the slabs the start-up thread allocated (waits for it); the caller owns them."""
    if _thread is None:
        return []
    _thread.join()
    got = list(_pinned)
    del _pinned[:]
    return got


def take_engine(cfg, device):
    """-> the handle of the context the start-up thread built, if it built one for exactly this pfann_config on this
    device (waits for the thread); else None.  The caller owns the handle from then on."""
    global _engine
    if _thread is None:
        return None
    _thread.join()
    got, _engine = _engine, None
    if got is None:
        return None
    h, cfg_bytes, dev = got
    if cfg_bytes == bytes(cfg) and dev == int(device):
        return h
    try:                                              # built for something else (should not happen): give it back
        lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpfann_amd.so"))
        lib.pfann_destroy.argtypes = [ctypes.c_void_p]
        lib.pfann_destroy(h)
    except OSError:
        pass
    return None


def _work(paths, job=None, slabs=0):
    try:
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7): it has to be the one in the
        # process BEFORE libpfann_amd.so is loaded, or the library binds to /opt/rocm's copy and the process ends up with
        # two runtimes (torch then sees no device).  find_spec does not import torch.
        import importlib.util
        spec = importlib.util.find_spec("torch")
        hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so") if spec else ""
        if not os.path.exists(hip):
            return                                    # an unknown torch layout: no prewarm rather than a second runtime
        hiplib = ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
        lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpfann_amd.so"))
        lib.pfann_prewarm.argtypes = [ctypes.c_int]
        lib.pfann_prewarm.restype = ctypes.c_int
        rc = lib.pfann_prewarm(_device())             # no device / no library: the tool itself will say so, loudly
        if rc == 0 and job is not None and os.environ.get("PFANN_PREWARM_ENGINE", "1") != "0":
            try:
                _engine_job(lib, job)
            except Exception:                         # noqa: BLE001 -- any surprise: the ordinary path loads the model
                pass
            try:
                if slabs > 0 and os.environ.get("PFANN_PREWARM_SLABS", "1") != "0":
                    _pinned_job(hiplib, job, slabs)
            except Exception:                         # noqa: BLE001 -- the loader then allocates its slabs itself
                pass
    except (OSError, AttributeError, ImportError, ValueError):
        pass
    for p in paths:                                   # model.pt / landmarkValue: into the page cache
        try:
            with open(p, "rb", buffering=0) as f:
                while f.read(1 << 24):
                    pass
        except OSError:
            pass


def engine_job_for(tool, argv):
    """(configs.json, model.pt, max_batch) of the model the tool is about to load, from its argv (builder.py:38-44,
    matcher.py:44-60, extractemb.py); None when that cannot be told without doing the tool's own work."""
    try:
        mb = int(os.environ.get("PFANN_MAX_BATCH", "9728"))
        if tool in ("matcher", "extractemb") and len(argv) > 2:
            return os.path.join(argv[2], "configs.json"), os.path.join(argv[2], "model.pt"), mb
        if tool == "builder" and len(argv) > 2:
            cfg = argv[3] if len(argv) > 3 else "configs/default.json"
            if os.path.isdir(cfg):
                return os.path.join(cfg, "configs.json"), os.path.join(cfg, "model.pt"), mb
            import json
            return cfg, os.path.join(json.load(open(cfg))["model_dir"], "model.pt"), mb
    except (OSError, ValueError, KeyError):
        pass
    return None


def start(paths=(), engine=None, slabs=3):
    """Idempotent; does nothing when this process is only going to launch ranks (PFANN_GPUS set, not yet a rank).
    engine = (configs.json path, model.pt path, max_batch): also build the encoder context and load its weights, and
    allocate `slabs` pinned PCM slabs for the file loader."""
    global _thread
    if _thread is not None or os.environ.get("PFANN_PREWARM", "1") == "0":
        return
    if os.environ.get("PFANN_GPUS") and "WORLD_SIZE" not in os.environ:
        return
    _thread = threading.Thread(target=_work, args=(list(paths), engine, slabs), name="pfann-prewarm", daemon=True)
    _thread.start()


def fast_exit(rc):
    """Flush what Python buffers, then leave without running destructors: every output file of the tool is closed by
    then, and the driver reclaims the device memory of a dead process at once (freeing the workspace allocation by
    allocation, tearing down torch and unpinning the slabs took 0.4 s of a 1.9 s matcher run)."""
    import logging
    logging.shutdown()
    try:
        sys.stdout.flush()
        sys.stderr.flush()
    except (OSError, ValueError):
        pass
    # a profiler that finalises at exit (rocprofv3 writes its trace database from an atexit hook) must get its normal exit
    profiled = any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER")) for k in os.environ) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")
    if os.environ.get("PFANN_FAST_EXIT", "1") == "0" or profiled:
        sys.exit(rc)
    os._exit(int(rc or 0))
