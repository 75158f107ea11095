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


def _device():
    return int(os.environ.get("PFANN_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def _work(paths):
    try:
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7): it has to be the one in the
        # process BEFORE libpfann_amd.so is loaded, or the library binds to /opt/rocm's copy and the process ends up with
        # two runtimes (torch then sees no device).  find_spec does not import torch.
        import importlib.util
        spec = importlib.util.find_spec("torch")
        hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so") if spec else ""
        if not os.path.exists(hip):
            return                                    # an unknown torch layout: no prewarm rather than a second runtime
        ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
        lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpfann_amd.so"))
        lib.pfann_prewarm.argtypes = [ctypes.c_int]
        lib.pfann_prewarm.restype = ctypes.c_int
        lib.pfann_prewarm(_device())                  # no device / no library: the tool itself will say so, loudly
    except (OSError, AttributeError, ImportError, ValueError):
        pass
    for p in paths:                                   # model.pt / landmarkValue: into the page cache
        try:
            with open(p, "rb", buffering=0) as f:
                while f.read(1 << 24):
                    pass
        except OSError:
            pass


def start(paths=()):
    """Idempotent; does nothing when this process is only going to launch ranks (PFANN_GPUS set, not yet a rank)."""
    global _thread
    if _thread is not None or os.environ.get("PFANN_PREWARM", "1") == "0":
        return
    if os.environ.get("PFANN_GPUS") and "WORLD_SIZE" not in os.environ:
        return
    _thread = threading.Thread(target=_work, args=(list(paths),), name="pfann-prewarm", daemon=True)
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
    if os.environ.get("PFANN_FAST_EXIT", "1") == "0":
        sys.exit(rc)
    os._exit(int(rc or 0))
