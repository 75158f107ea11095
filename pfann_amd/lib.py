"""ctypes binding of libpfann_amd.so (include/pfann_amd.h).  There is NO CPU fallback:
if the library is missing or no MI355X is visible, the product path raises."""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int16, c_int32, c_int64,
                    c_longlong, c_void_p)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpfann_amd.so")


class PfannError(RuntimeError):
    pass


class Config(Structure):
    _fields_ = [("segment_len", c_int32), ("stft_n", c_int32), ("stft_hop", c_int32), ("n_mels", c_int32),
                ("power", c_int32), ("pad_reflect", c_int32), ("log_mode", c_int32),
                ("spec_norm_max", c_int32), ("log_eps", c_float),
                ("d", c_int32), ("h", c_int32), ("u", c_int32), ("fuller", c_int32),
                ("activation", c_int32), ("relu_after_bn", c_int32),
                ("stride_t", c_int32 * 8), ("stride_f", c_int32 * 8), ("max_batch", c_int32)]


class MatchResult(Structure):
    _fields_ = [("song", c_int32), ("offset", c_int32), ("shift", c_int32), ("n_cand", c_int32),
                ("score", c_double)]


class WavInfo(Structure):
    _fields_ = [("n_frames", c_int64), ("data_pos", c_int64), ("n_ch", c_int32), ("sample_rate", c_int32),
                ("status", c_int32), ("reserved", c_int32)]


# every symbol include/pfann_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "version": (c_longlong, []),
    "seq_score": (c_int, [c_void_p, POINTER(c_int64), c_int, POINTER(c_float), c_int, POINTER(c_int64),
                          c_int, POINTER(c_float), c_int, c_float]),
    "pfann_last_error": (c_char_p, []),
    "pfann_create": (c_void_p, [POINTER(Config), c_int]),
    "pfann_destroy": (None, [c_void_p]),
    "pfann_set_melbank": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "pfann_load_weight": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "pfann_weights_missing": (c_int, [c_void_p]),
    "pfann_melspec": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "pfann_encode": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "pfann_segment_embed": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p]),
    "pfann_segment_embed_at": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "pfann_pcm16_to_mono": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "pfann_pcm16_files_to_mono": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64), c_int, c_void_p, c_int64,
                                          c_void_p, c_void_p]),
    "pfann_wav_probe": (c_int, [POINTER(c_char_p), c_int, c_int, POINTER(WavInfo)]),
    "pfann_wav_read": (c_int, [POINTER(c_char_p), c_int, c_int, POINTER(WavInfo), POINTER(c_int64), c_void_p, c_int64]),
    "pfann_resample_to_mono": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int64,
                                       c_void_p, c_void_p, c_void_p]),
    "pfann_debug_activation": (c_int64, [c_void_p, c_int, c_int64, c_void_p, c_int64]),
    "pfann_debug_keep": (None, [c_void_p, c_int]),
    "pfann_set_fused_layernorm": (c_int, [c_void_p, c_int]),
    "pfann_set_encoder_precision": (c_int, [c_void_p, c_int]),
    "pfann_prewarm": (c_int, [c_int]),
    "pfann_set_plan_batch": (c_int64, [c_void_p, c_int64]),
    "pfann_set_streams": (c_int, [c_void_p, c_int]),
    "pfann_db_create": (c_void_p, [c_int, c_int]),
    "pfann_db_destroy": (None, [c_void_p]),
    "pfann_db_set_prefilter": (c_int, [c_void_p, c_int]),
    "pfann_db_set_storage": (c_int, [c_void_p, c_int]),
    "pfann_db_dim": (c_int, [c_void_p]),
    "pfann_db_ntotal": (c_int64, [c_void_p]),
    "pfann_db_bytes": (c_int64, [c_void_p]),
    "pfann_db_load": (c_int, [c_void_p, c_void_p, c_int, c_int64, POINTER(c_int64), c_int, c_int64]),
    "pfann_search_topk": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "pfann_search_bound": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "pfann_search_topk_bounded": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pfann_topk_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p]),
    "pfann_bound_reduce": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "pfann_topk_merge_lists": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "pfann_match": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int,
                            c_float, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pfann_db_owned_songs": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "pfann_db_set_owned_songs": (c_int, [c_void_p, c_int, c_int]),
    "pfann_song_scores_to_seconds": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_double, c_int, c_void_p]),
    "pfann_match_pack": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "pfann_match_pick": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "pfann_prof_enable": (None, [c_int]),
    "pfann_prof_reset": (None, []),
    "pfann_prof_marker": (None, [c_void_p]),
    "pfann_prof_elapsed_ms": (c_double, [c_char_p, POINTER(c_int64)]),
    "pfann_prof_work": (c_double, [c_char_p]),
    "pfann_prof_tags": (c_int, [c_char_p, c_int]),
}

_lib = None


def load():
    """Loads the shared library and binds every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PfannError("%s not found: build it with `python -m pfann_amd.build` "
                         "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().pfann_last_error()
    return msg.decode("utf8", "replace") if msg else ""


def check(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise PfannError("%s failed (%s): %s" % (what, rc, last_error()))
    return rc


def current_stream_ptr(device=None):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def upload_async(arr, device, dtype):
    """Small host array -> device tensor WITHOUT making the host wait for the stream: staged through torch's caching
    pinned allocator.  A copy from pageable memory blocks the calling thread until everything queued before it on
    the stream has run, which would serialise the host-side pipeline (decode, launches, result writing) with the GPU."""
    import numpy as np
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype))
    return t.pin_memory().to(device, non_blocking=True)


_hip = None


def pinned_int16(n):
    """-> torch int16 tensor of n elements in PINNED host memory, allocated by hipHostMalloc through ctypes: the call
    releases the GIL.  `torch.empty(n, pin_memory=True)` holds it for the whole allocation -- 30-40 ms per 100 MB slab --
    and when the tools' loader thread took its first slabs that way, the thread that feeds the GPU stood still for 43 ms at
    the start of every run (PFANN_TIMELINE=1).  The memory is never freed: the slab pool lives as long as the process."""
    global _hip
    import torch
    try:
        if _hip is None:
            _hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
            _hip.hipHostMalloc.argtypes = [POINTER(c_void_p), ctypes.c_size_t, ctypes.c_uint]
            _hip.hipHostMalloc.restype = c_int
            _hip.hipHostFree.argtypes = [c_void_p]
            _hip.hipHostFree.restype = c_int
        ptr = c_void_p()
        if _hip.hipHostMalloc(ctypes.byref(ptr), int(n) * 2, 0) != 0 or not ptr.value:
            raise OSError("hipHostMalloc failed")
        t = torch.frombuffer((c_int16 * int(n)).from_address(ptr.value), dtype=torch.int16)
        if not t.is_pinned():
            del t
            _hip.hipHostFree(ptr)              # (the fallback below allocates its own block: this one must not leak)
            raise OSError("the runtime does not report the block as pinned")
        return t
    except (OSError, AttributeError):
        return torch.empty(int(n), dtype=torch.int16, pin_memory=True)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise PfannError("no HIP device visible: the pfann_amd hot path runs only on an MI355X "
                         "(there is no CPU fallback)")
