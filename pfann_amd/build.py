"""Builds pfann_amd/libpfann_amd.so (HIP, gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpfann_amd.so")
SOURCES = ["api.hip", "mel.hip", "encoder.hip", "encoder_fused.hip", "search.hip", "search_f16.hip", "rerank.hip", "wavio.hip"]
HEADERS = [os.path.abspath(__file__), os.path.join(CSRC, "common.h"), os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "search_common.h"),
           os.path.join(HERE, "..", "include", "pfann_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"] + os.environ.get("PFANN_HIPCC_FLAGS", "").split()
# mel.hip without the SLP vectoriser, i.e. without packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32:
# 125 -> 11 in melspec_kernel).  Round 5 (profiles/r5/NOTES.md, tools/ubench/xs_race_probe5.py): a melspec_kernel built WITH
# them returns wrong FFT bins for a handful of windows per launch whenever the batched fp16 scan (v_mfma_f32_32x32x16_f16)
# runs on another stream and shares compute units with it -- torch.fft (rocFFT, the same instruction mix) is perturbed the
# same way, sorts / softmax / GEMMs are not; built without them it is bit-stable beside the scan, and 2 % FASTER alone
# (1.167 vs 1.196 ms per 9728 windows).  tests/test_gpu_parity.py::test_melspec_is_bit_stable_beside_a_batched_search.
# search_f16.hip with -fno-honor-nans: the batched scan tests every 32x32 score block with maxima of MFMA results; with NaNs
# honoured each fmaxf operand is first canonicalised (v_max_f32 x, x): 18 instructions per block instead of 10.  Full pass
# 2.607 -> 2.548 ms, sampled pass (a running maximum per register) 0.617 -> 0.545 ms per 9728 x 1 M rows, same results
# (profiles/r5/scan_nnan_ab.txt).  Scores are inner products of finite fp16 rows; +-INFINITY (thresholds, initial maxima)
# is still honoured.
UNIT_FLAGS = {"mel.hip": ["-fno-slp-vectorize"], "search_f16.hip": ["-fno-honor-nans"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc] + FLAGS + UNIT_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
    with ThreadPoolExecutor(max_workers=5) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
