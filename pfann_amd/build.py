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
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "search_common.h"),
           os.path.join(HERE, "..", "include", "pfann_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"] + os.environ.get("PFANN_HIPCC_FLAGS", "").split()


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
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

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
