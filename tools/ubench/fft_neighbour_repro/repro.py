#!/usr/bin/env python
"""fp16-MFMA scan on a side stream vs torch.fft.rfft (rocFFT) on the main stream: does the FFT's output move?

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-honor-nans -shared scan_alone.hip -o libscan_alone.so
    python repro.py [--reps 20] [--serial]

Prints, per repetition, how many of the FFT's output elements differ from the quiet run's (same input, nothing else on
the GPU).  --serial runs the same kernels one after the other on ONE stream (expected: 0 everywhere)."""
import argparse
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def versions():
    out = {"torch": torch.__version__, "hip": str(torch.version.hip), "device": torch.cuda.get_device_name(0)}
    for name, cmd in (("rocm", "cat /opt/rocm/.info/version"), ("kernel", "uname -r"),
                      ("firmware", "rocm-smi --showfwinfo 2>/dev/null | grep -E 'MEC|SMC|RLC|SDMA|PSP|VCN' | head -12"),
                      ("vbios", "rocm-smi --showvbios 2>/dev/null | grep -i vbios | head -2")):
        try:
            out[name] = subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=20).stdout.strip()
        except Exception as x:      # noqa: BLE001
            out[name] = repr(x)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--queries", type=int, default=9728)
    ap.add_argument("--serial", action="store_true")
    a = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(HERE, "libscan_alone.so"))
    lib.scan_alone_launch.restype = ctypes.c_int
    lib.scan_alone_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    for k, v in versions().items():
        print("%-9s %s" % (k, v.replace("\n", "\n          ")))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    db = torch.nn.functional.normalize(torch.randn((a.rows, 128), device=dev, generator=g), dim=1).half().contiguous()
    q = torch.nn.functional.normalize(torch.randn((a.queries, 128), device=dev, generator=g), dim=1).half().contiguous()
    S = 30
    thr = torch.full((a.queries,), 0.33, device=dev)           # ~100 of 1 M random unit rows score above it
    cnt = torch.zeros((a.queries * S,), dtype=torch.int32, device=dev)
    keys = torch.zeros((a.queries * 8192,), dtype=torch.int64, device=dev)
    x = torch.randn((8192, 1024), device=dev, generator=g)
    side = torch.cuda.Stream()

    def scan(stream):
        with torch.cuda.stream(stream):
            cnt.zero_()
        rc = lib.scan_alone_launch(db.data_ptr(), a.rows, q.data_ptr(), a.queries, thr.data_ptr(), cnt.data_ptr(), keys.data_ptr(),
                                   S, ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0, rc
    quiet = torch.fft.rfft(x).clone()
    scan(torch.cuda.current_stream())
    torch.cuda.synchronize()
    print("survivors per query row (mean):", float(cnt.view(a.queries, S).sum(dim=1).float().mean()))
    total = 0
    for rep in range(a.reps):
        if a.serial:
            for _ in range(3):
                scan(torch.cuda.current_stream())
        else:
            side.wait_stream(torch.cuda.current_stream())
            for _ in range(3):
                scan(side)
        got = torch.fft.rfft(x)
        torch.cuda.synchronize()
        neq = got != quiet
        bad = int(neq.sum())
        total += bad
        print("rep %2d: %d of %d FFT output elements differ from the quiet run%s" % (
            rep, bad, got.numel(), "" if bad == 0 else " (max |diff| %.3g, %d of %d rows)" % (
                float((got - quiet).abs().max()), int(neq.any(dim=1).sum()), got.shape[0])))
    print("TOTAL differing elements over %d repetitions: %d  (%s)" % (a.reps, total, "one stream" if a.serial else "two streams"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
