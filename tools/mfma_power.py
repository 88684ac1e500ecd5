#!/usr/bin/env python3
"""Energy of the matrix pipe by MFMA shape and wave tile (tools/probes/mfma_power.hip): TFLOP/s, socket power and shader
clock on random / zero operand bits.  Calibration for DESIGN.md 4 (power cap).   python tools/mfma_power.py [--seconds 3]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from power_probe import measure  # noqa: E402

NAMES = {0: "32x32x16 wave 128x64  8 waves/CU", 1: "16x16x32 wave 128x64  8 waves/CU",
         2: "16x16x32 wave 128x128 4 waves/CU", 3: "32x32x16 wave 128x128 4 waves/CU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--variants", default="0,1,2,3")
    a = ap.parse_args()
    so = os.path.join(HERE, "probes", "libmfma_power.so")
    src = os.path.join(HERE, "probes", "mfma_power.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
    lib = C.CDLL(so)
    lib.mfma_power_run.restype = C.c_double
    lib.mfma_power_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    out = torch.zeros(4096, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for v in [int(x) for x in a.variants.split(",")]:
        for data in ("randn", "zeros"):
            seed = torch.randn(1 << 23, device="cuda").bfloat16()
            if data == "zeros":
                seed.zero_()
            iters = 4000
            fl = lib.mfma_power_run(v, seed.data_ptr(), out.data_ptr(), iters, 256, st)
            measure(f"{NAMES[v]} {data}", lambda: lib.mfma_power_run(v, seed.data_ptr(), out.data_ptr(), iters, 256, st), a.seconds,
                    flops=fl, chunk=10)


if __name__ == "__main__":
    main()
