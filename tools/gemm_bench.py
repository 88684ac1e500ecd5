#!/usr/bin/env python3
"""Micro-benchmark of the bf16 GEMM launches the encoders issue at batch 256 (through the C ABI).
   python tools/gemm_bench.py [--iters 20] [--tile 256] [--only fc1]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import _lib  # noqa: E402

SHAPES = {  # name: (M, N, K, kind, act)
    "qkv": (126976, 2304, 768, "bf16", 0), "out": (126976, 768, 768, "f32r", 0),
    "fc1": (126976, 3072, 768, "bf16", 1), "fc2": (126976, 768, 3072, "f32r", 0), "embed": (126976, 768, 256, "f32", 0),
    "t_qkv": (8192, 2304, 768, "bf16", 0), "t_out": (8192, 768, 768, "f32r", 0), "t_fc1": (8192, 3072, 768, "bf16", 2), "t_fc2": (8192, 768, 3072, "f32r", 0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tile", type=int, default=256)
    ap.add_argument("--only", default="")
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "const"],
                    help="operand values: the sustained clock depends on bit toggling (DVFS), see DESIGN.md 4.1")
    a = ap.parse_args()
    lib = _lib.load()
    lib.caco_set_gemm_tile(a.tile)
    dev = "cuda:0"
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, (M, N, K, kind, act) in SHAPES.items():
        if a.only and name not in a.only.split(","):
            continue
        A = (torch.randn(M, K, device=dev)).bfloat16()
        W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        bias = torch.randn(N, device=dev)
        if a.data == "zeros":
            A.zero_(); W.zero_()
        elif a.data == "const":
            A.fill_(1.0); W.fill_(1.0 / K)
        if kind == "bf16":
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            run = lambda: lib.caco_op_gemm_bf16(p(A), p(W), p(bias), M, N, K, act, p(out), st)
        else:
            out = torch.randn(M, N, device=dev)
            res = out if kind == "f32r" else None
            run = lambda: lib.caco_op_gemm_bf16_f32out(p(A), p(W), p(bias), p(res), M, N, K, p(out), st)
        for _ in range(3):
            assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f"{name:6s} M={M} N={N} K={K} {kind:5s} {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
