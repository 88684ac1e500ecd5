#!/usr/bin/env python3
"""out-proj + residual followed by LayerNorm: the two-launch form (w8 GEMM + layernorm kernel) against the fused
whole-row kernel (gemm_rowln.hip), at the audio encoder's shape (M = 126976, N = K = 768)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 126976, 768, 768
a = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
bias = torch.randn(N, device=dev)
g = torch.ones(N, device=dev)
b = torch.zeros(N, device=dev)
x = torch.randn(M, N, device=dev)
h = torch.empty(M, N, dtype=torch.bfloat16, device=dev)


def two():
    lib.caco_op_gemm_bf16_f32out(p(a), p(w), p(bias), p(x), M, N, K, p(x), st)
    lib.caco_op_layernorm(p(x), p(g), p(b), M, N, 1e-5, None, p(h), st)


def fused():
    lib.caco_op_gemm_resid_ln(p(a), p(w), p(bias), p(x), M, N, K, p(g), p(b), 1e-5, p(h), st)


for name, fn in (("w8 GEMM + layernorm", two), ("fused whole-row kernel", fused), ("w8 GEMM + layernorm", two), ("fused whole-row kernel", fused)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:24s} {ms * 1e3:8.1f} us")
