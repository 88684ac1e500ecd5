#!/usr/bin/env python3
"""Per-K-tile cycle stamps of the skewed-row-group GEMM (build: tools/build_variant.sh s8time gemm_s8.hip -DS8_TIMING;
run with CACO_LIB_PATH=cacophony_amd/_variants/libcaco_hip_s8time.so): cycles of each of the first 63 K-tiles of a
workgroup's run (K-tiles 0-3 of a period carry a row group's epilogue as filler)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cacophony_amd import _lib  # noqa: E402

SHAPES = {"qkv": (126976, 2304, 768, 0), "fc1": (126976, 3072, 768, 1)}
lib = _lib.load()
lib.caco_set_gemm_tile(6256)
dev = "cuda:0"
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else SHAPES):
    M, N, K, act = SHAPES[name]
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    buf = torch.zeros(M * N * 2 + 256 * 64 * 8, dtype=torch.uint8, device=dev)
    run = lambda: lib.caco_op_gemm_bf16(p(A), p(W), p(bias), M, N, K, act, p(buf), st)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    d = buf[M * N * 2:].view(torch.int64).cpu().view(256, 64).double()
    d = d[d[:, 0] > 0]
    dt = (d[:, 1:] - d[:, :-1])
    nk = K // 64
    print(f"{name}: {d.shape[0]} workgroups; mean cycles per K-tile (period 0 | 1 | 2 ...), K-tiles 0-3 carry a drain from period 1 on")
    for per in range(min(5, 63 // nk)):
        print("  period", per, " ".join(f"{dt[:, per * nk + k].mean():6.0f}" for k in range(nk) if per * nk + k < 63))
