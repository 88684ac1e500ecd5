#!/usr/bin/env python3
"""Calibration only (never on the product path): the vendor library's bf16 GEMM on an encoder shape, for counter comparisons
against gemm_bf16_w8 (bash tools/pmc_run.sh <tag> Cijk -- python tools/blaslt_gemm.py --only fc1)."""
import argparse
import torch

SHAPES = {"qkv": (126976, 2304, 768), "fc1": (126976, 3072, 768), "fc2": (126976, 768, 3072), "out": (126976, 768, 768)}
ap = argparse.ArgumentParser()
ap.add_argument("--only", default="fc1")
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
M, N, K = SHAPES[a.only]
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(a.iters):
    torch.matmul(A, W.t(), out=out)
torch.cuda.synchronize()
