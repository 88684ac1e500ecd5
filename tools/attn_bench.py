#!/usr/bin/env python3
"""Micro-benchmark of the fused attention kernel at the encoder shapes (through the C ABI).
   python tools/attn_bench.py [--iters 20] [--only audio]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import _lib  # noqa: E402

SHAPES = {"audio": (256, 500, 8, 96, 0), "audio30s": (64, 1500, 8, 96, 0), "text": (256, 32, 12, 64, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lib = _lib.load()
    dev = "cuda:0"
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, (B, S, heads, hd, causal) in SHAPES.items():
        if a.only and name not in a.only.split(","):
            continue
        H = heads * hd
        qkv = torch.randn(B, S, 3 * H, device=dev).bfloat16()
        mask = torch.ones(B, S, device=dev)
        mask[:, S - 4:] = 0
        out = torch.empty(B, S, H, dtype=torch.bfloat16, device=dev)
        run = lambda: lib.caco_op_attention(p(qkv), 3 * H, H, 2 * H, p(mask), B, S, heads, hd, causal, p(out), st)
        for _ in range(3):
            assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        fl = 4.0 * B * S * S * H * (0.5 if causal else 1.0)
        print(f"{name:9s} B={B} S={S} {heads}x{hd} causal={causal} {ms * 1e3:8.1f} us  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
