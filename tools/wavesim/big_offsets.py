#!/usr/bin/env python3
"""Buffers larger than 2^31 bytes on the simulator: the persistent GEMM's output (M = 400 000 x N = 3072 bf16 = 2.46 GB), the
attention kernels' qkv (2.2 GB, both kernels, causal and not), LayerNorm's fp32 input (2.46 GB).  Rows on both sides of the
2^31-byte line are compared with torch.  Needs ~8 GB of memory and ~5 minutes; not part of the test suite.
    python tools/wavesim/big_offsets.py"""
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
import simlib  # noqa: E402

sim = simlib.load()
P = simlib.ptr
g = torch.Generator().manual_seed(0)
ok = True


def report(what, err, bar):
    global ok
    good = err < bar
    ok = ok and good
    print(f"   {what}: max err {err:.4f} {'ok' if good else 'TOO LARGE'}")


M, N, K = 400000, 3072, 64
a = torch.randn(M, K, generator=g).bfloat16()
w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
b = torch.randn(N, generator=g)
o = torch.zeros(M, N, dtype=torch.bfloat16)
t = time.time()
simlib.check(sim.caco_op_gemm_bf16(P(a), P(w), P(b), M, N, K, 0, P(o), None))
print(f"gemm [{M}, {K}] x [{N}, {K}]^T -> {o.numel() * 2 / 1e9:.2f} GB ({time.time() - t:.0f} s)")
for r0 in (0, 349000, 349600, M - 300):            # 2^31 bytes = row 349 525
    report(f"rows {r0}..", (o[r0:r0 + 300].float() - (a[r0:r0 + 300].float() @ w.float().T + b)).abs().max().item(), 0.05)
del a, o

for (B, S, heads, hd, causal, small) in ((7400, 64, 8, 96, 0, "0"), (15000, 32, 12, 64, 1, "0"), (15000, 32, 12, 64, 1, "1")):
    os.environ["CACO_ATTN_SMALL"] = small
    simlib.sync_switches()
    H = heads * hd
    qkv = torch.randn(B * S, 3 * H, generator=g).bfloat16().view(B, S, 3 * H)
    mask = torch.ones(B, S)
    mask[-1, S - 5:] = 0
    out = torch.zeros(B, S, H, dtype=torch.bfloat16)
    t = time.time()
    simlib.check(sim.caco_op_attention(P(qkv), 3 * H, H, 2 * H, P(mask), B, S, heads, hd, causal, P(out), None))
    print(f"attention B={B} S={S} heads={heads}x{hd} causal={causal} CACO_ATTN_SMALL={small}: qkv {qkv.numel() * 2 / 1e9:.2f} GB ({time.time() - t:.0f} s)")
    for c in (0, B // 2, B - 1):
        q, k, v = [x.float().view(S, heads, hd).transpose(0, 1) for x in qkv[c].split(H, dim=-1)]
        sc = q @ k.transpose(1, 2) / math.sqrt(hd)
        m = mask[c][None, None, :].bool().expand(heads, S, S).clone()
        if causal:
            m &= torch.tril(torch.ones(S, S, dtype=torch.bool))[None]
        ref = (torch.softmax(sc.masked_fill(~m, float("-inf")), -1) @ v).transpose(0, 1).reshape(S, H)
        nv = int(mask[c].sum())
        report(f"clip {c}", (out[c, :nv].float() - ref[:nv]).abs().max().item(), 0.03)
    del qkv, out
os.environ.pop("CACO_ATTN_SMALL", None)
simlib.sync_switches()

rows, dim = 800000, 768
x = torch.randn(rows, dim, generator=g)
gm, bt = torch.randn(dim, generator=g), torch.randn(dim, generator=g)
o16 = torch.zeros(rows, dim, dtype=torch.bfloat16)
t = time.time()
simlib.check(sim.caco_op_layernorm(P(x), P(gm), P(bt), rows, dim, 1e-5, None, P(o16), None))
print(f"layernorm [{rows}, {dim}] fp32 = {x.numel() * 4 / 1e9:.2f} GB ({time.time() - t:.0f} s)")
for r0 in (0, 699000, 699100, rows - 100):          # 2^31 bytes = row 699 050
    ref = torch.nn.functional.layer_norm(x[r0:r0 + 100], (dim,), gm, bt, 1e-5)
    report(f"rows {r0}..", (o16[r0:r0 + 100].float() - ref).abs().max().item(), 0.07)
print("BIG OFFSETS OK" if ok else "BIG OFFSETS FAILED")
sys.exit(0 if ok else 1)
