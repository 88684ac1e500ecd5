#!/usr/bin/env python3
"""Step timeline of the ping-pong attention kernel from s_memtime stamps.
   build (after pasting tools/experimental/attention_pp.hip into attention.hip): tools/build_variant.sh pptime attention.hip -DATTN_PP_TIMING
   run:   CACO_LIB_PATH=cacophony_amd/_variants/libcaco_hip_pptime.so python tools/experimental/attn_timing.py
Every wave stamps the clock before and after each barrier; the first word is HW_ID (SIMD = bits 4-5, CU = 8-11)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cacophony_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, S, heads, hd = 256, 496, 8, 96
H = heads * hd
qkv = torch.randn(B * S, 2560, device=dev).bfloat16()
km = torch.ones(B, S, device=dev)
NWG = B * heads
tail = NWG * 8 * 96 * 8
buf = torch.zeros(B * S * H * 2 + tail, dtype=torch.uint8, device=dev)
run = lambda: lib.caco_op_attention(p(qkv), 2560, H, 2 * H, p(km), B, S, heads, hd, 0, p(buf), st)
for _ in range(3):
    assert run() == 0
torch.cuda.synchronize()
buf[B * S * H * 2:].zero_()
assert run() == 0
torch.cuda.synchronize()
st_ = buf[B * S * H * 2:].view(torch.int64).cpu().numpy().reshape(NWG, 8, 96)[:256]
hw = st_[:, :, 0]
print("SIMD id of waves 0..7 (first 4 workgroups):")
for w in range(4):
    print("  wg", w, [(int(h) >> 4) & 3 for h in hw[w]], "cu", [(int(h) >> 8) & 15 for h in hw[w]])
t = st_[:, :, 1:].astype(np.float64)
n = int((t[0, 0] > 0).sum())
t = t[:, :, :n]
# stamps come in pairs (before barrier, after barrier); work = after(k) -> before(k+1), wait = before(k) -> after(k)
before, after = t[:, :, 0::2], t[:, :, 1::2]
nb = before.shape[2]
wait = after - before
work = before[:, :, 1:] - after[:, :, :-1]
for g, name in ((slice(0, 4), "group A (waves 0-3)"), (slice(4, 8), "group B (waves 4-7)")):
    wk, wt = work[:, g].mean(axis=(0, 1)), wait[:, g].mean(axis=(0, 1))
    print(name, "barriers", nb)
    print("  work  between barriers (cycles x 100MHz-clock units):", np.round(wk[:12], 0), "... mean", round(float(wk.mean()), 1))
    print("  wait  at barriers:", np.round(wt[:12], 0), "... mean", round(float(wt.mean()), 1))
tot = (t[:, :, -1] - t[:, :, 0]).mean()
print("first barrier -> last barrier: %.0f ticks (s_memtime ticks at 100 MHz: x%.1f shader cycles at 2.4 GHz)" % (tot, 24.0))
