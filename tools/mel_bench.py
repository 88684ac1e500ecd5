#!/usr/bin/env python3
"""Time the fused mel front end at the bench shape (256 clips x 10 s) through the product API."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import frontend  # noqa: E402

wav = torch.randn(256, 160000, device="cuda") * 0.1
for dt in (torch.bfloat16, torch.float32):
    for _ in range(3):
        frontend.mel_patches_device(wav, 500, dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        frontend.mel_patches_device(wav, 500, dt)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    by = 256 * (160000 * 4 + 496 * 256 * (2 if dt == torch.bfloat16 else 4))
    print(f"mel patches {dt}: {ms * 1e3:.1f} us  {by / ms / 1e6:.0f} GB/s (incl. patch-meta kernel and torch allocation)")
