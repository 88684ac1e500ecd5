#!/usr/bin/env python3
"""Does a working set that fits the 256 MiB Infinity Cache cost less time / power than one that streams from HBM?
Device-to-device copies (read X/2, write X/2) of growing working sets X, with socket power and clock sampled.
Calibration for DESIGN.md 4 (energy per byte).   python tools/mall_power.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from power_probe import measure

for mb in (8, 16, 32, 64, 128, 192, 256, 512, 2048):
    n = (mb << 20) // 2 // 4
    a = torch.randn(n, device="cuda")
    b = torch.empty_like(a)
    reps = max(1, 2048 // mb)
    def run():
        for _ in range(reps):
            b.copy_(a)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    measure(f"copy working set {mb:5d} MB ({mb * reps / dt / 1e6 * 1.048576:6.2f} TB/s r+w)", run, 2.0, chunk=5)
