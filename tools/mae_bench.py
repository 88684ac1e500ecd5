#!/usr/bin/env python3
"""AudioMAE stage-1 forward (BASELINE configs[4]: batch 256, 100 visible + 396 restored patches, encoder + decoder)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import config as C, synth  # noqa: E402
from cacophony_amd.model import AudioMAE  # noqa: E402

B, V, R = 256, 100, 396
cfg = C.AudioMAEConfig(C.default_audio_config(), C.default_audio_config())
state = synth.make_audiomae_state(C.default_audio_config(), C.default_audio_config())
m = AudioMAE(cfg, device="cuda:0").load_state_dict(state)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, V, 256, generator=g).cuda()
perm = torch.stack([torch.randperm(496, generator=g) for _ in range(B)])
vis, res = perm[:, :V].sort(1).values, perm[:, V:].sort(1).values
f = lambda t: t.float().cuda()
args = (x, f(torch.ones(B, V)), f(vis // 8), f(vis % 8), f(res // 8), f(res % 8), f(torch.ones(B, R)))
for _ in range(2):
    y = m.forward(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    y = m.forward(*args)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"AudioMAE stage-1 forward B={B} V={V} R={R}: {ms:.2f} ms/step, {B / ms * 1e3:.0f} clips/s, out {tuple(y.shape)}, finite {bool(torch.isfinite(y).all())}")
print(f"algorithmic 111.0 GFLOP/clip -> {111.0 * B / ms:.0f} TFLOP/s")
