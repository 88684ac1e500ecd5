#!/usr/bin/env python3
"""Phase timeline of the persistent w8 GEMM, from s_memtime stamps (build from the parked 32x32x16 form, see the header of tools/experimental/gemm_w8_mf32.hip: build_variant.sh w8time gemm_w8.hip -DW8_MF32 -DW8_TIMING;
run with CACO_LIB_PATH=cacophony_amd/_variants/libcaco_hip_w8time.so).  Prints, per shape, the mean cycles per tile of
  K-loop (first barrier -> K-loop done) | epilogue issue | wait at the post-epilogue barrier | next tile's first K-tile incl. store drain
and the spread of the workgroups' epilogue start times."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import _lib  # noqa: E402

SHAPES = {"qkv": (126976, 2304, 768, "bf16", 0), "out": (126976, 768, 768, "f32r", 0), "fc1": (126976, 3072, 768, "bf16", 1),
          "fc2": (126976, 768, 3072, "f32r", 0)}
lib = _lib.load()
lib.caco_set_gemm_tile(256)
dev = "cuda:0"
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
NWG, NT = 256, 32
MDIV = int(os.environ.get("W8T_MDIV", "1"))
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else SHAPES):
    M, N, K, kind, act = SHAPES[name]
    M //= MDIV
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    esz = 2 if kind == "bf16" else 4
    tail = NWG * NT * 8 * 8
    buf = torch.zeros(M * N * esz + tail, dtype=torch.uint8, device=dev)
    if kind == "bf16":
        run = lambda: lib.caco_op_gemm_bf16(p(A), p(W), p(bias), M, N, K, act, p(buf), st)
    else:
        run = lambda: lib.caco_op_gemm_bf16_f32out(p(A), p(W), p(bias), p(buf), M, N, K, p(buf), st)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    buf[M * N * esz:].zero_()
    assert run() == 0
    torch.cuda.synchronize()
    d = buf[M * N * esz:].view(torch.int64).cpu().view(NWG, NT, 8).double()
    ok = d[:, :, 3] > 0
    tiles = ok.sum(1)
    kl = (d[:, :, 1] - d[:, :, 0])[ok]
    ep = (d[:, :, 2] - d[:, :, 1])[ok]
    bw = (d[:, :, 3] - d[:, :, 2])[ok]
    nxt = d[:, 1:, 0] - d[:, :-1, 3]
    nxt = nxt[ok[:, 1:] & (d[:, 1:, 0] > 0)]
    whole = d[:, 1:, 1] - d[:, :-1, 1]
    whole = whole[ok[:, 1:]]
    t0 = d[:, 0, 0].min()
    print(f"{name}: tiles/WG {tiles.min().item():.0f}-{tiles.max().item():.0f}; s_memtime ticks (100 MHz) per tile: "
          f"K-loop after 1st barrier {kl.mean():.0f}  epilogue {ep.mean():.0f}  barrier wait {bw.mean():.0f}  "
          f"next first K-tile (+store drain) {nxt.mean():.0f}  tile period {whole.mean():.0f}")
    for t in range(min(4, int(tiles.min().item()))):
        e = d[:, t, 1] - t0
        print(f"   tile {t}: K-loop end over WGs: min {e.min():.0f} median {e.median():.0f} max {e.max():.0f}")
