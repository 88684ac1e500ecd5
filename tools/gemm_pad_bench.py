#!/usr/bin/env python3
"""Does padding the leading dimensions change the global->LDS load rate?  (L2 channel striding experiment)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import _lib
lib = _lib.load(); lib.caco_set_gemm_tile(2256)
p = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in ((128000, 3072, 768), (128000, 768, 3072)):
    for pad_a, pad_w, pad_c in ((0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (32, 32, 0), (128, 128, 0), (192,192,0)):
        lda, ldw, ldc = K + pad_a, K + pad_w, N + pad_c
        A = torch.randn(M, lda, device="cuda").bfloat16(); W = (torch.randn(N, ldw, device="cuda") / K ** 0.5).bfloat16()
        bias = torch.randn(N, device="cuda"); out = torch.empty(M, ldc, dtype=torch.bfloat16, device="cuda")
        run = lambda: lib.caco_op_gemm_bf16_strided(p(A), lda, p(W), ldw, p(bias), M, N, K, 0, p(out), ldc, st)
        for _ in range(3): assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"M={M} N={N} K={K} pad A/W/C={pad_a}/{pad_w}/{pad_c}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)
        del A, W, out
