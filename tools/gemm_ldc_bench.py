#!/usr/bin/env python3
"""Does the output row stride matter?  The QKV shape (N = 2304) with padded leading dimensions of the output.
   python tools/gemm_ldc_bench.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import _lib
lib = _lib.load()
lib.caco_set_gemm_tile(8256)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, K = 128000, 768
A = torch.randn(M, K, device="cuda").bfloat16()
for N, ldcs in ((2304, (2304, 2432, 2560, 3072)), (3072, (3072, 3200)), (1536, (1536, 2048)), (2048, (2048,)), (2560, (2560,))):
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    for ldc in ldcs:
        out = torch.empty(M, ldc, dtype=torch.bfloat16, device="cuda")
        run = lambda: lib.caco_op_gemm_bf16_strided(p(A), K, p(W), K, p(bias), M, N, K, 0, p(out), ldc, st)
        for _ in range(3): assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"N={N} ldc={ldc}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s  ({ms*1e3/((M+255)//256*(N//256)/256):.2f} us per tile round)")

print("-- N = 768 bf16 output, output / operand strides")
N = 768
W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
bias = torch.randn(N, device="cuda")
for lda, ldc in ((768, 768), (768, 1024), (768, 1280), (1024, 768), (832, 768), (1024, 1024)):
    Ap = torch.randn(M, lda, device="cuda").bfloat16()
    out = torch.empty(M, ldc, dtype=torch.bfloat16, device="cuda")
    run = lambda: lib.caco_op_gemm_bf16_strided(p(Ap), lda, p(W), K, p(bias), M, N, K, 0, p(out), ldc, st)
    for _ in range(3): assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"N={N} lda={lda} ldc={ldc}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
print("-- K = 3072 operand stride (fc2's A = the MLP hidden rows)")
K2 = 3072
W2 = (torch.randn(N, K2, device="cuda") / K2 ** 0.5).bfloat16()
for lda in (3072, 3200, 3328):
    Ap = torch.randn(M, lda, device="cuda").bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    run = lambda: lib.caco_op_gemm_bf16_strided(p(Ap), lda, p(W2), K2, p(bias), M, N, K2, 0, p(out), N, st)
    for _ in range(3): assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"N={N} K={K2} lda={lda}: {ms*1e3:7.1f} us  {2.0*M*N*K2/ms/1e9:7.1f} TFLOP/s")
