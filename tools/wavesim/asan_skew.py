import os, subprocess, sys
import os; HERE=os.path.dirname(os.path.abspath(__file__)); REPO=os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import build_sim
lib = build_sim.build(asan=True, defines=("-DW8_F32_SKEW",), tag="skew", verbose=False)
rt = subprocess.run([build_sim.CXX, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
if not os.path.exists(rt):
    rt = subprocess.run([build_sim.CXX, "-print-file-name=libclang_rt.asan.so"], capture_output=True, text=True).stdout.strip()
code = r'''
import ctypes as C, math, sys
import numpy as np
sys.path.insert(0, REPO)
from cacophony_amd import _lib
lib = C.CDLL(LIB)
for name, (res, args) in _lib._SIGNATURES.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
P = lambda a: C.c_void_p(0 if a is None else a.ctypes.data)
def bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)
rng = np.random.default_rng(2)
lib.caco_set_gemm_tile(8256)
# exactly-sized heap buffers (np.empty of the exact shape): any access past row M or before row 0 is a heap overflow
for (M, N, K) in ((2048, 768, 768), (2000, 768, 768), (4100, 768, 1024), (2304, 512, 3072), (1793, 768, 640)):
    a = bf16(rng.standard_normal((M, K))); w = bf16(rng.standard_normal((N, K)) / math.sqrt(K))
    b = rng.standard_normal(N).astype(np.float32); x = rng.standard_normal((M, N)).astype(np.float32)
    ref = (a.astype(np.uint32) << 16).view(np.float32).astype(np.float64) @ (w.astype(np.uint32) << 16).view(np.float32).astype(np.float64).T + b + x
    rc = lib.caco_op_gemm_bf16_f32out(P(a), P(w), P(b), P(x), M, N, K, P(x), None)
    assert rc == 0, lib.caco_last_error()
    err = np.abs(x - ref).max() / (np.abs(ref).max())
    assert err < 1e-5, (M, N, K, err)
    print("ok", M, N, K, f"{err:.2e}")
print("DRIVER DONE")
'''
env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:detect_stack_use_after_return=0", OMP_NUM_THREADS="1")
r = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\nLIB = {lib!r}\n" + code], env=env, capture_output=True, text=True)
print(r.stdout[-1500:]); print(r.stderr[-1500:])
print("ASAN CLEAN" if ("DRIVER DONE" in r.stdout and "AddressSanitizer" not in r.stderr) else "ASAN REPORTS OR FAILURE")
