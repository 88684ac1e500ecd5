import ctypes as C, math, torch, sys, os, time
sys.path.insert(0, os.getcwd())
from cacophony_amd import _lib
lib = _lib.load()
DEV="cuda:0"
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for name,(M,N,K) in {"out":(126976,768,768),"fc2":(126976,768,3072)}.items():
    a = torch.randn(M,K,device=DEV).bfloat16(); w=(torch.randn(N,K,device=DEV)/math.sqrt(K)).bfloat16(); bias=torch.randn(N,device=DEV)
    g = torch.ones(N,device=DEV); b=torch.zeros(N,device=DEV); x=torch.randn(M,N,device=DEV); h=torch.empty(M,N,dtype=torch.bfloat16,device=DEV)
    tk = torch.zeros((M+255)//256,dtype=torch.int32,device=DEV)
    t_g = timeit(lambda: lib.caco_op_gemm_bf16_f32out(p(a),p(w),p(bias),p(x),M,N,K,p(x),st()))
    t_l = timeit(lambda: lib.caco_op_layernorm(p(x),p(g),p(b),M,N,C.c_float(1e-5),None,p(h),st()))
    t_f = timeit(lambda: lib.caco_op_gemm_resid_ln(p(a),p(w),p(bias),p(x),M,N,K,p(g),p(b),C.c_float(1e-5),p(h),p(tk),st()))
    t_n = timeit(lambda: lib.caco_op_gemm_resid_ln(p(a),p(w),p(bias),p(x),M,N,K,p(g),p(b),C.c_float(1e-5),p(h),p(tk),st()))
    print(f"{name}: gemm {t_g:.1f} us  layernorm {t_l:.1f} us  one launch {t_f:.1f} us  one launch without the tail's work (sc1 stores + tickets only) {t_n:.1f} us", flush=True)
