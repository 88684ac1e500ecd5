import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cacophony_amd import _lib
lib = _lib.load(); dev = "cuda:0"
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
for M in (4096, 70000):
    N = K = 768
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev); g = torch.randn(N, device=dev); b = torch.randn(N, device=dev); x0 = torch.randn(M, N, device=dev)
    x1 = x0.clone(); h1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    lib.caco_op_gemm_resid_ln(p(a), p(w), p(bias), p(x1), M, N, K, p(g), p(b), 1e-5, p(h1), st)
    for tile in (256, 128):
        lib.caco_set_gemm_tile(tile)
        x2 = x0.clone(); h2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        lib.caco_op_gemm_bf16_f32out(p(a), p(w), p(bias), p(x2), M, N, K, p(x2), st)
        lib.caco_op_layernorm(p(x2), p(g), p(b), M, N, 1e-5, None, p(h2), st)
        torch.cuda.synchronize()
        print(M, tile, "x equal", torch.equal(x1, x2), (x1 - x2).abs().max().item(), "h equal", torch.equal(h1, h2), (h1.float() - h2.float()).abs().max().item())
    lib.caco_set_gemm_tile(256)
