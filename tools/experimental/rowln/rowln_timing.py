import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cacophony_amd import _lib
lib = _lib.load(); dev = "cuda:0"
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 126976, 768, 768
a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
bias = torch.randn(N, device=dev); g = torch.ones(N, device=dev); b = torch.zeros(N, device=dev); x = torch.randn(M, N, device=dev)
nwg = M // 64
hbuf = torch.zeros(M * N * 2 + nwg * 32, dtype=torch.uint8, device=dev)
for _ in range(3):
    lib.caco_op_gemm_resid_ln(p(a), p(w), p(bias), p(x), M, N, K, p(g), p(b), 1e-5, p(hbuf), st)
torch.cuda.synchronize()
s = hbuf[M * N * 2:].view(torch.int64).cpu().numpy().reshape(nwg, 4).astype(np.float64)
kl = s[:, 1] - s[:, 0]; ep = s[:, 2] - s[:, 1]
print("K-loop cycles mean %.0f (min %.0f max %.0f); epilogue mean %.0f (min %.0f max %.0f)" % (kl.mean(), kl.min(), kl.max(), ep.mean(), ep.min(), ep.max()))
t0 = s[:, 0].min(); print("launch span %.0f cycles; first 8 WG starts %s" % (s[:, 2].max() - t0, (np.sort(s[:, 0])[:8] - t0)))
print("per-slab %.0f cycles" % (kl.mean() / 24))
