import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from cacophony_amd import _lib
lib = _lib.load()
B,S,heads,hd = 256,500,8,96
H=heads*hd
qkv = torch.randn(B,S,3*H,device="cuda").bfloat16()
mask = torch.ones(B,S,device="cuda"); mask[:,S-4:]=0
out = torch.zeros(B*S*H + 8192,dtype=torch.bfloat16,device="cuda")
p=lambda t: C.c_void_p(t.data_ptr())
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    lib.caco_op_attention(p(qkv),3*H,H,2*H,p(mask),B,S,heads,hd,0,p(out),st)
torch.cuda.synchronize()
d = out[B*S*H:B*S*H+8*16*4].view(torch.int64).cpu().view(8,16)[:, :10]
names=["-","-","-","R1 QK_A (+issue)","R2 QK_B|smA","R3 PV_A|smB (+rescale A)","R4 PV_B (+rescale B)","vmcnt","barrier","prologue+epilogue"]
print("per wave cycles (sums over 8 tiles):")
for k,n in enumerate(names): print(f"{n:16s}", " ".join(f"{int(v):7d}" for v in d[:,k].tolist()))
print("total", " ".join(f"{int(v):7d}" for v in d.sum(1).tolist()))
