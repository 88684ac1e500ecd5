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
names=["p1 rest(max B)","p1 vmcnt","p1 barrier","p1 QK_A + R1","p1 QK_B|max + R2","R3 PV_A|expB","R4 PV_B","p2 vmcnt","p2 barrier","p1 issue + epi"]
print("per wave cycles (sums over 8 tiles):")
for k,n in enumerate(names): print(f"{n:16s}", " ".join(f"{int(v):7d}" for v in d[:,k].tolist()))
print("total", " ".join(f"{int(v):7d}" for v in d.sum(1).tolist()))
