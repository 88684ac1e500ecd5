#!/usr/bin/env python3
"""Where does the fc1 GEMM lose 7 % between a back-to-back loop (546 us) and its place in the step (586 us)?
(round-2 verdict item 2a; written in round 3 without a GPU - not yet run).

Times ONE kernel of the audio layer with HIP events around every launch, in chains of growing length on the same
buffers the layer uses (x fp32 [M,768], h bf16 [M,768], a bf16 [M,3072], qkv bf16 [M,2560]):

    fc1 alone | LN -> fc1 | LN -> fc1 -> fc2 | out-proj -> LN -> fc1 -> fc2 | the whole layer (QKV, attention, out-proj, LN, fc1, fc2)

and prints the mean fc1 time per chain, plus rocm-smi socket power / shader clock sampled during each chain.  If fc1 slows
down as soon as LN precedes it, the A operand's cache residency is the cause (LN streams 585 MB through the Infinity Cache
right before); if only the long chains slow it, it is the clock the chip settles at under the whole layer's power draw.

    python tools/gemm_chain_bench.py [--iters 30] [--batch 256]"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import _lib  # noqa: E402


class Sampler(threading.Thread):
    """rocm-smi power / sclk every 50 ms while a chain runs (as tools/power_probe.py does)."""

    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.rows = False, []

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
                self.rows.append(out)
            except Exception:
                pass
            time.sleep(0.05)

    def summary(self):
        import re
        pw, ck = [], []
        for r in self.rows:
            pw += [float(x) for x in re.findall(r"(\d+\.\d+)(?=,|\s*$)", r)[:1]]
            ck += [int(x) for x in re.findall(r"\((\d+)Mhz\)", r)[:1]]
        return (sum(pw) / len(pw) if pw else float("nan")), (sum(ck) / len(ck) if ck else float("nan"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    lib = _lib.load()
    dev = "cuda:0"
    S, H, I, heads = 496, 768, 3072, 8
    M = a.batch * S
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = rnd(M, H).to(dev)
    h = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    act = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
    qkv = torch.empty(M, 2560, dtype=torch.bfloat16, device=dev)
    o = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    mask = torch.ones(a.batch, S, device=dev)
    gam, bet = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    W = {n: (rnd(out_, in_) / in_ ** 0.5).bfloat16().to(dev) for n, (out_, in_) in
         {"qkv": (2304, H), "out": (H, H), "fc1": (I, H), "fc2": (H, I)}.items()}
    B = {n: (rnd(w.shape[0]) * 0.1).to(dev) for n, w in W.items()}
    K = {
        "ln": lambda: lib.caco_op_layernorm(p(x), p(gam), p(bet), M, H, 1e-5, None, p(h), st),
        "qkv": lambda: lib.caco_op_gemm_bf16_strided(p(h), H, p(W["qkv"]), H, p(B["qkv"]), M, 2304, H, 0, p(qkv), 2560, st),
        "attn": lambda: lib.caco_op_attention(p(qkv), 2560, H, 2 * H, p(mask), a.batch, S, heads, H // heads, 0, p(o), st),
        "out": lambda: lib.caco_op_gemm_bf16_f32out(p(o), p(W["out"]), p(B["out"]), p(x), M, H, H, p(x), st),
        "fc1": lambda: lib.caco_op_gemm_bf16(p(h), p(W["fc1"]), p(B["fc1"]), M, I, H, 1, p(act), st),
        "fc2": lambda: lib.caco_op_gemm_bf16_f32out(p(act), p(W["fc2"]), p(B["fc2"]), p(x), M, H, I, p(x), st),
    }
    chains = [["fc1"], ["ln", "fc1"], ["ln", "fc1", "fc2"], ["out", "ln", "fc1", "fc2"], ["ln", "qkv", "attn", "out", "ln", "fc1", "fc2"]]
    for k in ("ln", "qkv", "attn", "out", "ln", "fc1", "fc2"):           # fill every buffer once, keep x bounded
        assert K[k]() == 0, _lib.last_error()
    x.copy_(rnd(M, H).to(dev))
    for chain in chains:
        ev = {k: [] for k in set(chain)}
        for _ in range(5):
            for k in chain:
                K[k]()
        torch.cuda.synchronize()
        smp = Sampler()
        smp.start()
        t0 = time.time()
        for it in range(a.iters):
            if it % 8 == 0:
                x.mul_(0.05)                                            # the residual adds would otherwise grow without bound
            for k in chain:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                K[k]()
                e1.record()
                ev[k].append((e0, e1))
        torch.cuda.synchronize()
        wall = time.time() - t0
        smp.stop = True
        smp.join()
        pw, ck = smp.summary()
        per = {k: sum(e0.elapsed_time(e1) for e0, e1 in v) / len(v) * 1e3 for k, v in ev.items()}
        print(f"{' -> '.join(chain):44s} fc1 {per['fc1']:7.1f} us   " + "  ".join(f"{k} {v:6.1f}" for k, v in sorted(per.items()) if k != "fc1") +
              f"   | {pw:6.0f} W  {ck:5.0f} MHz  ({wall:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
