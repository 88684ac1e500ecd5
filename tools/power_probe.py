#!/usr/bin/env python3
"""Is a kernel power-limited?  Runs one GEMM shape back to back for a few seconds on random / zero operands (and optionally a
forced kernel) while sampling rocm-smi: average socket power, sclk.  Same instruction stream, different operand bits.
   python tools/power_probe.py [--only fc1] [--tile 256] [--seconds 3]"""
import argparse
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import _lib  # noqa: E402

SHAPES = {"qkv": (126976, 2304, 768, "bf16", 0), "fc1": (126976, 3072, 768, "bf16", 1), "fc2": (126976, 768, 3072, "f32r", 0)}


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\):\s*([0-9.]+)", r)
            c = re.search(r"sclk clock level:.*\((\d+)Mhz\)", r)
            out.append((float(p.group(1)) if p else None, int(c.group(1)) if c else None))
        except Exception as e:
            out.append((None, None))
        time.sleep(0.15)


def measure(name, run, seconds, flops=None, chunk=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples))
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(chunk):
            run()
        torch.cuda.synchronize()
        n += chunk
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    pw = [s[0] for s in samples[1:] if s[0]]
    ck = [s[1] for s in samples[1:] if s[1]]
    tf = f"{flops * n / dt / 1e12:7.1f} TFLOP/s  " if flops else ""
    print(f"{name}: {dt / n * 1e6:9.1f} us  {tf}power avg {sum(pw) / max(1, len(pw)):.0f} W (max {max(pw, default=0):.0f})  "
          f"sclk avg {sum(ck) / max(1, len(ck)):.0f} MHz (min {min(ck, default=0)})  [{len(pw)} samples]", flush=True)


def stages(seconds):
    """The whole step and its non-GEMM kernels: which of them sit at the power cap?"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from cacophony_amd import config as Cfg, synth
    from cacophony_amd.model import create_caco_model
    lib = _lib.load()
    dev = torch.device("cuda:0")
    state = synth.make_caco_state(Cfg.default_audio_config(), Cfg.default_text_config(), Cfg.default_caco_config())
    model = create_caco_model(device=dev).load_state_dict(state)
    wav, ids, mask = bench._make_inputs(256, 0, dev)
    measure("step (encode_pairs, both towers)", lambda: model.encode_pairs(wav, ids, mask, 500), seconds, chunk=3)
    measure("audio tower only", lambda: model.encode_audio(wav, 500), seconds, chunk=3)
    measure("text tower only", lambda: model.encode_text(ids, mask, check_ids=False), seconds, chunk=10)
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    B, S, heads, hd = 256, 496, 8, 96
    H = heads * hd
    qkv = torch.randn(B * S, 2560, device=dev).bfloat16()
    km = torch.ones(B, S, device=dev)
    out = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev)
    measure("attention S=496", lambda: lib.caco_op_attention(p(qkv), 2560, H, 2 * H, p(km), B, S, heads, hd, 0, p(out), st), seconds,
            flops=2 * 2 * S * S * H * B)
    x = torch.randn(B * S, H, device=dev)
    g = torch.ones(H, device=dev)
    ob = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev)
    measure("layernorm", lambda: lib.caco_op_layernorm(p(x), p(g), p(g), B * S, H, 1e-5, None, p(ob), st), seconds)


def attn(seconds):
    """The audio attention launch on random and on zero operands (same instruction stream)."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    B, S, heads, hd = 256, 496, 8, 96
    H = heads * hd
    km = torch.ones(B, S, device=dev)
    out = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev)
    for data in ("randn", "zeros"):
        qkv = torch.randn(B * S, 2560, device=dev).bfloat16()
        if data == "zeros":
            qkv.zero_()
        measure(f"attention S=496 {data}", lambda: lib.caco_op_attention(p(qkv), 2560, H, 2 * H, p(km), B, S, heads, hd, 0, p(out), st),
                seconds, flops=2 * 2 * S * S * H * B)


def blaslt(seconds, only):
    """Calibration only (never on the product path): the vendor library's bf16 GEMM (torch.matmul -> hipBLASLt) on the same
    shapes and operand bits, no epilogue at all.  Does it get past the power cap where the w8 kernel does not?"""
    dev = "cuda:0"
    shapes = dict(SHAPES)
    shapes["sq8k"] = (8192, 8192, 8192, "bf16", 0)
    for name in only.split(","):
        M, N, K, _, _ = shapes[name]
        for data in ("randn", "zeros"):
            A = torch.randn(M, K, device=dev).bfloat16()
            W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
            if data == "zeros":
                A.zero_(); W.zero_()
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            Wt = W.t()
            measure(f"hipBLASLt {name} {M}x{N}x{K} {data}", lambda: torch.matmul(A, Wt, out=out), seconds, flops=2.0 * M * N * K, chunk=50)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blaslt", action="store_true")
    ap.add_argument("--stages", action="store_true")
    ap.add_argument("--attn", action="store_true")
    ap.add_argument("--only", default="fc1")
    ap.add_argument("--tile", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=3.0)
    a = ap.parse_args()
    if a.blaslt:
        return blaslt(a.seconds, a.only)
    if a.stages:
        return stages(a.seconds)
    if a.attn:
        return attn(a.seconds)
    lib = _lib.load()
    lib.caco_set_gemm_tile(a.tile)
    dev = "cuda:0"
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name in a.only.split(","):
        M, N, K, kind, act = SHAPES[name]
        for data in ("randn", "zeros"):
            A = torch.randn(M, K, device=dev).bfloat16()
            W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
            if data == "zeros":
                A.zero_(); W.zero_()
            bias = torch.randn(N, device=dev)
            if kind == "bf16":
                out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                run = lambda: lib.caco_op_gemm_bf16(p(A), p(W), p(bias), M, N, K, act, p(out), st)
            else:
                out = torch.randn(M, N, device=dev)
                run = lambda: lib.caco_op_gemm_bf16_f32out(p(A), p(W), p(bias), p(out), M, N, K, p(out), st)
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            stop, samples = threading.Event(), []
            th = threading.Thread(target=sample, args=(stop, samples))
            th.start()
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < a.seconds:
                for _ in range(50):
                    run()
                torch.cuda.synchronize()
                n += 50
            dt = time.perf_counter() - t0
            stop.set(); th.join()
            pw = [s[0] for s in samples[1:] if s[0]]
            ck = [s[1] for s in samples[1:] if s[1]]
            tf = 2.0 * M * N * K * n / dt / 1e12
            print(f"{name} tile {a.tile} {data:5s}: {dt / n * 1e6:7.1f} us  {tf:7.1f} TFLOP/s  power avg {sum(pw) / max(1, len(pw)):.0f} W (max {max(pw, default=0):.0f})  "
                  f"sclk avg {sum(ck) / max(1, len(ck)):.0f} MHz (min {min(ck, default=0)})  [{len(pw)} samples]", flush=True)


if __name__ == "__main__":
    main()
