#!/usr/bin/env python3
"""ThreadSanitizer run of the kernels on the wavesim build (build_sim.py --tsan): one TSan fiber per lane, fiber switches
carry no synchronisation, and the only happens-before edges inside a launch are the hardware's - a workgroup barrier, a
wave-level operation / CACO_WAVE_LDS_SYNC() within its wave.  Reported: two lanes of different waves touching the same LDS
bytes with no barrier between them, two workgroups touching the same global bytes in one launch (workgroups are dealt
round-robin over WAVESIM_THREADS workers; those of one worker are ordered, so the run is repeated with 2 and 3 workers).

    python tools/wavesim/tsan_check.py            # self-test (a planted LDS race and a planted global race must be reported,
                                                  # their barrier-ed twins must not), then the driver of asan_check.py
    python tools/wavesim/tsan_check.py [--skew] --pytest [pytest arguments]
                                                  # tests/test_wavesim.py (every kernel and the model paths: decoders, MAE,
                                                  # poolers, ...) with the TSan build loaded through CACO_SIM_LIB

Exit status 0 and "TSAN CLEAN" on the last line = self-test behaved and the library produced no report."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

SELFTEST = r'''
import ctypes as C, numpy as np, sys
lib = C.CDLL(LIB)
P = lambda a: C.c_void_p(a.ctypes.data)
a = np.arange(64, dtype=np.float32); o = np.zeros(64, np.float32); z = np.zeros(128, np.float32)
if WHICH == "clean":
    lib.selftest_handoff(P(a), P(o), 1); assert (o == a[::-1] * 2).all()
    lib.selftest_overlap(P(z), 64)
elif WHICH == "lds":
    lib.selftest_handoff(P(a), P(o), 0)
else:
    lib.selftest_overlap(P(z), 32)
print("DRIVER DONE")
'''


# second driver: the persistent GEMM kernels with several tiles per workgroup (WAVESIM_CUS = 3: LDS stages are reused from tile to
# tile), the audio tower's attention shape (two query blocks per wave) with a masked tail
DRIVER2 = r'''
import ctypes as C, math, os, sys
import numpy as np
sys.path.insert(0, REPO)
from cacophony_amd import _lib, config as Cfg, synth
lib = C.CDLL(LIB)
for name, (res, args) in _lib._SIGNATURES.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
P = lambda a: C.c_void_p(0 if a is None else a.ctypes.data)
def chk(rc, what=""):
    assert rc == 0, (what, lib.caco_last_error())
def bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)
rng = np.random.default_rng(1)
# persistent kernels with several tiles per workgroup (WAVESIM_CUS = 3): LDS stages are reused from tile to tile
for tile in (2256, 8256, 4256, 4128):
    lib.caco_set_gemm_tile(tile)
    for (M, N, K, act) in ((2100, 768, 512, 1), (1300, 1536, 256, 0)):
        a, w, b = bf16(rng.standard_normal((M, K))), bf16(rng.standard_normal((N, K)) / math.sqrt(K)), rng.standard_normal(N).astype(np.float32)
        out = np.zeros((M, N), np.uint16)
        chk(lib.caco_op_gemm_bf16(P(a), P(w), P(b), M, N, K, act, P(out), None), "gemm")
        x = rng.standard_normal((M, N)).astype(np.float32)
        chk(lib.caco_op_gemm_bf16_f32out(P(a), P(w), P(b), P(x), M, N, K, P(x), None), "gemm f32")
lib.caco_set_gemm_tile(256)
# the audio tower's attention shape (two query blocks per wave), and a masked tail
for (B, S, heads, hd, causal) in ((2, 500, 2, 96, 0), (1, 300, 2, 64, 0)):
    H = heads * hd
    qkv = bf16(rng.standard_normal((B, S, 3 * H)))
    mask = np.ones((B, S), np.float32); mask[-1, S - 7:] = 0
    out = np.zeros((B, S, H), np.uint16)
    chk(lib.caco_op_attention(P(qkv), 3 * H, H, 2 * H, P(mask), B, S, heads, hd, causal, P(out), None), "attention")
print("DRIVER DONE")
'''


def sites(err):
    """Unique (kernel source line, kernel source line) pairs of the reports."""
    out = set()
    for rep in err.split("WARNING: ThreadSanitizer: data race")[1:]:
        frames = re.findall(r"#0 \S+.*? (/\S+?:\d+)(?::\d+)? \(", rep)
        out.add(tuple(sorted(os.path.relpath(f, REPO) for f in frames[:2])))
    return sorted(out)


def run(code, lib, rt, threads, **extra):
    env = dict(os.environ, **extra, LD_PRELOAD=rt, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4", WAVESIM_THREADS=str(threads),
               OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\nLIB = {lib!r}\n" + code], env=env, capture_output=True, text=True)
    return r, sites(r.stderr)


def main():
    sys.path.insert(0, HERE)
    import asan_check
    import build_sim
    lib = build_sim.build(tsan=True, lib=os.path.join(HERE, "libcaco_sim_tsan.so"), verbose=False,
                          extra=[os.path.join(HERE, "selftest", "race_selftest.hip")])
    rt = subprocess.run([build_sim.CXX, "-print-file-name=libclang_rt.tsan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if "--pytest" in sys.argv:
        args = sys.argv[sys.argv.index("--pytest") + 1:] or ["tests/test_wavesim.py", "-q"]
        env = dict(os.environ, LD_PRELOAD=rt, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4", CACO_SIM_LIB=lib,
                   WAVESIM_THREADS="3", OMP_NUM_THREADS="1")
        if "--skew" in sys.argv[:sys.argv.index("--pytest")]:          # the `skew` variant's own cases on a TSan build of that variant
            env["CACO_SIM_SKEW_LIB"] = build_sim.build(tsan=True, defines=("-DW8_F32_SKEW",), tag="skew", verbose=False)
        r = subprocess.run([sys.executable, "-m", "pytest", *args], cwd=REPO, env=env, capture_output=True, text=True)
        sys.stdout.write(r.stdout[-1500:])
        s = sites(r.stderr)
        print(f"pytest under ThreadSanitizer: exit {r.returncode}, {len(s)} distinct race sites")
        for pair in s:
            print("   ", "  <->  ".join(pair))
        print("TSAN CLEAN" if (r.returncode == 0 and not s) else "TSAN REPORTS OR FAILURE")
        return 0 if (r.returncode == 0 and not s) else 1
    ok = True
    for which, expect in (("clean", 0), ("lds", 1), ("global", 1)):
        r, s = run(f"WHICH = {which!r}\n" + SELFTEST, lib, rt, 2)
        good = "DRIVER DONE" in r.stdout and (len(s) > 0) == bool(expect)
        print(f"self-test {which:<7} reports: {len(s)}  {'ok' if good else 'UNEXPECTED'}")
        ok = ok and good
    for threads in (2, 3):
        r, s = run(asan_check.DRIVER, lib, rt, threads)
        done = "DRIVER DONE" in r.stdout
        print(f"library driver, {threads} workers: {'finished' if done else 'FAILED (exit %d)' % r.returncode}, {len(s)} distinct race sites")
        for pair in s:
            print("   ", "  <->  ".join(pair))
        if not done:
            sys.stdout.write(r.stderr[-3000:])
        ok = ok and done and not s
    r, s = run(DRIVER2, lib, rt, 3, WAVESIM_CUS="3")
    done = "DRIVER DONE" in r.stdout
    print(f"multi-tile driver (3 CUs, 3 workers): {'finished' if done else 'FAILED (exit %d)' % r.returncode}, {len(s)} distinct race sites")
    for pair in s:
        print("   ", "  <->  ".join(pair))
    ok = ok and done and not s
    print("TSAN CLEAN" if ok else "TSAN REPORTS OR FAILURE")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
