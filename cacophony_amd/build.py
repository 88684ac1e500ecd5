"""Build libcaco_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m cacophony_amd.build [--force]

The shared library lands next to this file (cacophony_amd/libcaco_hip.so) so that it travels with
the repo snapshot to the GPU box; it is git-ignored.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libcaco_hip.so")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
SOURCES = ["api.hip", "gemm.hip", "gemm_x.hip", "gemm_w8.hip", "gemm_w4q.hip", "gemm_w4h.hip", "attention.hip", "attention_small.hip", "norm.hip", "pool.hip", "mel.hip", "topk.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_epilogue.h", "gemm_w8_epilogue.h", "gemm_w8_common.h", "gemm_w8_ktile.inc", "gemm_w8_skew.inc", "gemm_w8_skew_ktile.inc", "gemm_w8_epilogue_direct.inc", "attention_kpipe.inc", "gemm_w4q_ktile.inc", "gemm_w4h_ktile.inc", os.path.join(INCLUDE, "caco_hip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-ffp-contract=fast"]


# Per-source extra hipcc flags: where a variant build that WON its hardware A/B becomes the default - e.g.
# {"gemm_w8.hip": ["-DW8_F32_SKEW"], "attention.hip": ["-DATTN_LEAN"]} (tools/README.md "Flipping a variant").  The simulator build
# (tools/wavesim/build_sim.py) and tests/test_codegen_budget.py read the same table, so the CPU checks follow a flip.
EXTRA_FLAGS = {}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newest(paths):
    return max(os.path.getmtime(p if os.path.isabs(p) else os.path.join(CSRC, p)) for p in paths)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= _newest([src] + HEADERS):
        return obj
    cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build_library(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[cacophony_amd.build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
