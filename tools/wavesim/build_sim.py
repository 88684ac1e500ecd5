"""Build tools/wavesim/libcaco_sim.so: the kernel sources of cacophony_amd/csrc compiled for x86 against the wavesim
functional model (wavesim.h) instead of hipcc / gfx950.  Same C ABI as libcaco_hip.so (include/caco_hip.h), "device"
pointers are host pointers.  TEST INFRASTRUCTURE: used by tests/test_wavesim*.py (CPU) only.

    python tools/wavesim/build_sim.py [--force] [--asan] [--extra file.hip ...]

Source translation (the only edits made to a kernel source; everything else is the file as hipcc sees it):
  * `asm volatile("s_waitcnt ..." ::: "memory")`           -> wavesim::s_waitcnt("...")
  * `asm volatile("s_waitcnt vmcnt(%0) ..." :: "n"(E) ...)` -> wavesim::s_waitcnt_n("...", E)
  * `extern __shared__ ... T NAME[];`                       -> T* NAME = (T*)wavesim::dyn_lds();
  * `__attribute__((amdgpu_...(...)))` on kernels           -> dropped (occupancy hints)
  * any other inline assembly must sit in the #else branch of an `#ifdef WAVESIM` and carry the marker `/* hw-only */`
Textually included kernel pieces (`csrc/*.inc`, e.g. the K-tile body of gemm_w8) get the same translation; the translated copy is
written next to the generated source, where the quote-include finds it first.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "cacophony_amd", "csrc")
INCLUDE = os.path.join(REPO, "include")
GEN = os.path.join(HERE, "_gen")
SHIM = os.path.join(HERE, "shim")
LIB = os.path.join(HERE, "libcaco_sim.so")
SOURCES = ["api.hip", "gemm.hip", "gemm_x.hip", "gemm_w8.hip", "gemm_w4q.hip", "gemm_w4h.hip", "attention.hip", "attention_small.hip", "norm.hip", "pool.hip", "mel.hip", "topk.hip"]
CXX = os.environ.get("WAVESIM_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-O2", "-g1", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-ffp-contract=off",
         "-Wno-unknown-attributes", "-Wno-unused-value", "-Wno-ignored-attributes", "-Wno-c++20-extensions",
         "-Wno-macro-redefined", "-Wno-pass-failed", "-x", "c++"]

_ASM = re.compile(r'asm\s+volatile\s*\(\s*"(s_waitcnt[^"]*)"\s*(.*?)\)\s*;')
_EXT = re.compile(r'extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(char|float|int)\s+(\w+)\s*\[\s*\]\s*;')


def translate(text: str) -> str:
    def asm(m):
        body, ops = m.group(1), m.group(2)
        n = re.search(r'"n"\s*\((.*)\)\s*:\s*"memory"', ops)
        if "%0" in body:
            if not n:
                raise ValueError(f"cannot translate asm operands: {m.group(0)}")
            return f'wavesim::s_waitcnt_n("{body}", {n.group(1)});'
        return f'wavesim::s_waitcnt("{body}");'
    text = _ASM.sub(asm, text)
    text = re.sub(r"__attribute__\(\(amdgpu_[a-z_]+\([^)]*\)\)\)", "", text)      # kernel-only attributes
    text = _EXT.sub(lambda m: f"{m.group(1)}* {m.group(2)} = reinterpret_cast<{m.group(1)}*>(wavesim::dyn_lds());", text)
    # a line marked /* hw-only */ sits in the #else branch of an #ifdef WAVESIM (the simulator compiles the other branch)
    left = [l for l in text.splitlines() if re.search(r"\basm\s+volatile", l) and '""' not in l and "/* hw-only */" not in l]
    if left:
        raise ValueError("untranslated inline asm:\n" + "\n".join(left))
    return text


def _newer(out, deps):
    return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps)


def build(force: bool = False, asan: bool = False, extra=(), lib: str = LIB, verbose: bool = True, replace=None, defines=(),
          tag: str = "", tsan: bool = False, ubsan: bool = False) -> str:
    """replace = {"attention.hip": "/path/to/variant.hip"}: a kernel source swapped for an experimental one (tools/experimental),
    defines = ("-DATTN_FAST_PASS", ...): extra compiler flags; both need a `tag` (object files and library are kept apart)."""
    replace = dict(replace or {})
    if (replace or defines) and not tag:
        raise ValueError("a variant build needs a tag")
    if tag and lib == LIB:          # a sanitizer build of a variant gets its own library too (it used to overwrite the plain variant's)
        lib = os.path.join(HERE, f"libcaco_sim_{tag}{'_asan' if asan else ''}{'_tsan' if tsan else ''}{'_ubsan' if ubsan else ''}.so")
    os.makedirs(GEN, exist_ok=True)
    os.makedirs(os.path.join(SHIM, "hip"), exist_ok=True)
    shim = os.path.join(SHIM, "hip", "hip_runtime.h")
    if not os.path.exists(shim):
        with open(shim, "w") as f:
            f.write('// stands in for <hip/hip_runtime.h> in the wavesim build\n#pragma once\n#include "wavesim.h"\n')
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [
        os.path.join(HERE, "wavesim.h"), os.path.join(INCLUDE, "caco_hip.h"), os.path.abspath(__file__)]
    flags = FLAGS + list(defines) + (["-fsanitize=address", "-fno-omit-frame-pointer"] if asan else [])
    if ubsan:          # undefined behaviour in the kernel sources' integer / pointer arithmetic (signed overflow, shifts, misaligned or null access ...)
        flags = flags + ["-fsanitize=undefined,float-cast-overflow,float-divide-by-zero", "-fno-sanitize=vptr,function", "-fno-omit-frame-pointer"]
    if tsan:                                    # kernels instrumented, the runtime (wavesim.cpp) only annotated: see wavesim.cpp
        flags = flags + ["-DWAVESIM_TSAN"]
    tag = ("." + tag if tag else "") + (".asan" if asan else "") + (".tsan" if tsan else "") + (".ubsan" if ubsan else "")

    # textually included kernel pieces (csrc/*.inc) get the same translation; the copy in _gen/ is found first by the
    # quote-include of the generated source next to it
    for inc in sorted(f for f in os.listdir(CSRC) if f.endswith(".inc")):
        with open(os.path.join(CSRC, inc)) as f:
            t = translate(f.read())
        dst = os.path.join(GEN, inc)
        if not os.path.exists(dst) or open(dst).read() != t:
            with open(dst, "w") as f:
                f.write(t)
    headers = headers + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".inc")]

    sys.path.insert(0, REPO)
    from cacophony_amd.build import EXTRA_FLAGS as product_extra          # -D flags a flipped variant added to the product build
    headers = headers + [os.path.join(REPO, "cacophony_amd", "build.py")]

    # A variant's -D macros usually touch one translation unit: a unit whose source (and the csrc headers / .inc pieces, which any
    # unit may include) never names one of them compiles to the same object as in the untagged build, so that object is shared.
    # (-U<macro> counts as well: the "base" arm of a flipped variant undoes the product build's own -D, which comes first on the command line)
    macros = [re.sub(r"^-[DU]", "", d).split("=")[0] for d in defines if d.startswith(("-D", "-U"))]
    san_tag = (".asan" if asan else "") + (".tsan" if tsan else "") + (".ubsan" if ubsan else "")

    def closure_text(path, seen=None):
        """The unit's source plus every csrc file it includes with quotes, transitively (conditional includes count too)."""
        seen = set() if seen is None else seen
        if path in seen or not os.path.exists(path):
            return ""
        seen.add(path)
        text = open(path).read()
        for inc in re.findall(r'#include\s+"([^"]+)"', text):
            text += closure_text(os.path.join(CSRC, inc), seen)
        return text

    def untouched(path, base):
        if not macros or base in replace or not base.endswith(".hip"):
            return False
        text = closure_text(path)
        return not any(re.search(rf"\b{re.escape(m)}\b", text) for m in macros)

    def one(src):
        path = src if os.path.isabs(src) else os.path.join(CSRC, src)
        base = os.path.basename(path)
        if base in replace:
            path = replace[base]
        obj_tag = san_tag if untouched(path, base) else tag
        obj = os.path.join(GEN, base.replace(".hip", obj_tag + ".o").replace(".cpp", obj_tag + ".o"))
        if not force and _newer(obj, [path] + headers):
            return obj
        gen = os.path.join(GEN, base.replace(".hip", obj_tag + ".cpp"))
        if base.endswith(".hip"):
            with open(path) as f:
                t = translate(f.read())
            with open(gen, "w") as f:
                f.write(f'#line 1 "{path}"\n' + t)
        else:
            gen = path
        san = ["-fsanitize=thread"] if (tsan and base.endswith(".hip")) else []
        use_defines = list(defines) if obj_tag == tag else []
        use_flags = [f for f in flags if f not in defines]
        cmd = [CXX, *use_flags, *[f for f in product_extra.get(base, []) if f.startswith("-D")], *use_defines, *san, "-I", SHIM, "-I", HERE, "-I", CSRC, "-I", INCLUDE, "-c", gen, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{base}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            sys.stderr.write(r.stderr)
        return obj

    srcs = list(SOURCES) + list(extra) + [os.path.join(HERE, "wavesim.cpp")]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, srcs))
    if force or not _newer(lib, objs):
        tmp = f"{lib}.{os.getpid()}.tmp"           # link beside it and rename: a process that has the old library mapped keeps it
        cmd = [CXX, "-shared", "-fPIC", "-o", tmp, *objs, "-lpthread"] + (["-fsanitize=address"] if asan else []) + (["-fsanitize=thread"] if tsan else []) + (["-fsanitize=undefined,float-cast-overflow,float-divide-by-zero"] if ubsan else [])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, lib)
        if verbose:
            print(f"[wavesim] linked {lib}")
    return lib


if __name__ == "__main__":
    extra = []
    if "--extra" in sys.argv:
        extra = sys.argv[sys.argv.index("--extra") + 1:]
    tsan = "--tsan" in sys.argv
    ubsan = "--ubsan" in sys.argv
    build(force="--force" in sys.argv, asan="--asan" in sys.argv, extra=extra, tsan=tsan, ubsan=ubsan,
          lib=os.path.join(HERE, "libcaco_sim_tsan.so") if tsan else (os.path.join(HERE, "libcaco_sim_ubsan.so") if ubsan else LIB))
