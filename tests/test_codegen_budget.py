"""Static code-generation budgets of the hot kernels (CPU: hipcc cross-compiles gfx950 without a GPU).

Round 4 found three pieces of compiler-generated overhead in the persistent GEMM by reading its assembly block by block
(profiles/r4_cpu/epilogue_budget.txt): 256 v_mov per wave and output tile for zeroing / copying the accumulators, a bias-only bf16
epilogue of 320 VALU where 128 do the work (v_perm / v_alignbit / v_pk_mov around v_cvt_pk), and a two-branch erff of ~5 500
instructions.  These checks keep them from coming back unnoticed with a compiler or source change, and pin the resources the
design depends on (no spills in the default kernels, 2 waves per SIMD for the 8-wave GEMM).  They say nothing about speed.
"""
import os
import re
import subprocess
from collections import Counter

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "cacophony_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast --cuda-device-only -S".split()

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


def _assembly(src, tmp_path_factory):
    out = os.path.join(str(tmp_path_factory.mktemp("isa")), src.replace(".hip", ".s"))
    r = subprocess.run([HIPCC, *FLAGS, "-I", os.path.join(REPO, "include"), os.path.join(CSRC, src), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


@pytest.fixture(scope="module")
def w8_asm(tmp_path_factory):
    return _assembly("gemm_w8.hip", tmp_path_factory)


@pytest.fixture(scope="module")
def attention_asm(tmp_path_factory):
    return _assembly("attention.hip", tmp_path_factory)


def _kernels(text):
    """name -> (list of basic blocks (lists of instructions), metadata dict)"""
    out = {}
    lines = text.split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if not m:
            continue
        name = m.group(1)
        end = next((j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end")), None)
        if end is None:
            continue
        blocks, cur = [], []
        for s in lines[i + 1:end]:
            s = s.split(";")[0].strip()
            if not s:
                continue
            if re.match(r"\.LBB\d+_\d+:", s):
                blocks.append(cur)
                cur = []
                continue
            if not s.startswith("."):
                cur.append(s)
        blocks.append(cur)
        if any("s_endpgm" in x for b in blocks for x in b):
            out[name] = blocks
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?); Occupancy: (\d+)", text, flags=re.S):
        body = m.group(2)
        g = lambda key: int(re.search(rf"; {key}:? ?=? ?(\d+)", body).group(1))
        meta[m.group(1)] = dict(vgpr=g("NumVgprs"), scratch=g("ScratchSize"), occ=int(m.group(3)))
    return out, meta


def _find(d, *parts):
    hits = [k for k in d if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return hits[0]


def test_persistent_gemm_resources_and_overhead(w8_asm):
    kernels, meta = _kernels(w8_asm)
    w8 = {k: v for k, v in meta.items() if "gemm_bf16_w8_kernel" in k}
    assert len(w8) >= 13
    for name, m in w8.items():
        assert m["scratch"] == 0, f"{name} spills {m['scratch']} bytes"
        assert m["vgpr"] <= 256 and m["occ"] == 2, f"{name}: {m}"          # two waves per SIMD is what the K-loop is scheduled for
    # QKV (bf16 + bias), fc1 (+ SiLU), text fc1 (+ erf-GELU): template arguments <EPI 0, ACT, MODE 1>
    qkv, fc1, gelu = (_find(kernels, f"gemm_bf16_w8_kernelILi0ELi{a}ELi1E") for a in (0, 1, 2))
    for name in (qkv, fc1, gelu):
        blocks = kernels[name]
        ops = Counter(i.split()[0] for b in blocks for i in b)
        assert ops["v_perm_b32"] == 0 and ops["v_alignbit_b32"] == 0, f"{name}: the bf16 conversion is scalarised again: {ops['v_perm_b32']} v_perm"
        # no block that is (almost) nothing but register moves: the accumulators are never zeroed or copied by VALU
        for b in blocks:
            movs = sum(1 for i in b if i.startswith(("v_mov_b32", "v_mov_b64", "v_pk_mov_b32")))
            assert not (len(b) >= 64 and movs >= 0.8 * len(b)), f"{name}: a block of {len(b)} instructions with {movs} register moves"
        mf = [i for b in blocks for i in b if "v_mfma" in i]
        assert len(mf) == 128 and sum(1 for i in mf if i.rstrip().endswith(", 0")) == 32, f"{name}: peeled first K-tile expected (128 MFMAs, 32 with C = 0)"
    store = lambda name: max((b for b in kernels[name] if any(i.startswith("buffer_store") for i in b)), key=len)
    n_qkv, n_fc1, n_gelu = (len(store(k)) for k in (qkv, fc1, gelu))
    assert n_qkv <= 420, f"bias-only epilogue block: {n_qkv} instructions (368 when written on register pairs, 570 scalarised)"
    assert n_fc1 <= 900, f"SiLU epilogue block: {n_fc1} instructions (832 in round 4)"
    assert n_gelu <= 2600, f"erf-GELU epilogue block: {n_gelu} instructions (2 209 with the rational erf, ~5 500 with the library's erff)"
    assert Counter(i.split()[0] for i in store(fc1))["v_exp_f32_e32"] == 128          # one exp + one rcp per element, nothing more
    # the fp32 + residual kernel: 32 loads + 32 stores of 16 bytes per lane and tile in its epilogue, no spills (checked above)
    f32 = _find(kernels, "gemm_bf16_w8_kernelILi1ELi0ELi2E")
    ops = Counter(i.split()[0] for i in store(f32))
    assert ops["buffer_store_dwordx4"] == 32 and ops["buffer_load_dwordx4"] >= 32


def test_attention_resources(attention_asm):
    kernels, meta = _kernels(attention_asm)
    audio = _find(meta, "attention_kernelILi96ELb0ELi4ELi2ELb0E")
    assert meta[audio]["vgpr"] <= 256 and meta[audio]["occ"] == 2 and meta[audio]["scratch"] <= 8, meta[audio]
    text = _find(meta, "attention_kernelILi64ELb1ELi4ELi1ELb0E")
    assert meta[text]["scratch"] == 0 and meta[text]["occ"] >= 3, meta[text]
    # per 64-key tile and wave of the two-block kernel: 24 score MFMAs + 24 P.V MFMAs, 64 exponentials
    loop_mfma = sum(1 for b in kernels[audio] for i in b if "v_mfma_f32_32x32x16_bf16" in i)
    assert loop_mfma == 48, loop_mfma


def test_no_kernel_of_the_library_spills_beyond_the_known_few(tmp_path_factory):
    """Every kernel of every translation unit: scratch (= spilled registers) is 0, except the two known cases - the two-block audio
    attention kernel (8 bytes, outside its loop) and the experimental four-wave GEMM gemm_w4q (<= 128 bytes, outside its K-loop; never a default kernel)."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f not in ("gemm_w8.hip", "attention.hip", "api.hip"))
    with ThreadPoolExecutor(4) as ex:
        texts = list(ex.map(lambda s: _assembly(s, tmp_path_factory), srcs))
    bad = []
    for src, text in zip(srcs, texts):
        _, meta = _kernels(text)
        assert meta or src in ("api.hip",), f"{src}: no kernels found"
        for name, m in meta.items():
            limit = 128 if "gemm_bf16_w4q_kernel" in name else 0
            if m["scratch"] > limit:
                bad.append((src, name[:80], m["scratch"]))
    assert not bad, bad
