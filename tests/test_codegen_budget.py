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
if REPO not in __import__("sys").path:
    __import__("sys").path.insert(0, REPO)
CSRC = os.path.join(REPO, "cacophony_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast --cuda-device-only -S".split()

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


def _assembly(src, tmp_path_factory, defines=()):
    out = os.path.join(str(tmp_path_factory.mktemp("isa")), src.replace(".hip", ".s"))
    from cacophony_amd.build import EXTRA_FLAGS          # the product build's per-source flags (a flipped variant)
    r = subprocess.run([HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), *defines, "-I", os.path.join(REPO, "include"), os.path.join(CSRC, src), "-o", out], capture_output=True, text=True)
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


def _vregs(operand):
    """'v[12:15]' / 'v7' -> set of VGPR numbers (anything else: empty)."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", operand)
    return {int(m.group(1))} if m else set()


def test_skewed_row_block_variant_budget(tmp_path_factory):
    """Variant build `skew` (-DW8_F32_SKEW, csrc/gemm_w8_skew.inc: the fp32 + residual epilogue under the K-loop; never part of the
    product library).  Resources the design depends on, the counted waits, and the one thing inline-assembly loads put at risk:
    the residual registers are written by a load the compiler does not track and become valid only at the kernel's own
    s_waitcnt vmcnt(12) of the NEXT event - nothing may read, copy or spill them in between."""
    text = _assembly("gemm_w8.hip", tmp_path_factory, ["-DW8_F32_SKEW"])
    kernels, meta = _kernels(text)
    # two instantiations: <false> the plain fp32 + residual form, <true> the LayerNorm-fold producer (bf16 copy + row statistics)
    for fold, tag in ((False, "gemm_bf16_w8s_kernelILb0E"), (True, "gemm_bf16_w8s_kernelILb1E")):
        _skew_kernel_checks(text, kernels, meta, _find(meta, tag), fold)


def _skew_kernel_checks(text, kernels, meta, name, fold):
    assert meta[name]["scratch"] == 0 and meta[name]["vgpr"] <= 256 and meta[name]["occ"] == 2, meta[name]
    # the other kernels of the translation unit are the default build's (the variant only adds one)
    assert all(m["scratch"] == 0 for k, m in meta.items() if "gemm_bf16_w8_kernel" in k)
    blocks = kernels[name]
    flat = [i for b in blocks for i in b]
    ops = Counter(i.split()[0] for i in flat)
    assert ops["v_mfma_f32_16x16x32_bf16"] % 64 == 0 and ops["v_mfma_f32_16x16x32_bf16"] <= 1024, ops["v_mfma_f32_16x16x32_bf16"]
    assert not any(o.startswith("scratch_") for o in ops)
    # nine event sites (block 0's sits once ahead of the period loop and once at its bottom): 4 residual loads (inline assembly:
    # "offen offset:<n> nt") + 4 direct fp32 stores each; the final burst of the circular form: 28 tracked loads (system-scope
    # policy, not "nt") + 28 stores
    rd_loads = [i for i in flat if i.startswith("buffer_load_dwordx4") and " nt" in i and "lds" not in i]
    stores = [i for i in flat if i.startswith("buffer_store_dwordx4")]
    burst_loads = [i for i in flat if i.startswith("buffer_load_dwordx4") and "sc0 sc1" in i and "lds" not in i]
    assert len(rd_loads) == 36 and len(stores) == 36 + 28 and len(burst_loads) == 28, (len(rd_loads), len(stores), len(burst_loads))
    # fold producer: 4 bf16 row pieces + 1 statistics pair per completed block (9 event sites + 7 blocks of the burst), never masked
    assert sum(1 for i in flat if i.startswith("buffer_store_dwordx2")) == (5 * (9 + 7) if fold else 0)
    # waits: vmcnt(0) only before the loop and at the very end; the events' and the K-tile barriers' counted waits are 12 / 4
    waits = Counter(re.search(r"vmcnt\((\d+)\)", i).group(1) for i in flat if i.startswith("s_waitcnt") and "vmcnt" in i)
    assert waits["17" if fold else "12"] >= 17 and waits["4"] >= 8, waits
    loop_blocks = [b for b in blocks if any("v_mfma" in i for i in b)]
    assert not any("vmcnt(0)" in i for b in loop_blocks for i in b), "a full drain inside the K-loop"
    # the residual registers: the same 16 at every event site ...
    rd_regs = set()
    for i in rd_loads:
        rd_regs |= _vregs(i.split()[1].rstrip(","))
    assert len(rd_regs) == 16, sorted(rd_regs)
    # ... and on EVERY path between a residual load and the event wait that makes it valid (the plain `s_waitcnt vmcnt(12)`; the
    # K-tile barrier's wait carries lgkmcnt(0) as well and leaves the event's loads in flight) no instruction may name one of the
    # registers in flight.  Forward data flow over the kernel's control-flow graph (labels, branches, fall-through).
    def names(instr):
        regs = set()
        for tok in re.findall(r"v\[\d+:\d+\]|v\d+", instr):
            regs |= _vregs(tok)
        return regs

    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(j for j in range(start, len(lines)) if lines[j].startswith(".Lfunc_end"))
    labels, body, cur = ["entry"], {"entry": []}, "entry"
    for l in lines[start + 1:end]:
        l = l.split(";")[0].strip()
        if not l:
            continue
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            labels.append(cur)
            body[cur] = []
        elif not l.startswith("."):
            body[cur].append(l)
    succ = {}
    for n, lab in enumerate(labels):
        out = [i.split()[-1] for i in body[lab] if i.startswith(("s_cbranch", "s_branch"))]
        last = body[lab][-1] if body[lab] else ""
        if not last.startswith(("s_branch", "s_endpgm")) and n + 1 < len(labels):
            out.append(labels[n + 1])
        succ[lab] = out

    def transfer(lab, state, check):
        state = set(state)
        for i in body[lab]:
            if i.startswith("s_waitcnt") and ("vmcnt(0)" in i or ((f"vmcnt({17 if fold else 12})") in i and "lgkmcnt" not in i)):
                state = set()
                continue
            if i.startswith("buffer_load_dwordx4") and " nt" in i and "lds" not in i:
                if check:
                    assert not (names(" ".join(i.split()[2:])) & state), f"{lab}: address of a residual load in a register still in flight: {i}"
                state |= _vregs(i.split()[1].rstrip(","))
                continue
            if check:
                hit = names(i) & state
                assert not hit, f"{lab}: `{i}` names residual registers {sorted(hit)} before the wait that makes them valid"
        return state

    inn = {lab: set() for lab in labels}
    changed = True
    while changed:
        changed = False
        for lab in labels:
            out = transfer(lab, inn[lab], False)
            for t in succ[lab]:
                if not out <= inn[t]:
                    inn[t] |= out
                    changed = True
    assert any(inn[lab] for lab in labels), "the data flow found no residual register in flight anywhere: the check checks nothing"
    for lab in labels:
        transfer(lab, inn[lab], True)
    # hipcc pads nothing INSIDE an inline-assembly string and its hazard recogniser does not see the string as a memory instruction
    # (cdna_hip_programming.md, inline assembly: "an "s" operand fresh from readfirstlane -> a buffer_* inside reading it as
    # descriptor, soffset or base" needs its wait states by hand): a vector instruction writing one of the descriptor's scalar
    # registers must lie at least 5 wait states ahead of every residual load, on the straight-line code before it (labels crossed:
    # the conservative reading).  Today the descriptor is built from scalar loads / s_mov long before the first event.
    flat_lines = [l.split(";")[0].strip() for l in lines[start + 1:end]]
    flat_lines = [l for l in flat_lines if l and not l.startswith(".") or re.match(r"\.LBB\d+_\d+:", l or "")]
    n_checked = 0
    for at, l in enumerate(flat_lines):
        if not (l.startswith("buffer_load_dwordx4") and " nt" in l and "lds" not in l):
            continue
        m = re.search(r"s\[(\d+):(\d+)\]", l)
        desc = set(range(int(m.group(1)), int(m.group(2)) + 1))
        states, j = 0, at - 1
        while j >= 0 and states < 5:
            t = flat_lines[j]
            j -= 1
            if t.endswith(":"):
                continue
            op = t.split()[0]
            if op == "s_nop":
                states += int(t.split()[1]) + 1
                continue
            if op.startswith(("v_readfirstlane", "v_readlane")) or (op.startswith("v_cmp") and t.split()[1].startswith("s")):
                dst = t.split()[1].rstrip(",")
                mm = re.fullmatch(r"s\[(\d+):(\d+)\]", dst)
                written = set(range(int(mm.group(1)), int(mm.group(2)) + 1)) if mm else ({int(dst[1:])} if re.fullmatch(r"s\d+", dst) else set())
                assert not (written & desc), f"`{t}` writes a descriptor register {states} wait states ahead of `{l}` (needs 5)"
            states += 1
        n_checked += 1
    assert n_checked == 36


def test_attention_resources(attention_asm):
    kernels, meta = _kernels(attention_asm)
    audio = _find(meta, "attention_kernelILi96ELb0ELi4ELi2ELb0E")
    assert meta[audio]["vgpr"] <= 256 and meta[audio]["occ"] == 2 and meta[audio]["scratch"] <= 8, meta[audio]
    text = _find(meta, "attention_kernelILi64ELb1ELi4ELi1ELb0E")
    assert meta[text]["scratch"] == 0 and meta[text]["occ"] >= 3, meta[text]
    # per 64-key tile and wave of the two-block kernel: 24 score MFMAs + 24 P.V MFMAs, 64 exponentials
    loop_mfma = sum(1 for b in kernels[audio] for i in b if "v_mfma_f32_32x32x16_bf16" in i)
    assert loop_mfma == 48, loop_mfma


def test_attention_lean_variant_budget(tmp_path_factory):
    """Variant build `attn_lean` (-DATTN_LEAN; the default attention kernels stay the round-2 instruction streams): no attention
    kernel spills, the two-block kernel's output epilogue carries no v_perm / v_alignbit around its bf16 conversions and is at
    least 120 instructions shorter per wave, and its tile loop is unchanged in matrix work."""
    base_k, base_m = _kernels(_assembly("attention.hip", tmp_path_factory, ["-UATTN_LEAN"]))      # explicit: stays the other arm after a flip
    lean_k, lean_m = _kernels(_assembly("attention.hip", tmp_path_factory, ["-DATTN_LEAN"]))
    assert lean_m and all(m["scratch"] == 0 for m in lean_m.values()), {k[-40:]: m for k, m in lean_m.items() if m["scratch"]}
    audio = _find(lean_m, "attention_kernelILi96ELb0ELi4ELi2ELb0E")
    assert lean_m[audio]["vgpr"] <= 256 and lean_m[audio]["occ"] == 2
    n = lambda ks, name: sum(len(b) for b in ks[name])
    ops = Counter(i.split()[0] for b in lean_k[audio] for i in b)
    assert ops["v_perm_b32"] == 0 and ops["v_alignbit_b32"] == 0, ops
    assert n(base_k, audio) - n(lean_k, audio) >= 120, (n(base_k, audio), n(lean_k, audio))
    assert sum(1 for b in lean_k[audio] for i in b if "v_mfma_f32_32x32x16_bf16" in i) == 48


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
