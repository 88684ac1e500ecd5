#!/usr/bin/env python3
"""Static instruction mix of the loops of every kernel in one translation unit (hipcc --cuda-device-only -S for gfx950):
per loop (a backward branch to an earlier block label) the number of MFMA, other VALU (transcendentals apart), LDS, vector
memory, scalar and wait instructions in the loop body as laid out.  These kernels are bound by the instructions a wave
issues, so the count per tile is the CPU-side proxy for a kernel change while no GPU is at hand.
    python tools/isa_loop_stats.py <file.hip> [kernel-substring] [-D...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
filt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
extra = [a for a in sys.argv[2:] if a.startswith("-")]
path = src if os.path.exists(src) else os.path.join(REPO, "cacophony_amd", "csrc", src)
asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=fast",
                      "--cuda-device-only", "-S", "-I", os.path.join(REPO, "include"), "-I", os.path.join(REPO, "cacophony_amd", "csrc"),
                      *extra, path, "-o", "-"], capture_output=True, text=True).stdout


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if re.match(r"v_(exp|rcp|sqrt|log|rsq|sin|cos)", op):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op in ("s_waitcnt", "s_barrier", "s_nop", "s_sleep"):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


name, blocks, order = None, {}, []
kernels = {}
for line in asm.splitlines():
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, blocks, order = m.group(1), {"entry": []}, ["entry"]
        kernels[name] = (blocks, order)
        continue
    if name is None:
        continue
    if line.startswith(".Lfunc_end"):
        name = None
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
        blocks[m.group(1)] = []
        order.append(m.group(1))
        continue
    t = line.split(";")[0].strip()
    if t and not t.startswith("."):
        blocks[order[-1]].append(t)

for k, (blocks, order) in kernels.items():
    dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("caco::(anonymous namespace)::", "")
    if filt not in dem:
        continue
    pos = {b: i for i, b in enumerate(order)}
    loops = []
    for b in order:
        for ins in blocks[b]:
            m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", ins)
            if m:
                tgt = m.group(1) or m.group(2)
                if tgt in pos and pos[tgt] <= pos[b]:
                    loops.append((pos[tgt], pos[b]))
    print(dem[:110])
    for lo, hi in sorted(set(loops)):
        cnt = {}
        for b in order[lo:hi + 1]:
            for ins in blocks[b]:
                c = klass(ins.split()[0])
                cnt[c] = cnt.get(c, 0) + 1
        if cnt.get("mfma", 0) == 0 and sum(cnt.values()) < 40:
            continue
        tot = sum(cnt.values())
        print(f"   loop {order[lo]:>10s}..{order[hi]:<10s} {tot:5d} instr: " + "  ".join(f"{c} {cnt.get(c, 0)}" for c in ("mfma", "valu", "trans", "lds", "vmem", "salu", "smem", "wait", "other")))
