#!/usr/bin/env python3
"""Per-kernel ISA comparison of two gfx950 assembly files (hipcc --cuda-device-only -S): which kernels are identical
instruction for instruction, which changed, which are new.  Kernel arguments' metadata, symbol hashes and debug labels are
ignored; only the instruction stream between `<kernel>:` and its `s_endpgm` / `.Lfunc_end` counts.
    python tools/isa_funcs.py old.s new.s"""
import re
import subprocess
import sys


def kernels(path):
    out, cur, name = {}, None, None
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if line.startswith(".Lfunc_end"):
            name, cur = None, None
            continue
        if cur is not None:
            t = line.split(";")[0].strip()
            if t and not t.startswith("."):
                cur.append(re.sub(r"\.LBB\d+_", ".LBB_", t))      # block labels carry the function's index in the file
    return {k: v for k, v in out.items() if any("s_endpgm" in i for i in v)}


def demangle(n):
    r = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    return r.replace("caco::(anonymous namespace)::", "")[:100]


def base(n):
    """Kernel name without its argument list and without template arguments that were ADDED with a default (a kernel that
    gained a parameter or a defaulted template argument keeps its identity): `void attention_kernel<96, false, 4, 2`."""
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("caco::(anonymous namespace)::", "")
    d = d.split("(")[0]
    return d


def norm_args(ins):
    """The stream with the immediate offsets of scalar loads blanked: a kernel that gained an argument reads its hidden
    arguments (grid size ...) 8 bytes further on and is otherwise the same code."""
    return [re.sub(r"^(s_load_dword\w*\s+\S+\s+s\[\d+:\d+\],)\s*0x[0-9a-f]+", r"\1 <off>", i) for i in ins]


def opcode_mix(ins):
    from collections import Counter
    return Counter(i.split()[0] for i in ins)


old, new = kernels(sys.argv[1]), kernels(sys.argv[2])
# a kernel whose signature changed has a new mangled name: pair it with the old kernel of the same name + leading template
# arguments (longest common prefix of the demangled name before the argument list)
only_old = {k for k in old if k not in new}
only_new = {k for k in new if k not in old}
pairs = {}
for kn in sorted(only_new):
    bn = base(kn)
    cands = [ko for ko in only_old if bn.rstrip(">").startswith(base(ko).rstrip(">")) and ko not in pairs.values()]
    if cands:
        pairs[kn] = max(cands, key=lambda ko: len(base(ko)))
rc = 0
for kn, ko in sorted(pairs.items()):
    if old[ko] == new[kn]:
        print(f"identical  {demangle(kn)}  ({len(new[kn])} instructions; signature changed, instruction stream the same)")
    elif norm_args(old[ko]) == norm_args(new[kn]):
        nd = sum(1 for a, b in zip(old[ko], new[kn]) if a != b)
        print(f"identical* {demangle(kn)}  ({len(new[kn])} instructions; signature changed; the same stream up to {nd} kernel-argument load offsets)")
    else:
        a, b = opcode_mix(old[ko]), opcode_mix(new[kn])
        delta = {op: b[op] - a[op] for op in set(a) | set(b) if a[op] != b[op]}
        print(f"CHANGED    {demangle(kn)}  ({len(old[ko])} -> {len(new[kn])} instructions; signature changed; opcode count changes: {dict(sorted(delta.items()))})")
        rc = 1
    del old[ko], new[kn]
for k in sorted(set(old) | set(new)):
    if k not in old:
        print(f"NEW        {demangle(k)}  ({len(new[k])} instructions)")
    elif k not in new:
        print(f"REMOVED    {demangle(k)}")
        rc = 1
    elif old[k] == new[k]:
        print(f"identical  {demangle(k)}  ({len(new[k])} instructions)")
    else:
        a, b = opcode_mix(old[k]), opcode_mix(new[k])
        delta = {op: b[op] - a[op] for op in set(a) | set(b) if a[op] != b[op]}
        delta = dict(sorted(delta.items())) if len(delta) <= 14 else f"{len(delta)} opcodes, net {sum(delta.values()):+d}"
        print(f"CHANGED    {demangle(k)}  ({len(old[k])} -> {len(new[k])} instructions; opcode count changes: {delta})")
        rc = 1
sys.exit(rc)
