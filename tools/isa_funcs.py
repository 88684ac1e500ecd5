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


old, new = kernels(sys.argv[1]), kernels(sys.argv[2])
rc = 0
for k in sorted(set(old) | set(new)):
    if k not in old:
        print(f"NEW        {demangle(k)}  ({len(new[k])} instructions)")
    elif k not in new:
        print(f"REMOVED    {demangle(k)}")
        rc = 1
    elif old[k] == new[k]:
        print(f"identical  {demangle(k)}  ({len(new[k])} instructions)")
    else:
        nd = sum(1 for a, b in zip(old[k], new[k]) if a != b) + abs(len(old[k]) - len(new[k]))
        print(f"CHANGED    {demangle(k)}  ({len(old[k])} -> {len(new[k])} instructions, {nd} differ)")
        rc = 1
sys.exit(rc)
