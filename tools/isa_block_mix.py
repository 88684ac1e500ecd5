#!/usr/bin/env python3
"""Opcode histogram of every basic block of one kernel in a gfx950 assembly file (hipcc --cuda-device-only -S):
    python tools/isa_block_mix.py file.s <kernel-name-regex> [min_block_size]
A CPU-side way to spot compiler-generated overhead (permutes, moves, re-materialised addresses) in issue-bound code."""
import re
import sys
from collections import Counter

path, pat = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and re.search(pat, l)]
for start in starts:
    end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
    print("==", lines[start][:150])
    name, cur, blks = "entry", [], []
    for l in lines[start + 1:end]:
        s = l.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m:
            blks.append((name, cur, ""))
            name, cur = m.group(1), []
            if "Loop" in l:
                name += " [" + l.split(";")[1].strip()[:50] + "]"
            continue
        if s.startswith("."):
            continue
        cur.append(s)
    blks.append((name, cur, ""))
    for name, b, _ in blks:
        if len(b) >= minsz:
            c = Counter(i.split()[0] for i in b)
            print(f"{name}: {len(b)} instr: " + ", ".join(f"{k} {v}" for k, v in c.most_common(18)))
