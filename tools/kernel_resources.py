#!/usr/bin/env python3
"""Static resources of every kernel of the library, from hipcc's own `; Kernel info:` blocks (device-only -S build with the
library's flags): code size, VGPRs / AGPRs, SGPRs, scratch (= spills), static LDS, occupancy (waves per SIMD).
    python tools/kernel_resources.py [file.hip ...] > profiles/<round>/kernel_resources.txt
No GPU needed.  Dynamic LDS is a launch argument and not in the table (DESIGN.md lists it per kernel)."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "cacophony_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast --cuda-device-only -S".split()


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def one(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I", os.path.join(REPO, "include"), path, "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{path}:\n{r.stderr[-2000:]}")
        text = open(out).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?); Occupancy: (\d+)", text, flags=re.S):
        body = m.group(2)
        g = lambda key: int(re.search(rf"; {key}:? ?=? ?(\d+)", body).group(1))
        rows.append(dict(name=m.group(1), code=g("codeLenInByte"), vgpr=g("NumVgprs"), agpr=g("NumAgprs"), sgpr=g("TotalNumSgprs"),
                         scratch=g("ScratchSize"), lds=g("LDSByteSize"), occ=int(m.group(3))))
    return rows


def short(name):
    if name.startswith("_Z"):                    # c++filt does not know the bf16 mangling (DF16b): name + integer template arguments by hand
        m = re.search(r"\d+([a-z0-9_]*kernel)(I(?:L[ib]\d+E)+E)?", name)
        if m:
            targs = re.findall(r"L[ib](\d+)E", m.group(2) or "")
            return m.group(1) + (f"<{', '.join(targs)}>" if targs else "")
    name = name.replace("caco::(anonymous namespace)::", "").replace("caco::", "")
    name = re.sub(r"\(.*\)$", "", name)          # drop the argument list
    return re.sub(r"^void ", "", name)


def main():
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    paths = [f if os.path.isabs(f) or os.path.exists(f) else os.path.join(CSRC, f) for f in files]
    with ThreadPoolExecutor(4) as ex:
        results = list(ex.map(one, paths))
    print(f"{'kernel':<78} {'code B':>7} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>7} {'LDS B':>7} {'waves/SIMD':>10}")
    for path, rows in zip(paths, results):
        if not rows:
            continue
        print(f"-- {os.path.basename(path)}")
        names = demangle([r["name"] for r in rows])
        for r, n in zip(rows, names):
            print(f"{short(n)[:78]:<78} {r['code']:>7} {r['vgpr']:>5} {r['agpr']:>5} {r['sgpr']:>5} {r['scratch']:>7} {r['lds']:>7} {r['occ']:>10}")
    spilled = [short(n) for rows in results for r, n in zip(rows, demangle([x["name"] for x in rows])) if r["scratch"]]
    print(f"\nkernels with scratch: {len(spilled)}" + ("".join("\n  " + s for s in spilled) if spilled else ""))


if __name__ == "__main__":
    main()
