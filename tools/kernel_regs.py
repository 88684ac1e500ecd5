#!/usr/bin/env python3
"""Register / spill / LDS report of every kernel in one translation unit.
   python tools/kernel_regs.py <file.hip> [extra hipcc flags]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=fast",
       "-I", os.path.join(REPO, "include"), *extra, "-Rpass-analysis=kernel-resource-usage", "-c",
       os.path.join(REPO, "cacophony_amd", "csrc", src), "-o", "/tmp/_regs.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r" (TotalSGPRs|VGPRs Spill|SGPRs Spill|VGPRs|AGPRs|ScratchSize|Occupancy|LDS Size)( \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(3))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = name.replace("caco::(anonymous namespace)::", "").replace("(caco::GemmArgs)", "")[:70]
    print(f"{name:70s} v={v.get('VGPRs')} a={v.get('AGPRs')} s={v.get('TotalSGPRs')} vspill={v.get('VGPRs Spill')} "
          f"sspill={v.get('SGPRs Spill')} scratch={v.get('ScratchSize')} occ={v.get('Occupancy')}")
