#!/bin/bash
# ISA identity check of the shipped kernels against a git ref (default HEAD): compiles every csrc/*.hip of both trees
# for gfx950 (device only, -S) with the library's flags and diffs the assembly, ignoring the per-build __hip_cuid symbol.
#   tools/isa_diff.sh [ref] [file.hip ...]
# A translation unit that differs is compared kernel by kernel (tools/isa_funcs.py: identical / CHANGED / NEW).
# Exit status 0 = no existing kernel's instruction stream changed (new kernels are allowed).  Used while the GPU pool is closed: the default build must
# stay the build that was last verified on hardware; new kernels go in behind switches.
set -u
REF=${1:-HEAD}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
FILES=("$@")
[ ${#FILES[@]} -eq 0 ] && FILES=(api gemm gemm_x gemm_w8 attention norm pool mel topk)
OLD=$(mktemp -d) ; NEW=$(mktemp -d)
git -C "$REPO" archive "$REF" cacophony_amd/csrc include | tar -x -C "$OLD"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast --cuda-device-only -S"
rc=0
for f in "${FILES[@]}"; do
  f=${f%.hip}
  ( /opt/rocm/bin/hipcc $FLAGS -I "$OLD/include" "$OLD/cacophony_amd/csrc/$f.hip" -o "$OLD/$f.s" 2>/dev/null ) &
  ( /opt/rocm/bin/hipcc $FLAGS -I "$REPO/include" "$REPO/cacophony_amd/csrc/$f.hip" -o "$NEW/$f.s" 2>/dev/null ) &
done
wait
for f in "${FILES[@]}"; do
  f=${f%.hip}
  if [ ! -s "$OLD/$f.s" ]; then echo "$f: not in $REF (new file)"; continue; fi
  if diff -q <(grep -v "__hip_cuid" "$OLD/$f.s") <(grep -v "__hip_cuid" "$NEW/$f.s") >/dev/null; then echo "$f: identical"
  else echo "$f: differs - per kernel:"; python3 "$REPO/tools/isa_funcs.py" "$OLD/$f.s" "$NEW/$f.s" | sed 's/^/    /'; [ ${PIPESTATUS[0]} -ne 0 ] && rc=1; fi
done
rm -rf "$OLD" "$NEW"
exit $rc
