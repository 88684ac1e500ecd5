#!/bin/bash
# Make a variant that WON its hardware A/B the default (tools/README.md "Flipping a variant"):
#   bash tools/flip_variant.sh gemm_w8.hip -DW8_F32_SKEW [attention.hip -DATTN_LEAN ...]      add flags to a unit
#   bash tools/flip_variant.sh --unflip gemm_w8.hip -DW8_F32_SKEW                               take one back
#   bash tools/flip_variant.sh --show
# Edits EXTRA_FLAGS in cacophony_amd/build.py (the one table the product build, the simulator build, tests/test_codegen_budget.py and
# tools/build_variant.sh read), rebuilds cacophony_amd/libcaco_hip.so, and runs the CPU checks that must follow a flip: the ISA budgets
# (incl. the data-flow proof of skew's hand-counted waits on THIS toolchain), the ABI, and the kernel sources on the simulator - the
# comparison arms of the variant tests become explicit -U builds by themselves.  Rehearsed in round 6 with skew + attn_lean on a copy
# of the tree (profiles/r6_cpu/flip_rehearsal.txt).  It does NOT touch hardware: the next step is `gpu_session.sh truth` on the new default.
set -e
cd "$(dirname "$0")/.."
if [ "${1:-}" = --show ]; then python -c "from cacophony_amd.build import EXTRA_FLAGS; print('EXTRA_FLAGS =', EXTRA_FLAGS)"; exit 0; fi
python - "$@" <<'PY'
import ast, re, sys
args = sys.argv[1:]
path = "cacophony_amd/build.py"
text = open(path).read()
m = re.search(r"^EXTRA_FLAGS = (\{.*\})$", text, re.M)
table = ast.literal_eval(m.group(1))
undo = bool(args) and args[0] == "--unflip"
args = args[1:] if undo else args
if not args or len(args) % 2:
    sys.exit("usage: flip_variant.sh [--unflip] <unit.hip> <-Dflag> [<unit.hip> <-Dflag> ...] | --show")
import os
for unit, flag in zip(args[::2], args[1::2]):
    if not os.path.exists(os.path.join("cacophony_amd/csrc", unit)) or not flag.startswith("-D"):
        sys.exit(f"bad pair: {unit} {flag}")
    cur = table.setdefault(unit, [])
    if undo:
        if flag in cur: cur.remove(flag)
        if not cur: del table[unit]
    elif flag not in cur:
        cur.append(flag)
open(path, "w").write(text[:m.start(1)] + repr(table).replace("'", '"') + text[m.end(1):])
print("EXTRA_FLAGS =", table)
PY
python -m cacophony_amd.build --force
python -m pytest tests/test_codegen_budget.py tests/test_abi.py tests/test_wavesim.py -q -x
echo "flip done on the CPU side.  Next: gpurun -- 'bash tools/gpu_session.sh truth' on the new default; drop the variant from tools/build_variants.sh;"
echo "regenerate profiles/<round>/isa_vs_*.txt (tools/isa_diff.sh) and commit the A/B record (ab_variants.txt, predictions_vs_measured.txt) next to the flip."
