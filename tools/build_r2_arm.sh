#!/bin/bash
# The bisection arm that IS the last binary an MI355X has run: the kernel sources of commit cccbeef (end of round 2; the
# builder's last gpurun session, profiles/r2_v3, ran f08f66f, and the three commits after it left the device code unchanged),
# exported with `git archive` into a temp directory and built there with THAT commit's own build.py ->
#   cacophony_amd/_variants/libcaco_hip_r2.so        (git-ignored; travels with gpurun like the other variants)
# Nothing of it enters the product tree.  The library predates caco_set_switch / caco_get_switch: cacophony_amd/_lib.py binds
# the two switch entry points a variant library lacks to stubs that answer CACO_ERR_INVALID / INT32_MIN ("unknown switch"; only under
# CACO_ALLOW_VARIANT_LIB=1, never for the product library); tools/ab_variants.py leaves them unbound.
# Use: CACO_ALLOW_VARIANT_LIB=1 CACO_LIB_PATH=$PWD/cacophony_amd/_variants/libcaco_hip_r2.so python -m pytest tests/test_gpu_ops.py -m gpu -k "gemm or ragged"
#      python tools/ab_variants.py default r2 classic libm_erf r2addr
set -e
cd "$(dirname "$0")/.."
REV=${1:-cccbeef}
T=$(mktemp -d)
git archive "$REV" cacophony_amd/csrc cacophony_amd/build.py cacophony_amd/__init__.py include | tar x -C "$T"
: > "$T/cacophony_amd/__init__.py"           # build.py only: the archived package is not imported
(cd "$T" && python -c "
import sys; sys.path.insert(0, '.')
from cacophony_amd import build
print(build.build_library(force=True, verbose=False))")
mkdir -p cacophony_amd/_variants
cp "$T/cacophony_amd/libcaco_hip.so" cacophony_amd/_variants/libcaco_hip_r2.so
rm -rf "$T"
python - <<'PY'
import ctypes
lib = ctypes.CDLL("cacophony_amd/_variants/libcaco_hip_r2.so")
lib.caco_version.restype = ctypes.c_char_p
print("built cacophony_amd/_variants/libcaco_hip_r2.so:", lib.caco_version().decode())
PY
