#!/bin/bash
# Energy per launch at a FIXED shader clock (rocm-smi performance-determinism mode, below the throttle point), so that the
# socket power of two kernels doing the same work in the same time compares directly.   bash tools/energy_ab.sh [MHz] [shapes]
export CACO_ALLOW_VARIANT_LIB=1      # cacophony_amd/_lib.py refuses CACO_LIB_PATH without it
MHZ=${1:-1500}; SHAPES=${2:-fc1}
rocm-smi --setperfdeterminism $MHZ 2>&1 | grep -v "^$" | tail -3
for v in "" _variants/libcaco_hip_mf16.so _variants/libcaco_hip_mf16_noepi.so _variants/libcaco_hip_mf16_nodma.so _variants/libcaco_hip_mf16_noreads.so; do
  [ -n "$v" ] && [ ! -f cacophony_amd/$v ] && continue
  echo "== lib ${v:-default}"
  if [ -n "$v" ]; then export CACO_LIB_PATH=$PWD/cacophony_amd/$v; else unset CACO_LIB_PATH; fi
  python tools/power_probe.py --only $SHAPES --seconds 2 2>&1 | grep -v amdgpu.ids
done
unset CACO_LIB_PATH
echo "== hipBLASLt"
python tools/power_probe.py --blaslt --only $SHAPES --seconds 2 2>&1 | grep -v amdgpu.ids
echo "== register-resident MFMA loops"
python tools/mfma_power.py --seconds 2 --variants 0,1 2>&1 | grep -v amdgpu.ids
rocm-smi --resetperfdeterminism 2>&1 | tail -2
