#!/bin/bash
# A/B of library variants inside the step on ONE box: tools/ab_bench.sh [reps] default variant1 variant2 ...
# (variants = names under cacophony_amd/_variants/libcaco_hip_<name>.so from tools/build_variant.sh)
export CACO_ALLOW_VARIANT_LIB=1      # cacophony_amd/_lib.py refuses CACO_LIB_PATH without it
REPS=$1; shift
for i in $(seq $REPS); do for v in "$@"; do
  if [ $v = default ]; then unset CACO_LIB_PATH; else export CACO_LIB_PATH=$PWD/cacophony_amd/_variants/libcaco_hip_$v.so; fi
  python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']
print('$v', d['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in s.items() if k.startswith('audio.g') or k in ('audio.ln','audio.attention')})"
done; done
