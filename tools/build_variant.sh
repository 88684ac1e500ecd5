#!/bin/bash
# Build an ablation / A-B variant of the library: tools/build_variant.sh <name> <file.hip[,file2.hip...]> <extra hipcc flags...>
# -> cacophony_amd/_variants/libcaco_hip_<name>.so  (use with CACO_ALLOW_VARIANT_LIB=1 CACO_LIB_PATH=...)
# The listed translation units are recompiled with the extra flags, every other object is the product's own.
# SRC_OVERRIDE=<path> (single file only): compile that file in place of cacophony_amd/csrc/<file.hip> (headers still from csrc/):
# a variant whose source lives outside the product tree is built WITHOUT ever writing into cacophony_amd/csrc/.
set -e
NAME=$1; SRCS=$2; shift 2
cd "$(dirname "$0")/.."
python -m cacophony_amd.build >/dev/null
mkdir -p cacophony_amd/_variants
OBJS=""; OTHERS=$(ls cacophony_amd/csrc/_obj/*.o)
for SRC in ${SRCS//,/ }; do
  OBJ=cacophony_amd/_variants/${SRC%.hip}_$NAME.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast -I include -I cacophony_amd/csrc "$@" -c ${SRC_OVERRIDE:-cacophony_amd/csrc/$SRC} -o $OBJ &
  OBJS="$OBJS $OBJ"
  OTHERS=$(echo "$OTHERS" | grep -v "/${SRC%.hip}.o")
done
wait
for o in $OBJS; do [ -s "$o" ] || { echo "compile failed: $o"; exit 1; }; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o cacophony_amd/_variants/libcaco_hip_$NAME.so $OBJS $OTHERS
echo built cacophony_amd/_variants/libcaco_hip_$NAME.so
