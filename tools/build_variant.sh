#!/bin/bash
# Build an ablation / A-B variant of the library: tools/build_variant.sh <name> <file.hip> <extra hipcc flags...>
# -> cacophony_amd/_variants/libcaco_hip_<name>.so  (use with CACO_LIB_PATH=...)
# SRC_OVERRIDE=<path>: compile that file in place of cacophony_amd/csrc/<file.hip> (headers still from csrc/): a variant whose source
# lives outside the product tree is built WITHOUT ever writing into cacophony_amd/csrc/.
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/.."
python -m cacophony_amd.build >/dev/null
mkdir -p cacophony_amd/_variants
OBJ=cacophony_amd/_variants/${SRC%.hip}_$NAME.o
EXTRA=""
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast $EXTRA -I include -I cacophony_amd/csrc "$@" -c ${SRC_OVERRIDE:-cacophony_amd/csrc/$SRC} -o $OBJ
OTHERS=$(ls cacophony_amd/csrc/_obj/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o cacophony_amd/_variants/libcaco_hip_$NAME.so $OBJ $OTHERS
echo built cacophony_amd/_variants/libcaco_hip_$NAME.so
