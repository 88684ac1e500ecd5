#!/bin/bash
# Build an ablation / A-B variant of the library: tools/build_variant.sh <name> <file.hip[,file2.hip...]> <extra hipcc flags...>
# -> cacophony_amd/_variants/libcaco_hip_<name>.so  (use with CACO_ALLOW_VARIANT_LIB=1 CACO_LIB_PATH=...)
# The listed translation units are recompiled with the product's own flags (cacophony_amd/build.py FLAGS + its per-unit EXTRA_FLAGS:
# after a flip a variant differs from the product by its own macro and nothing else) + the extra flags; every other object is the
# product's own.  A failed compile fails the script: the unit's previous object is removed first and every compile's exit code is read.
# SRC_OVERRIDE=<path> (single file only): compile that file in place of cacophony_amd/csrc/<file.hip> (headers still from csrc/):
# a variant whose source lives outside the product tree is built WITHOUT ever writing into cacophony_amd/csrc/.
set -e
NAME=$1; SRCS=$2; shift 2
cd "$(dirname "$0")/.."
python -m cacophony_amd.build >/dev/null
mkdir -p cacophony_amd/_variants
rm -f cacophony_amd/_variants/libcaco_hip_$NAME.so
OBJS=""; PIDS=""; OTHERS=$(ls cacophony_amd/csrc/_obj/*.o)
for SRC in ${SRCS//,/ }; do
  OBJ=cacophony_amd/_variants/${SRC%.hip}_$NAME.o
  rm -f "$OBJ"
  BASE=$(python -c "import sys; from cacophony_amd import build as b; print(' '.join(b.FLAGS + b.EXTRA_FLAGS.get(sys.argv[1], [])))" "$SRC")
  /opt/rocm/bin/hipcc $BASE -I include -I cacophony_amd/csrc "$@" -c ${SRC_OVERRIDE:-cacophony_amd/csrc/$SRC} -o $OBJ &
  PIDS="$PIDS $!"
  OBJS="$OBJS $OBJ"
  OTHERS=$(echo "$OTHERS" | grep -v "/${SRC%.hip}.o")
done
for p in $PIDS; do wait $p || { echo "compile failed (variant $NAME)"; exit 1; }; done
for o in $OBJS; do [ -s "$o" ] || { echo "compile failed: $o"; exit 1; }; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o cacophony_amd/_variants/libcaco_hip_$NAME.so $OBJS $OTHERS
echo built cacophony_amd/_variants/libcaco_hip_$NAME.so
