#!/bin/bash
# Build a variant of the library with extra -D flags for ONE source (or several: a comma-separated list):
#   tools/build_variant.sh <out.so> <file.hip>[,<file2.hip>...] [-DX=1 ...]
# (objects of the other sources are taken from the last regular build).  Used for same-box A/B runs (tools/ab_step.py) and for the
# debug-knob builds of the timing tools (-DMMSA_DEBUG_KNOBS: csrc/common.h MMSA_KNOB; mmsa.lib loads the variant named by MMSA_LIB).
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
P=$ROOT/multimodal-sam-adapter_amd
OUT=$1; SRCS=$2; shift 2
mkdir -p $(dirname $ROOT/$OUT)
OBJS=$(ls $P/build/*.o)
NEW=""
for SRC in ${SRCS//,/ }; do
  B=$(basename $SRC .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -fno-slp-vectorize -w "$@" -c $P/csrc/$SRC -o /tmp/variant_${B}_$$.o
  OBJS=$(echo "$OBJS" | grep -v "/$B.o")
  NEW="$NEW /tmp/variant_${B}_$$.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/$OUT $OBJS $NEW
echo built $OUT
