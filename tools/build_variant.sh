#!/bin/bash
# Build a variant of the library with extra -D flags for ONE source:  tools/build_variant.sh <out.so> <file.hip> [-DX=1 ...]
# (objects of the other sources are taken from the last regular build).  Used for same-box A/B runs (tools/ab_step.py).
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
P=$ROOT/multimodal-sam-adapter_amd
OUT=$1; SRC=$2; shift 2
mkdir -p $(dirname $ROOT/$OUT)
B=$(basename $SRC .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -fno-slp-vectorize "$@" -c $P/csrc/$SRC -o /tmp/variant_${B}_$$.o
OBJS=$(ls $P/build/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/$OUT $OBJS /tmp/variant_${B}_$$.o
echo built $OUT
