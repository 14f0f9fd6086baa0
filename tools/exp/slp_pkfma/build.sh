#!/bin/bash
# builds ./repro (gfx950) and writes the packed-FMA lines of both builds' ISA to isa_slp_on.txt / isa_slp_off.txt
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 --offload-arch=gfx950 -c slp_on.hip -o slp_on.o
$HIPCC -O3 --offload-arch=gfx950 -fno-slp-vectorize -c slp_off.hip -o slp_off.o
$HIPCC -O3 --offload-arch=gfx950 -c repro.hip -o repro.o
$HIPCC --offload-arch=gfx950 slp_on.o slp_off.o repro.o -o repro
$HIPCC -O3 --offload-arch=gfx950 -S --cuda-device-only slp_on.hip -o slp_on.s
$HIPCC -O3 --offload-arch=gfx950 -fno-slp-vectorize -S --cuda-device-only slp_off.hip -o slp_off.s
grep -n "v_pk_fma_f32\|v_pk_mul_f32\|v_pk_add_f32" slp_on.s > isa_slp_on.txt || true
grep -n "v_pk_fma_f32\|v_pk_mul_f32\|v_pk_add_f32" slp_off.s > isa_slp_off.txt || true
echo "packed fp32 instructions: SLP build $(wc -l < isa_slp_on.txt), scalar build $(wc -l < isa_slp_off.txt)"
rm -f slp_on.o slp_off.o repro.o slp_on.s slp_off.s
