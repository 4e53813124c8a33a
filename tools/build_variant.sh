#!/bin/bash
# Development: a variant library = the shipped objects with ONE unit recompiled under extra flags.
#   tools/build_variant.sh NAME UNIT "FLAGS"     -> gnn-builder_amd/libgnnb_v_NAME.so   (loaded through GNNB_HIP_LIB)
# e.g. tools/build_variant.sh ablate k_stack_zf "-DGNNB_ZF_ABLATE"   (git-ignored like every .so; travels with gpurun)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/gnn-builder_amd/csrc
NAME=$1; UNIT=$2; FLAGS=$3
make -C "$C" >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I"$R/include" -I"$C" -Wall -Wno-unused-function $FLAGS -c -o "$C/build/v_${NAME}_${UNIT}.o" "$C/$UNIT.hip"
OBJS=""
for u in $(sed -n 's/^UNITS := //p' "$C/Makefile"); do
  if [ "$u" = "$UNIT" ]; then OBJS="$OBJS $C/build/v_${NAME}_${UNIT}.o"; else OBJS="$OBJS $C/build/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/gnn-builder_amd/libgnnb_v_$NAME.so" $OBJS
echo "$R/gnn-builder_amd/libgnnb_v_$NAME.so"
