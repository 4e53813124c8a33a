#!/bin/bash
# Development: a variant library = the shipped objects with SEVERAL units recompiled under extra flags.
#   tools/build_variant2.sh NAME "FLAGS" UNIT [UNIT ...]   -> gnn-builder_amd/libgnnb_v_NAME.so   (loaded through GNNB_HIP_LIB)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/gnn-builder_amd/csrc
NAME=$1; FLAGS=$2; shift 2
make -C "$C" >/dev/null
for UNIT in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I"$R/include" -I"$C" -Wall -Wno-unused-function $FLAGS -c -o "$C/build/v_${NAME}_${UNIT}.o" "$C/$UNIT.hip" &
done
wait
OBJS=""
for u in $(sed -n 's/^UNITS := //p' "$C/Makefile"); do
  if [[ " $* " == *" $u "* ]]; then OBJS="$OBJS $C/build/v_${NAME}_${u}.o"; else OBJS="$OBJS $C/build/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/gnn-builder_amd/libgnnb_v_$NAME.so" $OBJS
echo "$R/gnn-builder_amd/libgnnb_v_$NAME.so"
