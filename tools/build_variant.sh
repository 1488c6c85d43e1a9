#!/bin/bash
# A variant build of the library for same-box A/Bs:  bash tools/build_variant.sh <name> "<extra flags>" [sources...]
#   -> build/ab/libsmx_<name>.so (the listed sources -- default smx_recon.hip -- recompiled with the flags, the other objects
#   taken from the in-tree build, which must be current: python -m surfelmeshing_amd.build)
set -e
NAME=$1; EXTRA=$2; shift; shift
SRCS=${@:-smx_recon.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/surfelmeshing_amd/csrc; O=$ROOT/build/ab/obj_$NAME
mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I $ROOT/include -I $C"
OBJS=""
for s in smx_buffer.hip smx_depth.hip smx_recon.hip smx_nn.hip smx_synth.hip smx_driver.cpp; do
  b=${s%.*}
  if echo " $SRCS " | grep -q " $s "; then
    X=""; [ "${s##*.}" = cpp ] && X="-x hip"
    /opt/rocm/bin/hipcc $FLAGS $EXTRA $X -c $C/$s -o $O/$b.o &
    OBJS="$OBJS $O/$b.o"
  else
    OBJS="$OBJS $C/$b.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/ab/libsmx_$NAME.so $OBJS
echo build/ab/libsmx_$NAME.so
