#!/bin/bash
# bash tools/build_variant_nn.sh <name> <nn-source.hip> [extra flags]: build/ab/libsmx_<name>.so with another smx_nn source
set -e
N=$1; SRC=$2; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/surfelmeshing_amd/csrc
mkdir -p $R/build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I $R/include -I $C "$@" -x hip -c $SRC -o $R/build/ab/nn_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/ab/libsmx_$N.so $C/smx_buffer.o $C/smx_depth.o $C/smx_recon.o $R/build/ab/nn_$N.o $C/smx_synth.o $C/smx_driver.o
