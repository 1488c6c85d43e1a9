#!/bin/bash
# bash tools/build_variant.sh <name> [extra hipcc flags...]: compiles the in-tree smx_recon.hip with extra flags (e.g.
# -DSMX_STAMPS) and links build/ab/libsmx_<name>.so with the other in-tree objects (A/B runs: SMX_LIB_PATH).
set -e
N=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/surfelmeshing_amd/csrc
mkdir -p $R/build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I $R/include -I $C "$@" -c $C/smx_recon.hip -o $R/build/ab/recon_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/ab/libsmx_$N.so $C/smx_buffer.o $C/smx_depth.o $R/build/ab/recon_$N.o $C/smx_nn.o $C/smx_synth.o $C/smx_driver.o
