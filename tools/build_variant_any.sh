#!/bin/bash
# bash tools/build_variant_any.sh <name> <depth-source.hip|-> <recon-source.hip|-> [extra hipcc flags]: links
# build/ab/libsmx_<name>.so from the given smx_depth / smx_recon sources ("-" = the in-tree object) and the other in-tree
# objects (A/B runs on one GPU box: SMX_LIB_PATH, tools/ab_libs.sh).
set -e
N=$1; D=$2; RC=$3; shift; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/surfelmeshing_amd/csrc
mkdir -p $R/build/ab
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I $R/include -I $C"
DO=$C/smx_depth.o; RO=$C/smx_recon.o
if [ "$D" != "-" ]; then /opt/rocm/bin/hipcc $FL "$@" -x hip -c $D -o $R/build/ab/depth_$N.o; DO=$R/build/ab/depth_$N.o; fi
if [ "$RC" != "-" ]; then /opt/rocm/bin/hipcc $FL "$@" -x hip -c $RC -o $R/build/ab/recon_$N.o; RO=$R/build/ab/recon_$N.o; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/ab/libsmx_$N.so $C/smx_buffer.o $DO $RO $C/smx_nn.o $C/smx_synth.o $C/smx_driver.o
