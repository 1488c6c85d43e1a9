#!/bin/bash
# bash tools/isa_regs.sh <source.hip> [pattern] [extra hipcc flags]: registers, LDS and scratch of the kernels of one source
# file (device-only compile to assembly), to see what a change did to the occupancy before it goes to the GPU box.
SRC=$1; PAT=${2:-.}; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $R/build/isa
OUT=$R/build/isa/$(basename $SRC .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I $R/include -I $R/surfelmeshing_amd/csrc "$@" -x hip --cuda-device-only -S $SRC -o $OUT 2> /dev/null || exit 1
python3 - "$OUT" "$PAT" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    if not re.search(sys.argv[2], name):
        continue
    g = lambda k: int(re.search(k + r'\s+(\d+)', body).group(1))
    v = g(r'\.amdhsa_next_free_vgpr'); a = g(r'\.amdhsa_accum_offset') if 'accum_offset' in body else 0
    waves = min(8, 512 // max(8, (v + 7) // 8 * 8))
    print('%-70s vgpr %3d sgpr %3d lds %6d scratch %4d  waves/SIMD %d' % (name[:70], v, g(r'\.amdhsa_next_free_sgpr'),
          g(r'\.amdhsa_group_segment_fixed_size'), g(r'\.amdhsa_private_segment_fixed_size'), waves))
PY
