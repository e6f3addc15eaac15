#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [extra hipcc flags]
# Registers / LDS / scratch of every kernel of one source file (device asm metadata).
src=$1; shift
out=/tmp/$(basename $src .hip).s
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S \
    -I$(dirname $src) -I$(dirname $0)/../include "$@" $src -o $out || exit 1
python3 - $out <<'PY'
import re, sys
text = open(sys.argv[1]).read()
for block in text.split('  - .agpr_count:')[1:]:
    get = lambda key: (re.search(r'\.%s:\s*(\S+)' % key, block) or [None, '?'])[1]
    name = get('name')
    print(f"{name[:70]:70s} vgpr {get('vgpr_count'):>4s} agpr {block.split()[0]:>4s} "
          f"sgpr {get('sgpr_count'):>4s} lds {get('group_segment_fixed_size'):>7s} "
          f"scratch {get('private_segment_fixed_size'):>5s} wg {get('max_flat_workgroup_size')}")
PY
