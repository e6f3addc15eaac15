#!/bin/bash
# builds front-end experiment variants: tools/micro/fx_build.sh name "flags" ...
cd $(dirname $0)/../..
mkdir -p tools/micro/bin
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DNO_STAMPS $flags -Iinclude -Iemphases_amd/csrc \
      tools/micro/frontend_bench.hip -o tools/micro/bin/fe_$name 2>&1 | grep -E "error" 
done
ls tools/micro/bin
