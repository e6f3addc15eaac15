#!/bin/bash
# Runs on the GPU box: BASELINE configs[2] at the opt-in precisions, fused and unfused
# position-wise kernels; per-kernel times from the side records.
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out
mkdir -p $out
cd $repo
for precision in bf16x3 bf16x3_fast; do
  for fuse in 1 0; do
    EMPHASES_FUSE_QKV=$fuse python3 bench.py --config transformer --precision $precision --steps 50 --warmup 5 \
        --no-cpu-baseline --no-side --side-records $out/split_${precision}_fuse$fuse.json \
        > $out/split_${precision}_fuse$fuse.line 2>> $out/split_try.err
    python3 - $out/split_${precision}_fuse$fuse.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], 'ms_per_step', round(d['ms_per_step'], 4))
print('   ', {k: round(v, 1) for k, v in d['kernels_us_per_step'].items()})
PY
  done
done
tail -3 $out/split_try.err
