#!/bin/bash
# Runs on the GPU box: BASELINE configs[2] at the opt-in precisions; per-kernel times from the
# side records.
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out
mkdir -p $out
cd $repo
for precision in bf16x3 bf16x3_fast bf16x6; do
    python3 bench.py --config transformer --precision $precision --steps 50 --warmup 5 \
        --no-cpu-baseline --no-side --side-records $out/split_${precision}.json \
        > $out/split_${precision}.line 2>> $out/split_try.err
    python3 - $out/split_${precision}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], 'ms_per_step', round(d['ms_per_step'], 4), round(d['ms_per_step_min'], 4), round(d['ms_per_step_max'], 4))
print('   ', {k: round(v, 1) for k, v in d['kernels_us_per_step'].items()})
PY
done
grep -v amdgpu.ids $out/split_try.err | tail -3
