#!/bin/bash
# Runs on the GPU box (ONE box, so the pairs compare): BASELINE configs[2] at bf16x3 with round
# 5's position-wise kernels (tiles of 32, two launches per layer) and with round 6's (tiles of
# 16, one launch); three alternations.  And attention_split_kernel's in-kernel timeline.
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out/r6_ab.txt
cd $repo
: > $out
for lap in 1 2 3; do
  for mode in "32 0" "16 1"; do
    set -- $mode
    EMPHASES_SPLIT_TILE=$1 EMPHASES_FUSE_QKV=$2 python3 bench.py --config transformer --precision bf16x3 --steps 50 --warmup 5 \
        --no-cpu-baseline --no-side --side-records /tmp/ab_side.json > /tmp/ab.json 2>/dev/null
    python3 - $1 $2 >> $out <<'PY'
import json, sys
d = json.load(open('/tmp/ab.json')); s = json.load(open('/tmp/ab_side.json'))
k = s['kernels_us_per_step']
pw = sum(v for n, v in k.items() if 'split' in n)
print(f"tile {sys.argv[1]} fuse {sys.argv[2]}: {d['ms_per_step']:.4f} ms per step ({d['ms_per_step_min']:.4f}-{d['ms_per_step_max']:.4f}); "
      f"position-wise {pw:.1f} us, attention {k['attention_frames']:.1f} us per step (eager, kernel-exact)")
PY
  done
done
cat $out
$repo/tools/micro/bin/attention_split_bench_32 > $repo/gpurun_out/r6_attention_stamps.txt 2>&1
cat $repo/gpurun_out/r6_attention_stamps.txt
