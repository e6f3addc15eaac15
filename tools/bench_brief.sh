#!/bin/bash
# usage: tools/bench_brief.sh [bench args]  -> one short line
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-side --no-api "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('ms/step %.3f  conv %.1f us (%.1f TF, %.0f%%)  ' % (d['ms_per_step'], r['avg_launch_us'], r['achieved'], 100*r['frac']), {k: round(v,1) for k,v in d['kernels_us_per_step'].items()})"
