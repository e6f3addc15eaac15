#!/bin/bash
# Runs on the GPU box: the N > 1 and strong-scaling bench lines rehearsed on the one GPU
# (two gloo ranks sharing it; one nccl = RCCL rank), each with `roofline` and `cpu_baseline`.
# usage: tools/ranks_round.sh <tag>   -> gpurun_out/<tag>_bench_2ranks_gloo_1gpu.json, <tag>_strong_*.json
tag=${1:-r6}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out
mkdir -p $out
cd $repo
python3 bench.py --gpus 2 --backend gloo > $out/${tag}_bench_2ranks_gloo_1gpu.json 2> $out/${tag}_ranks.err
for workload in corpus longform; do
    python3 bench.py --workload $workload --steps 5 --warmup 2 > $out/${tag}_strong_${workload}_1gpu.json 2>> $out/${tag}_ranks.err
    python3 bench.py --workload $workload --steps 5 --warmup 2 --gpus 2 --backend gloo \
        > $out/${tag}_strong_${workload}_2ranks_gloo_1gpu.json 2>> $out/${tag}_ranks.err
done
python3 - $out $tag <<'PY'
import json, sys
out, tag = sys.argv[1:3]
for name in ('bench_2ranks_gloo_1gpu', 'strong_corpus_1gpu', 'strong_corpus_2ranks_gloo_1gpu',
             'strong_longform_1gpu', 'strong_longform_2ranks_gloo_1gpu'):
    try:
        line = json.load(open(f'{out}/{tag}_{name}.json'))
        print(name, line['n_gpus'], 'ranks:', round(line['value'], 1), line['unit'], '| roofline.frac',
              round(line['roofline']['frac'], 3), '| cpu_baseline', 'cpu_baseline' in line)
    except Exception as error:
        print(name, 'FAILED', repr(error))
PY
tail -5 $out/${tag}_ranks.err
