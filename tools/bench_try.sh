#!/bin/bash
# Runs on the GPU box: the driver's own bench command, its line's size, and a
# rocprofv3 kernel trace of the same workload (kernel-exact timer vs the trace).
# usage: tools/bench_try.sh <tag>
tag=${1:-try}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out
mkdir -p $out
cd $repo
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_args.json 2> $out/${tag}_bench.err
echo "rc=$? line bytes: $(wc -c < $out/${tag}_bench_driver_args.json)"
cp bench_side.json $out/${tag}_bench_side.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_1 -- \
    python3 $repo/bench.py --steps 50 --warmup 10 --regions 3 --no-cpu-baseline --no-api --no-side --streams 1 \
    --side-records /tmp/side_prof.json > $out/${tag}_bench_under_rocprof.json 2>> $out/${tag}_bench.err
cp $(find /tmp/prof_1 -name '*kernel_stats.csv' | head -1) $out/${tag}_bench_kernel_stats_1stream.csv
head -12 $out/${tag}_bench_kernel_stats_1stream.csv
tail -5 $out/${tag}_bench.err
cat $out/${tag}_bench_driver_args.json
