#!/bin/bash
# Runs on the GPU box (through gpurun): the round's bench line, the rocprofv3
# kernel statistics of the same workload and the two HBM PMC passes.
# usage: tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*
tag=${1:-r6}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out
mkdir -p $out
cd $repo
# (the line on stdout; the full record - every side measurement - in the side file)
python3 bench.py --side-records $out/${tag}_bench_side.json > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --side-records $out/${tag}_bench_driver_args_side.json \
    > $out/${tag}_bench_driver_args.json 2>> $out/${tag}_bench.err
python3 bench.py --streams 1 --no-cpu-baseline --no-api --no-side --side-records $out/${tag}_bench_1stream_side.json \
    > $out/${tag}_bench_1stream.json 2>> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
for streams in 1 2; do
    rm -rf /tmp/prof_$streams
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$streams -- \
        python3 $repo/bench.py --steps 50 --warmup 10 --regions 3 --no-cpu-baseline --no-api --no-side --streams $streams --side-records /tmp/side_prof.json \
        > /dev/null 2>> $out/${tag}_bench.err
    cp $(find /tmp/prof_$streams -name '*kernel_stats.csv' | head -1) \
        $out/${tag}_bench_kernel_stats_${streams}stream.csv
done
for counter in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$counter
    rocprofv3 --pmc $counter --output-format csv -d /tmp/pmc_$counter -- \
        python3 $repo/bench.py --steps 10 --warmup 2 --no-preroll --regions 1 --no-cpu-baseline --no-api --no-side --streams 1 --side-records /tmp/side_prof.json \
        > /dev/null 2>> $out/${tag}_bench.err
    cp $(find /tmp/pmc_$counter -name '*counter_collection.csv' | head -1) \
        $out/${tag}_bench_pmc_$(echo $counter | tr A-Z a-z).csv
done
# BASELINE configs[2]: Transformer config
cd $repo
python3 bench.py --config transformer --steps 50 --warmup 5 --no-cpu-baseline --no-side --side-records $out/${tag}_transformer_bench_side.json \
    > $out/${tag}_transformer_bench.json 2>> $out/${tag}_bench.err
for precision in bf16x3 bf16x3_fast bf16x6; do
    python3 bench.py --config transformer --precision $precision --steps 50 --warmup 5 --no-cpu-baseline --no-side \
        --side-records $out/${tag}_transformer_${precision}_side.json > $out/${tag}_transformer_$precision.json 2>> $out/${tag}_bench.err
done
cd /tmp
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -- \
    python3 $repo/bench.py --config transformer --steps 20 --warmup 5 --regions 3 --no-cpu-baseline --no-side \
    --streams 1 --side-records /tmp/side_prof.json > /dev/null 2>> $out/${tag}_bench.err
cp $(find /tmp/prof_t -name '*kernel_stats.csv' | head -1) \
    $out/${tag}_transformer_kernel_stats_1stream.csv
for counter in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmct_$counter
    rocprofv3 --pmc $counter --output-format csv -d /tmp/pmct_$counter -- \
        python3 $repo/bench.py --config transformer --steps 4 --warmup 1 --no-preroll --regions 1 --no-cpu-baseline --no-side \
        --streams 1 --no-graph --side-records /tmp/side_prof.json > /dev/null 2>> $out/${tag}_bench.err
    cp $(find /tmp/pmct_$counter -name '*counter_collection.csv' | head -1) \
        $out/${tag}_transformer_pmc_$(echo $counter | tr A-Z a-z).csv
done
# the opt-in precisions under rocprofv3: the split kernels' own durations
for config in conv transformer; do
    rm -rf /tmp/prof_s_$config
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s_$config -- \
        python3 $repo/bench.py --config $config --precision bf16x3 --steps 20 --warmup 5 --regions 3 --no-cpu-baseline \
        --no-api --no-side --streams 1 --side-records /tmp/side_prof.json > /dev/null 2>> $out/${tag}_bench.err
    cp $(find /tmp/prof_s_$config -name '*kernel_stats.csv' | head -1) \
        $out/${tag}_${config}_bf16x3_kernel_stats_1stream.csv
done
ls -la $out | grep $tag
cat $out/${tag}_bench.json
