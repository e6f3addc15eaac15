#!/bin/bash
# Runs on the GPU box: HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and matrix-pipe
# counters of the split Transformer kernels of BASELINE configs[2] at precision bf16x3.
# usage: tools/split_pmc.sh <tag>   -> gpurun_out/<tag>_split_pmc.txt
tag=${1:-r6}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out/${tag}_split_pmc.txt
mkdir -p $repo/gpurun_out
: > $out
run="python3 $repo/bench.py --config transformer --precision bf16x3 --steps 4 --warmup 1 --no-preroll --regions 1 --no-cpu-baseline --no-side --no-api --streams 1 --no-graph --side-records /tmp/side_pmc.json"
# (position_wise16_kernel: five launches of block + next projections, one of the first
# layer's projections and one of the last layer's block per step - the mean is over all)
for kernel in position_wise16_kernel attention_split_kernel; do
    for counter in FETCH_SIZE WRITE_SIZE; do
        echo "== $kernel: $counter (KB units of the counter; FETCH doubles on gfx950 per the guide)" >> $out
        $repo/tools/pmc_kernel.sh $kernel $counter -- $run >> $out 2>&1
    done
    echo "== $kernel: matrix pipe" >> $out
    $repo/tools/pmc_kernel.sh $kernel SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- $run >> $out 2>&1
done
# the frame-rate convs of configs[1] at bf16x3
run="python3 $repo/bench.py --precision bf16x3 --steps 10 --warmup 2 --no-preroll --regions 1 --no-cpu-baseline --no-side --no-api --streams 1 --no-graph --side-records /tmp/side_pmc.json"
kernel=conv1d_split_kernel
for counter in FETCH_SIZE WRITE_SIZE; do
    echo "== $kernel: $counter (KB units of the counter; FETCH doubles on gfx950 per the guide)" >> $out
    $repo/tools/pmc_kernel.sh $kernel $counter -- $run >> $out 2>&1
done
echo "== $kernel: matrix pipe" >> $out
$repo/tools/pmc_kernel.sh $kernel SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- $run >> $out 2>&1
echo "== $kernel: vector and LDS" >> $out
$repo/tools/pmc_kernel.sh $kernel SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES -- $run >> $out 2>&1
cat $out
