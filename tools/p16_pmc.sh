#!/bin/bash
# Runs on the GPU box: where the time of position_wise16_kernel (csrc/block_split16.hip) goes -
# wave / wait cycles, vector / matrix / LDS instruction counts, HBM bytes - on BASELINE
# configs[2] at precision bf16x3; and the conv + front-end co-residency micro-benchmark.
# usage: tools/p16_pmc.sh <tag>   -> gpurun_out/<tag>_p16_pmc.txt, <tag>_corun_split.txt
tag=${1:-r6}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out/${tag}_p16_pmc.txt
mkdir -p $repo/gpurun_out
: > $out
run="python3 $repo/bench.py --config transformer --precision bf16x3 --steps 4 --warmup 1 --no-preroll --regions 1 --no-cpu-baseline --no-side --no-api --streams 1 --no-graph --side-records /tmp/side_pmc.json"
kernel=position_wise16_kernel
echo "== $kernel: waves and waits" >> $out
$repo/tools/pmc_kernel.sh $kernel SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $run >> $out 2>&1
echo "== $kernel: vector and matrix" >> $out
$repo/tools/pmc_kernel.sh $kernel SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- $run >> $out 2>&1
echo "== $kernel: LDS" >> $out
$repo/tools/pmc_kernel.sh $kernel SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -- $run >> $out 2>&1
echo "== $kernel: scalar, memory instructions" >> $out
$repo/tools/pmc_kernel.sh $kernel SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM -- $run >> $out 2>&1
for counter in FETCH_SIZE WRITE_SIZE; do
    echo "== $kernel: $counter (KB units of the counter; FETCH doubles on gfx950 per the guide)" >> $out
    $repo/tools/pmc_kernel.sh $kernel $counter -- $run >> $out 2>&1
done
cat $out
co=$repo/gpurun_out/${tag}_corun_split.txt
: > $co
for w in 256 128; do $repo/tools/micro/bin/corun_split$w >> $co 2>&1; done
cat $co
