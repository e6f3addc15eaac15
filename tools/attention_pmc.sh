#!/bin/bash
# Runs on the GPU box: SQ counters of the bf16-split attention kernel (both template
# instances are averaged: read the x-count) and of the fp32 kernel, tools/attention_bench.py.
# usage: tools/attention_pmc.sh <tag>  -> gpurun_out/<tag>_attention_pmc.txt
tag=${1:-r5}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out/${tag}_attention_pmc.txt
mkdir -p $repo/gpurun_out
: > $out
cmd="python3 $repo/tools/attention_bench.py"
for kernel in "attention_split_kernel<40, 2>" "attention_split_kernel<40, 3>" "attention_group_kernel"; do
  echo "== $kernel: matrix pipe" >> $out
  $repo/tools/pmc_kernel.sh "$kernel" SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES -- $cmd >> $out 2>&1
  echo "== $kernel: waits" >> $out
  $repo/tools/pmc_kernel.sh "$kernel" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -- $cmd >> $out 2>&1
  echo "== $kernel: vector / LDS" >> $out
  $repo/tools/pmc_kernel.sh "$kernel" SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -- $cmd >> $out 2>&1
done
cat $out
