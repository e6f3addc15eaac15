#!/bin/bash
# Runs on the GPU box: matrix-pipe / vector / LDS utilisation counters of the round's
# kernels (one rocprofv3 --pmc pass each; counters of a pass must fit together).
# usage: tools/pmc_round.sh <tag>   -> gpurun_out/<tag>_pmc_utilisation.txt
tag=${1:-r6}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out/${tag}_pmc_utilisation.txt
mkdir -p $repo/gpurun_out
: > $out
conv="python3 $repo/bench.py --steps 10 --warmup 2 --no-preroll --regions 1 --no-cpu-baseline --no-api --no-side --streams 1"
tr="python3 $repo/bench.py --config transformer --steps 4 --warmup 1 --no-preroll --regions 1 --no-cpu-baseline --no-side --streams 1"
echo "== conv1d_stack_kernel: matrix pipe" >> $out
$repo/tools/pmc_kernel.sh conv1d_stack_kernel SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- $conv >> $out 2>&1
echo "== conv1d_stack_kernel: waits" >> $out
$repo/tools/pmc_kernel.sh conv1d_stack_kernel SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -- $conv >> $out 2>&1
echo "== frontend_kernel: vector / LDS" >> $out
$repo/tools/pmc_kernel.sh frontend_kernel SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- $conv >> $out 2>&1
echo "== frontend_kernel: waits" >> $out
$repo/tools/pmc_kernel.sh frontend_kernel SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -- $conv >> $out 2>&1
echo "== frontend_kernel: busy cycles" >> $out
$repo/tools/pmc_kernel.sh frontend_kernel SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- $conv >> $out 2>&1
echo "== attention_group_kernel: matrix pipe" >> $out
$repo/tools/pmc_kernel.sh attention_group_kernel SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- $tr >> $out 2>&1
cat $out
python3 $repo/tools/pmc_utilisation_json.py $repo/gpurun_out $tag > /dev/null
