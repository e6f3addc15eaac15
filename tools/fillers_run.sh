#!/bin/bash
# Runs on the GPU box: tools/micro/fillers (what hides under an fp32 MFMA) and its SQ counters.
# usage: tools/fillers_run.sh <tag>   -> gpurun_out/<tag>_coexec.txt, <tag>_coexec_pmc.txt
tag=${1:-r5}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out
mkdir -p $out
timeout 300 $repo/tools/micro/bin/fillers > $out/${tag}_coexec.txt 2>&1
cat $out/${tag}_coexec.txt
