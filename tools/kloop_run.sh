#!/bin/bash
# Runs on the GPU box: the micro-benchmarks behind DESIGN §6 "What the conv K loop costs".
# usage: tools/kloop_run.sh <tag>   -> gpurun_out/<tag>_kloop.txt   (binaries: see the .hip headers)
tag=${1:-r5}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out/${tag}_kloop.txt
bin=$repo/tools/micro/bin
mkdir -p $repo/gpurun_out
: > $out
echo "== tools/micro/kloop.hip: the conv K loop alone, 150 chunks (a short kernel: the clock of short kernels)" >> $out
for name in kloop kloop_BARRIER kloop_PACKED kloop_NO_TRANSFORM kloop_NO_A_READS kloop_NO_A_READS_NO_TRANSFORM kloop_BARRIER_NO_A_READS_NO_TRANSFORM; do
    echo -n "$name: " >> $out; timeout 60 $bin/$name 150 >> $out 2>&1
done
echo "== the same loop kept up for 1 500 / 6 000 chunks" >> $out
for chunks in 1500 6000; do echo -n "kloop: " >> $out; timeout 60 $bin/kloop $chunks >> $out 2>&1; done
echo "== tools/micro/mfma_rate.hip: a bare MFMA stream on the whole chip" >> $out
timeout 100 $bin/mfma_rate >> $out 2>&1
echo "== tools/micro/clock_probe.hip: what is added to an MFMA stream (2 waves per SIMD, 2.3 ms kernels)" >> $out
timeout 100 $bin/clock_probe >> $out 2>&1
echo "== tools/micro/icache.hip: is once-per-launch code slow to fetch?" >> $out
timeout 60 $bin/icache >> $out 2>&1
cat $out
