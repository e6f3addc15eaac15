#!/bin/bash
# Runs on the GPU box: tools/timer_check.py under rocprofv3's kernel trace.
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tc
rocprofv3 --kernel-trace --output-format csv -d /tmp/tc -- python3 $repo/tools/timer_check.py /tmp/tc.json > $out/timer_check.txt 2> $out/timer_check.err
python3 $repo/tools/timer_check.py --join /tmp/tc.json /tmp/tc >> $out/timer_check.txt 2>> $out/timer_check.err
python3 $repo/tools/timer_check.py /tmp/tc_plain.json >> $out/timer_check.txt 2>> $out/timer_check.err
cat $out/timer_check.txt; tail -5 $out/timer_check.err
