#!/bin/bash
# usage: tools/kernel_clock.sh <kernel-substring> -- <program> [args]
# Average shader clock while a kernel runs: GRBM_GUI_ACTIVE (cycles) / duration.
pattern=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
out=/tmp/clk_$$
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out -- "$@" > /dev/null 2>&1
python3 - "$pattern" $out <<'PY'
import csv, glob, sys, collections
pattern, root = sys.argv[1], sys.argv[2]
sums = collections.defaultdict(float); counts = collections.Counter(); dur = {}
for path in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        if pattern in row['Kernel_Name']:
            sums[row['Counter_Name']] += float(row['Counter_Value'])
            counts[row['Counter_Name']] += 1
            if 'Start_Timestamp' in row:
                dur[row['Dispatch_Id']] = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
for path in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        if pattern in row['Kernel_Name']:
            dur[row['Dispatch_Id']] = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
mean = {k: sums[k] / counts[k] for k in sums}
for name in sorted(mean):
    print(f'{name:32s} {mean[name]:16.1f}  (x{counts[name]})')
if dur:
    d = sorted(dur.values()); med = d[len(d) // 2]
    print(f'median duration {med / 1e3:.1f} us over {len(d)} dispatches')
    if 'GRBM_GUI_ACTIVE' in mean:
        print(f'clock ~ {mean["GRBM_GUI_ACTIVE"] / med:.2f} GHz (GRBM_GUI_ACTIVE / duration)')
PY
