#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-name-substring> <counters...> -- <program> [args]
# One rocprofv3 --pmc pass (counters must fit one pass); prints the per-dispatch
# mean of each counter for kernels whose name contains the substring.
pattern=$1; shift
counters=()
while [ "$1" != "--" ]; do counters+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
out=/tmp/pmc_$$
rocprofv3 --pmc "${counters[@]}" --output-format csv -d $out -- "$@" > /dev/null 2>&1
python3 - "$pattern" $out <<'PY'
import csv, glob, sys, collections
pattern, root = sys.argv[1], sys.argv[2]
sums = collections.defaultdict(float); counts = collections.Counter()
for path in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        if pattern in row['Kernel_Name']:
            sums[row['Counter_Name']] += float(row['Counter_Value'])
            counts[row['Counter_Name']] += 1
for name in sorted(sums):
    print(f'{name:36s} {sums[name] / counts[name]:16.1f}  (x{counts[name]})')
PY
