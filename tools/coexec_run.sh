#!/bin/bash
# Runs on the GPU box: tools/micro/coexec and its SQ counters.
# usage: tools/coexec_run.sh <tag>   -> gpurun_out/<tag>_coexec.txt, <tag>_coexec_pmc.txt
tag=${1:-r4}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out
mkdir -p $out
$repo/tools/micro/bin/coexec > $out/${tag}_coexec.txt 2>&1
cat $out/${tag}_coexec.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/coexec_pmc
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA --output-format csv -d /tmp/coexec_pmc -- $repo/tools/micro/bin/coexec > /dev/null 2>&1
python3 - /tmp/coexec_pmc > $out/${tag}_coexec_pmc.txt <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
rows = collections.OrderedDict()
for path in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        key = (int(row['Dispatch_Id']), row['Kernel_Name'][:60], row.get('Workgroup_Size', ''))
        rows.setdefault(key, {})[row['Counter_Name']] = float(row['Counter_Value'])
names = sorted({n for r in rows.values() for n in r})
print('dispatch kernel workgroup ' + ' '.join(names))
for key in sorted(rows):
    print(key[0], key[1], key[2], ' '.join(f'{rows[key].get(n, 0):.0f}' for n in names))
PY
cat $out/${tag}_coexec_pmc.txt
