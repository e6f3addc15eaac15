"""Instruction mix of the hottest loop of a kernel in hipcc -S output.
python tools/isa_mix.py file.s <kernel-substring> [min_loop_len]"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith('_Z') and key in l)
end = next(i for i in range(start, len(text)) if text[i].strip().startswith('s_endpgm'))
body = text[start:end]
labels = {}
for i, line in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', line)
    if m:
        labels[m.group(1)] = i
loops = []
for i, line in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', line) or re.search(r's_branch\s+(\.LBB\d+_\d+)', line)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def classify(op):
    if op.startswith('v_pk_'): return 'valu_pk'
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_'): return 'salu'
    return 'other'
for lo, hi in sorted(loops, key=lambda p: p[1] - p[0]):
    if hi - lo < int(sys.argv[3]) if len(sys.argv) > 3 else 50:
        continue
    mix = collections.Counter()
    ops = collections.Counter()
    for line in body[lo:hi + 1]:
        parts = line.strip().split()
        if not parts or parts[0].startswith((';', '.')) or parts[0].endswith(':'):
            continue
        mix[classify(parts[0])] += 1
        ops[parts[0]] += 1
    print(f'loop lines {lo}..{hi}: {dict(mix)}')
    print('   ', ops.most_common(24))
