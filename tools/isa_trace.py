"""Compact trace of a loop in hipcc -S output: memory ops and waits in order,
with the number of VALU instructions between them.
python tools/isa_trace.py file.s <kernel-substring> <first-line> <last-line>  (lines relative to kernel start, as printed by isa_mix.py)"""
import re
import sys
text = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith('_Z') and key in l)
lo, hi = int(sys.argv[3]), int(sys.argv[4])
valu = 0
out = []
for line in text[start + lo:start + hi + 1]:
    parts = line.strip().split(None, 1)
    if not parts or parts[0].startswith((';', '.')):
        continue
    op = parts[0]
    if op.endswith(':'):
        out.append(f'[{valu}] {op}'); valu = 0
        continue
    if op.startswith('v_'):
        valu += 1
        continue
    if op.startswith(('ds_', 'global_', 's_waitcnt', 's_cbranch', 's_branch', 's_barrier')):
        arg = parts[1].split(';')[0].strip() if len(parts) > 1 else ''
        if op.startswith('ds_') or op.startswith('global_'):
            arg = ''
        out.append(f'[{valu}] {op} {arg}'.strip()); valu = 0
print('\n'.join(out))
