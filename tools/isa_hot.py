"""Dynamic instruction mix of the front-end's per-frame loop on its hot path
(interior frames: the rare edge-frame branch is skipped).
python tools/isa_hot.py file.s <kernel-substring>"""
import collections, re, sys
text = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith('_Z') and key in l and ': ' in l)
end = next(i for i in range(start, len(text)) if text[i].strip().startswith('s_endpgm'))
body = text[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
# the frame loop = the backward branch with the largest span that contains v_sqrt
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        lo = labels[m.group(1)]
        if any('v_sqrt' in x for x in body[lo:i]):
            loops.append((i - lo, lo, i))
span, lo, hi = min(loops)
loop = body[lo:hi + 1]
# skip forward branches whose target lies inside the loop and that jump over > 100 lines (taken on the hot path)
out = []
i = 0
while i < len(loop):
    l = loop[i]
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels:
        target = labels[m.group(1)] - lo
        if target > i + 100 and target < len(loop):
            out.append(l); i = target; continue
    out.append(l); i += 1
c = collections.Counter()
for l in out:
    l = l.strip()
    if not l or l.startswith(('.', ';')): continue
    c[l.split()[0]] += 1
group = lambda p: sum(v for k, v in c.items() if k.startswith(p))
print(f'loop lines {lo}..{hi} ({hi-lo}), hot path {len(out)} lines: VALU {group("v_")} (packed {group("v_pk")}), '
      f'LDS {group("ds_")}, VMEM {group("global_")}, SALU {group("s_")}')
print(sorted(c.items(), key=lambda kv: -kv[1]))
