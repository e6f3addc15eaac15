"""Busy / overlap statistics of a rocprofv3 --kernel-trace csv:
python tools/timeline.py <kernel_trace.csv> [skip_first_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
events = sorted(
    (int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'])
    for r in rows if 'emph::' in r['Kernel_Name'])
# steady state: the middle half of the run
events = events[len(events) // 4:3 * len(events) // 4]
span = events[-1][1] - events[0][0]
points = sorted([(s, 1) for s, _, _ in events] + [(e, -1) for _, e, _ in events])
depth, last, busy, overlapped = 0, points[0][0], 0, 0
for time, delta in points:
    if depth >= 1:
        busy += time - last
    if depth >= 2:
        overlapped += time - last
    depth += delta
    last = time
total = sum(e - s for s, e, _ in events)
print(f'{len(events)} kernels over {span / 1e3:.1f} us: some kernel running '
      f'{100 * busy / span:.1f} % of the time, two or more '
      f'{100 * overlapped / span:.1f} %; sum of kernel durations '
      f'{100 * total / span:.1f} % of the span')
by = {}
for s, e, name in events:
    key = name.split('(')[0][-40:]
    by.setdefault(key, []).append(e - s)
for key, values in sorted(by.items(), key=lambda item: -sum(item[1])):
    print(f'  {key:42s} n={len(values):4d} mean {sum(values) / len(values) / 1e3:7.1f} us')
