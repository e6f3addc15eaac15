"""A window of a rocprofv3 --kernel-trace csv as text: one line per kernel, times in us from
the window's start, one column per queue.  python tools/timeline_dump.py <trace.csv> [kernels]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 36
events = sorted(
    (int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'))
    for r in rows if 'emph::' in r['Kernel_Name'])
middle = len(events) // 2
window = events[middle:middle + count]
origin = window[0][0]
queues = sorted({q for _, _, _, q in window})
for start, end, name, queue in window:
    short = name.split('emph::')[1].split('(')[0][:34]
    column = queues.index(queue)
    print(f'{(start - origin) / 1e3:8.1f} {(end - origin) / 1e3:8.1f} {(end - start) / 1e3:6.1f}  '
          + ' ' * (36 * column) + short)
