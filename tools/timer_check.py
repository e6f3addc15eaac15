"""How `runtime.LaunchTimer` (events bound to a kernel's own dispatch packet) compares
with what rocprofv3's kernel trace reads for the SAME dispatches.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tc -- python3 tools/timer_check.py out.json
    python3 tools/timer_check.py --join out.json /tmp/tc      # per-mode table

Modes (BASELINE configs[1] batch, one stream): eager passes with the timer and the
engine's recorded events; eager passes with the timer alone (no recorded events around
the launches); the same right behind 300 graph replays (clocks up); graph replays
(no timer: only the trace sees them)."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out):
    import torch
    import bench
    from emphases_amd import config as cfg, runtime
    device = torch.device('cuda', 0)
    audios, alignments, _ = bench.workload(0)
    runner = bench.Runner(cfg.DEFAULT, None, device, audios, alignments, streams=1)
    engine = runner.engine
    record = {'modes': []}

    def eager(name, passes, events, hot):
        if hot:
            for _ in range(300):
                runner.step()
        engine.timers = [] if events else None
        # (timers None -> the one-call path; keep the step-by-step path in both)
        with runtime.LaunchTimer(1 << 14) as timer:
            for _ in range(passes):
                if events:
                    engine.forward(runner.packed, runner.plan, runner.meta)
                else:
                    engine.timers = []
                    engine.forward(runner.packed, runner.plan, runner.meta)
            torch.cuda.synchronize()
        engine.timers = None
        per_pass = timer.launches // passes
        table = timer.microseconds.reshape(passes, per_pass)
        record['modes'].append({
            'name': name, 'passes': passes, 'launches_per_pass': per_pass,
            'mean_us': table.mean(axis=0).tolist(),
            'median_us': np.median(table, axis=0).tolist()})
        print(name, per_pass, np.round(np.median(table, axis=0), 2).tolist(), flush=True)

    for _ in range(3):
        engine.forward(runner.packed, runner.plan, runner.meta)
    torch.cuda.synchronize()
    eager('eager_cold', 50, True, False)
    eager('eager_hot', 50, True, True)
    eager('eager_hot_again', 50, True, True)
    for _ in range(500):
        runner.step()
    torch.cuda.synchronize()
    record['graph_replays'] = 300 + 300 + 500
    with open(out, 'w') as file:
        json.dump(record, file)


def join(out, trace_dir):
    """The trace's durations of the same dispatches, in order."""
    import csv
    record = json.load(open(out))
    path = glob.glob(os.path.join(trace_dir, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = []
    with open(path) as file:
        for row in csv.DictReader(file):
            if 'emph::' in row['Kernel_Name']:
                rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp']),
                             row['Kernel_Name'].split('emph::')[1].split('(')[0]))
    rows.sort()
    names = [r[2] for r in rows]
    durations = np.array([(r[1] - r[0]) * 1e-3 for r in rows])
    gaps = np.array([0.] + [(rows[i][0] - rows[i - 1][1]) * 1e-3 for i in range(1, len(rows))])
    print('dispatches in trace', len(rows))
    # the sequence: 3 warm passes (K launches each), then per mode [300 replays] + passes
    per = record['modes'][0]['launches_per_pass']
    at = 3 * per
    for mode in record['modes']:
        if mode['name'] != 'eager_cold':
            at += 300 * per
        count = mode['passes'] * per
        table = durations[at:at + count].reshape(mode['passes'], per)
        idle = gaps[at:at + count].reshape(mode['passes'], per)
        print(mode['name'])
        print('  kernels      ', names[at:at + per])
        print('  trace  median', np.round(np.median(table, axis=0), 2).tolist())
        print('  timer  median', np.round(mode['median_us'], 2).tolist())
        print('  gap in front ', np.round(np.median(idle, axis=0), 2).tolist())
        at += count
    table = durations[at:at + 500 * per].reshape(500, per)
    idle = gaps[at:at + 500 * per].reshape(500, per)
    print('graph replays (trace only)')
    print('  kernels      ', names[at:at + per])
    print('  trace  median', np.round(np.median(table[100:], axis=0), 2).tolist())
    print('  trace  mean  ', np.round(table[100:].mean(axis=0), 2).tolist())
    print('  gap in front ', np.round(np.median(idle[100:], axis=0), 2).tolist())


if __name__ == '__main__':
    if sys.argv[1] == '--join':
        join(sys.argv[2], sys.argv[3])
    else:
        main(sys.argv[1])
