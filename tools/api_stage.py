"""Public-API call latency percentiles on the bench workload and the share of the host gather:
python tools/api_stage.py   (CALLS=n; the gather's streaming stores are hard-wired since round 3)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, emphases_amd
from emphases_amd import runtime, session
audios, alignments, _ = bench.workload(0)
floats = [torch.from_numpy(a) for a in audios]
lib = runtime.library()
orig = lib.emph_host_gather
spent = {'gather': 0.0, 'n': 0}
class Wrap:
    def __getattr__(self, name):
        f = getattr(lib, name)
        if name != 'emph_host_gather':
            return f
        def g(*a):
            t = time.perf_counter(); r = f(*a); spent['gather'] += time.perf_counter() - t; spent['n'] += 1; return r
        return g
runtime_library = runtime.library
runtime.library = lambda: Wrap()
for _ in range(8):
    emphases_amd.from_alignments_and_audios(alignments, floats, 16000)
stage0 = session._Lane.stage
st = {'stage': 0.0}
def stage(self, *a, **k):
    t = time.perf_counter(); r = stage0(self, *a, **k); st['stage'] += time.perf_counter() - t; return r
session._Lane.stage = stage
spent['gather'] = 0; spent['n'] = 0
laps = []
records = []
CALLS = int(os.environ.get('CALLS', 40))
PAUSE = float(os.environ.get('PAUSE', 0.002))
for index in range(CALLS):
    if PAUSE:
        time.sleep(PAUSE)
    before = (st['stage'], spent['gather'])
    t = time.perf_counter()
    emphases_amd.from_alignments_and_audios(alignments, floats, 16000)
    laps.append(time.perf_counter() - t)
    records.append((index, laps[-1], st['stage'] - before[0], spent['gather'] - before[1]))
for index, lap, stage_time, gather_time in records:
    if lap > 4e-3:
        print('  slow call %d: %.2f ms (stage %.2f, gather %.2f)' % (
            index, lap * 1e3, stage_time * 1e3, gather_time * 1e3))
laps = np.sort(laps) * 1e3
n = len(laps)
print('call p10 %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f mean %.3f ms; mean stage %.3f gather %.3f' % (
    laps[n // 10], laps[n // 2], laps[n * 9 // 10], laps[n * 99 // 100], laps[-1], laps.mean(),
    st['stage'] / n * 1e3, spent['gather'] / n * 1e3))
print('cpu affinity', len(os.sched_getaffinity(0)), 'threads', os.cpu_count())
