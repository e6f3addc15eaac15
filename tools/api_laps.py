import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench, emphases_amd
audios, alignments, _ = bench.workload(0)
floats = [torch.from_numpy(a) for a in audios]
for _ in range(8):
    emphases_amd.from_alignments_and_audios(alignments, floats, 16000)
laps = []
for _ in range(60):
    t = time.perf_counter()
    emphases_amd.from_alignments_and_audios(alignments, floats, 16000)
    laps.append((time.perf_counter() - t) * 1e3)
print(' '.join(f'{l:.1f}' for l in laps))
