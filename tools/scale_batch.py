"""Frames/s of one device as the ragged batch grows (per-launch fixed costs
amortise): python tools/scale_batch.py [counts...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import emphases_amd  # noqa: E402

device = torch.device('cuda', 0)
engine = emphases_amd.engine.Engine(emphases_amd.config.DEFAULT, None, device)
for count in [int(a) for a in sys.argv[1:]] or [64, 256, 1024]:
    audios, alignments, _ = bench.workload(0, count=min(count, 64))
    audios = [audios[i % len(audios)] for i in range(count)]
    alignments = [alignments[i % len(alignments)] for i in range(count)]
    plan = bench.build_plan(audios, alignments)
    packed = torch.cat(
        [torch.from_numpy(a).reshape(-1) for a in audios]).to(device)
    meta = engine.upload(plan)
    replay, scores, _ = engine.capture(packed, plan, meta)
    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    steps = max(3, 2000 // count)
    start = time.perf_counter()
    for _ in range(steps):
        replay()
    torch.cuda.synchronize()
    elapsed = (time.perf_counter() - start) / steps
    print(f'{count:5d} x 10 s: {elapsed * 1e3:8.3f} ms  '
          f'{count / elapsed:10.0f} utterances/s  tile {meta["tile"]}', flush=True)
