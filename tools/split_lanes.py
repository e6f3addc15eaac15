"""Experiment: the word-rate half of batch k - 1 and the frame-rate half of batch k as two
independent branches of ONE HIP graph per step, so that the decoder (77 CUs for 32 us) runs
under the next front-end instead of in front of it.  A lane alternates between two engines
(the branches of a graph touch different workspaces).
python tools/split_lanes.py [lanes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, arguments = sys.argv[:1], sys.argv[1:]
import torch

import bench
import emphases_amd
from emphases_amd import config as cfg

n_lanes = int(arguments[0]) if arguments else 2
device = torch.device('cuda', 0)
audios, alignments, _ = bench.workload(0)
plan = bench.build_plan(audios, alignments)
packed = torch.cat([torch.from_numpy(a).reshape(-1) for a in audios]).to(device)


def clock(step, steps=400, laps=7):
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    times, host = [], []
    for _ in range(laps):
        start = time.perf_counter()
        for _ in range(steps):
            step()
        host.append((time.perf_counter() - start) / steps * 1e6)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - start) / steps * 1e6)
    print(f'   (host side of a step: {sorted(host)[len(host) // 2]:.1f} us)')
    return sorted(times)[len(times) // 2]


baseline = bench.Runner(cfg.DEFAULT, None, device, audios, alignments, streams=2)
print(f'one graph per batch, two lanes:          {clock(baseline.step):7.1f} us per step')
reference = baseline.step().clone()
torch.cuda.synchronize()

lanes = []
for index in range(n_lanes):
    stream = torch.cuda.Stream(device=device)
    side = torch.cuda.Stream(device=device)
    pair = []
    for _ in range(2):
        engine = emphases_amd.engine.Engine(cfg.DEFAULT, None, device)
        meta = engine.upload(plan)
        with torch.cuda.stream(stream):
            for _ in range(2):                       # buffers, attribute calls
                engine.forward_frames(packed, plan, meta)
                scores, _ = engine.forward_words(plan, meta)
        pair.append((engine, meta, scores))
    torch.cuda.synchronize()
    graphs = []
    for turn in range(2):                            # frames of pair[turn] || words of pair[1 - turn]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    engine, meta, _ = pair[1 - turn]
                    engine.forward_words(plan, meta)
                engine, meta, _ = pair[turn]
                engine.forward_frames(packed, plan, meta)
                torch.cuda.current_stream().wait_stream(side)
        graphs.append(graph)
    lanes.append((stream, pair, graphs, [0]))
torch.cuda.synchronize()
counter = [0]


def step():
    """Frames of this step's batch; the scores returned are those of the lane's PREVIOUS batch."""
    stream, pair, graphs, turn = lanes[counter[0] % n_lanes]
    counter[0] += 1
    with torch.cuda.stream(stream):
        graphs[turn[0]].replay()
    scores = pair[1 - turn[0]][2]
    turn[0] ^= 1
    return scores


print(f'{n_lanes} lanes, words(k-1) || frames(k) per graph: {clock(step):7.1f} us per step')
for _ in range(2 * n_lanes):
    scores = step()
torch.cuda.synchronize()
print('same bits as the one-graph path:', bool(torch.equal(scores[baseline.columns], reference[baseline.columns])))
