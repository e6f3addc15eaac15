"""What a call through the `torch.ops.emphases_amd.*` seams costs against `Engine.forward`
on the same batch (BASELINE configs[1]: 64 x 10 s): the ops convert between the caller's
back-to-back layout and the library's packed one segment by segment in Python
(`ops._scatter` / `_gather`) and build their tables per call.

    python tools/ops_cost.py          -> one JSON object on stdout
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import emphases_amd  # noqa: E402
from emphases_amd import config as cfg  # noqa: E402


def clock(call, rounds=30):
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    laps = []
    for _ in range(rounds):
        start = time.perf_counter()
        call()
        torch.cuda.synchronize()
        laps.append(time.perf_counter() - start)
    return float(np.median(laps)) * 1e3


def main():
    device = torch.device('cuda', 0)
    audios, alignments, bounds = bench.workload(0)
    engine = emphases_amd.get_engine(None, 0)
    plan = bench.build_plan(audios, alignments)
    packed = torch.cat([torch.from_numpy(a).reshape(-1) for a in audios]).to(device)
    meta = engine.upload(plan)
    result = {'workload': '64 x 10 s, conv config, audio resident on the device'}
    result['engine_forward_eager_ms'] = clock(
        lambda: engine.forward(packed, plan, meta))
    replay, _, _ = engine.capture(packed, plan, meta)
    result['engine_graph_replay_ms'] = clock(replay)
    samples = torch.tensor([0] + [a.shape[1] for a in audios]).cumsum(0)
    words = torch.tensor([0] + [b.shape[1] for b in bounds]).cumsum(0)
    all_bounds = torch.from_numpy(np.concatenate(bounds, axis=1))
    ops = torch.ops.emphases_amd
    result['op_prominence_forward_ms'] = clock(
        lambda: ops.prominence_forward(packed, samples, all_bounds, words))
    result['op_logmel_ms'] = clock(lambda: ops.logmel(packed, samples))
    mel = ops.logmel(packed, samples)
    frames = torch.tensor([0] + [1000] * len(audios)).cumsum(0)
    weight = torch.from_numpy(engine.state['input_layer.weight']).to(device)
    bias = torch.from_numpy(engine.state['input_layer.bias']).to(device)
    result['op_conv1d_same_act_ms'] = clock(
        lambda: ops.conv1d_same_act(mel, weight, bias, frames, 'relu'))
    hidden = ops.conv1d_same_act(mel, weight, bias, frames, 'relu')
    result['op_segment_reduce_ms'] = clock(
        lambda: ops.segment_reduce(hidden, all_bounds, frames, words, 'sum'))
    # the layout conversion alone: 64 slices in, 64 out
    from emphases_amd import ops as module
    result['scatter_plus_gather_ms'] = clock(lambda: module._gather(
        module._scatter(mel, plan, plan.frame_off, plan.frames, plan.ld_frames),
        plan.frame_off, plan.frames))
    print(json.dumps(result))


if __name__ == '__main__':
    main()
