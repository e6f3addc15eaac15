"""The seven frame-rate layers of BASELINE configs[1] (64 x 1000 frames) as three launches:
emph_conv1d_stack (fp32 MFMA, F(4,3)) against emph_conv1d_split (bf16x3, direct form);
HIP events around 20 repetitions.   usage (GPU box): python tools/conv_split_bench.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emphases_amd import batch, runtime, synth  # noqa: E402


def main():
    lib = runtime.library()
    device = 'cuda:0'
    count, frames = 64, 1000
    segments = [batch.Segment(i, 0, 1, 0, 0, frames, np.array([[0], [frames]], dtype=np.int64))
                for i in range(count)]
    plan = batch.Plan(segments, [0] * count, [0] * count)
    spans_host = plan.conv_spans()
    spans = torch.from_numpy(spans_host).to(device)
    ld = plan.ld_frames
    weights = [synth.weights(40 + l, (80, 80, 3), 0.12) for l in range(7)]
    biases = torch.from_numpy(np.concatenate([synth.weights(50 + l, (80,), 0.3) for l in range(7)])).to(device)
    plain = torch.from_numpy(np.concatenate([runtime.conv_winograd4_pack(w) for w in weights])).to(device)
    split = torch.from_numpy(np.concatenate([runtime.conv_split_pack(w) for w in weights])).to(device)
    x = torch.randn(80, ld, generator=torch.Generator().manual_seed(3)).to(device)
    a, b = torch.zeros_like(x), torch.zeros_like(x)

    def run(name, call):
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record()
        for _ in range(20):
            call()
        end.record()
        torch.cuda.synchronize()
        print(f'{name:44s} {begin.elapsed_time(end) / 20 * 1e3:8.1f} us for the seven layers')

    def stack(function, packs, pack_stride, extra, groups=(3, 2, 2)):
        source, done = x, 0
        for group, size in enumerate(groups):
            target = (a, b)[group & 1]
            relu = sum(1 << l for l in range(size) if done + l >= 1)
            runtime.check(function(
                source.data_ptr(), ld, target.data_ptr(), ld,
                packs[done * pack_stride:].data_ptr(), biases[done * 80:].data_ptr(), size, relu,
                spans.data_ptr(), len(spans_host), *extra), 'conv')
            source, done = target, done + size
        return source

    pack_floats = plain.numel() // 7
    pack_bytes = split.numel() // 7
    run('emph_conv1d_stack (fp32 MFMA, F(4,3))', lambda: stack(lib.emph_conv1d_stack, plain, pack_floats, (None, None)))
    want = stack(lib.emph_conv1d_stack, plain, pack_floats, (None, None)).clone()
    run('emph_conv1d_split (bf16x3, direct form) 3+2+2', lambda: stack(lib.emph_conv1d_split, split, pack_bytes, (None, None)))
    run('emph_conv1d_split (bf16x3, direct form) 4+3', lambda: stack(lib.emph_conv1d_split, split, pack_bytes, (None, None), (4, 3)))
    run('emph_conv1d_split (bf16x3, direct form) 5+2', lambda: stack(lib.emph_conv1d_split, split, pack_bytes, (None, None), (5, 2)))
    got = stack(lib.emph_conv1d_split, split, pack_bytes, (None, None), (4, 3)).clone()
    columns = np.concatenate([np.arange(o, o + n) for o, n in zip(plan.frame_off, plan.frames)])
    delta = (got[:, columns] - want[:, columns]).abs().max().item()
    print(f'worst |split - fp32| after seven layers {delta:.2e} at scale {want[:, columns].abs().max().item():.1f}')


if __name__ == '__main__':
    main()
