"""One attention launch of BASELINE configs[2] (64 x 1000 frames, 2 heads of 40): the fp32-MFMA
kernel (emph_attention, tile 256) against the bf16-split kernel (emph_attention_split, 2 and 3
pieces), HIP events around 20 back-to-back launches each; worst difference to a float64
reference on sampled queries.   usage (GPU box): python tools/attention_bench.py
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
    count, frames, channels, heads = 64, 1000, 80, 2
    segments = [batch.Segment(i, 0, 1, 0, 0, frames, np.array([[0], [frames]], dtype=np.int64))
                for i in range(count)]
    plan = batch.Plan(segments, [0] * count, [0] * count)
    axis = runtime.AXIS_FRAMES
    host, offsets = plan.pack_metadata([(axis, 256), (axis, 64)])
    buffer = torch.from_numpy(host).to(device)
    start, size = offsets[('tiles', axis, 256)]
    tiles = buffer[start:start + size]
    start64, size64 = offsets[('tiles', axis, 64)]
    tiles64 = buffer[start64:start64 + size64]
    ld = plan.ld_frames
    generator = torch.Generator().manual_seed(5)
    qk = (torch.randn(2 * channels, ld, generator=generator) * 1.5).to(device)
    v = torch.randn(ld, channels, generator=generator).to(device)
    outs = {}

    def run(name, call):
        out = torch.zeros(channels, ld, device=device)
        for _ in range(3):
            call(out)
        torch.cuda.synchronize()
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record()
        for _ in range(20):
            call(out)
        end.record()
        torch.cuda.synchronize()
        outs[name] = out
        flops = 4. * channels * count * frames * frames
        us = begin.elapsed_time(end) / 20 * 1e3
        print(f'{name:28s} {us:8.1f} us per launch   {flops / us * 1e-6:7.1f} TFLOP/s algorithmic')

    run('fp32 MFMA (emph_attention)', lambda out: runtime.check(lib.emph_attention(
        qk.data_ptr(), v.data_ptr(), out.data_ptr(), ld, channels, heads, tiles.data_ptr(),
        size // 4, 256, None, None), 'emph_attention'))
    for pieces in (2, 32, 3):
        images = torch.zeros(lib.emph_split_kv_bytes(ld, count, channels, heads, pieces),
                             dtype=torch.uint8, device=device)

        def split_only(out, pieces=pieces, images=images):
            runtime.check(lib.emph_split_kv(
                qk.data_ptr(), v.data_ptr(), ld, channels, heads, tiles64.data_ptr(), size64 // 4,
                64, pieces, images.data_ptr(), None), 'emph_split_kv')

        def attend_only(out, pieces=pieces, images=images):
            runtime.check(lib.emph_attention_split(
                qk.data_ptr(), images.data_ptr(), out.data_ptr(), ld, channels, heads,
                tiles.data_ptr(), size // 4, 256, None, pieces, None), 'emph_attention_split')

        def both(out):
            split_only(out)
            attend_only(out)
        run(f'split_kv, {pieces} pieces', split_only)
        run(f'attention_split, {pieces} pieces', attend_only)
        run(f'both, {pieces} pieces', both)
    # sampled utterances against float64
    d = channels // heads
    outs = {name: out for name, out in outs.items() if not name.startswith('split_kv')}
    worst = {name: 0. for name in outs}
    for index in (0, 17, 63):
        off = int(plan.frame_off[index])
        q = qk[:channels, off:off + frames].T.reshape(frames, heads, d).double().cpu()
        k = qk[channels:, off:off + frames].T.reshape(frames, heads, d).double().cpu()
        vv = v[off:off + frames].reshape(frames, heads, d).double().cpu()
        scores = torch.einsum('qhd,khd->hqk', q, k) / np.sqrt(d)
        want = torch.einsum('hqk,khd->qhd', torch.softmax(scores, -1), vv).reshape(frames, channels).T
        for name, out in outs.items():
            worst[name] = max(worst[name], float(
                (out[:, off:off + frames].double().cpu() - want).abs().max()))
    for name, value in worst.items():
        print(f'{name:28s} worst |out - float64| = {value:.2e}')


if __name__ == '__main__':
    main()
