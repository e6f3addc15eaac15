"""Developer check on a GPU box: stage-by-stage deltas of the HIP path against
the committed golden vectors, plus rough kernel timings."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emphases_amd  # noqa: E402
from emphases_amd import batch, engine as engine_module, runtime, synth  # noqa: E402


def load_case(cases, name):
    if f'{name}/pcm' in cases:
        audio = synth.pcm_to_float(cases[f'{name}/pcm'])
    else:
        audio = cases[f'{name}/audio'][None]
    bounds = cases[f'{name}/bounds_frames'].astype(np.int64)
    batch_size = int(cases[f'{name}/batch_size'])
    return audio, bounds, None if batch_size < 0 else batch_size


def main():
    print(torch.cuda.get_device_name(0))
    cases = np.load(os.path.join(ROOT, 'tests', 'golden', 'cases.npz'))
    names = sorted({key.split('/')[0] for key in cases.files})
    eng = emphases_amd.get_engine()
    for name in names:
        audio, bounds, batch_size = load_case(cases, name)
        alignment = emphases_amd.Alignment.from_frames(
            bounds, synth.word_names(bounds.shape[1]))
        segments = batch.chunk_utterance(
            alignment, audio.shape[1], batch_size)
        plan = batch.Plan(segments, [0], [audio.shape[1]])
        stages = {}
        device_audio = torch.from_numpy(audio[0]).to(eng.device)
        scores, logits = eng.forward(device_audio, plan, stages=stages)
        torch.cuda.synchronize()
        columns = torch.from_numpy(plan.word_columns()).to(eng.device)
        frame_columns = torch.from_numpy(np.concatenate([
            np.arange(o, o + n) for o, n in
            zip(plan.frame_off, plan.frames)])).to(eng.device)
        report = [f'{name:20s}']
        got_scores = scores[columns].cpu().numpy()
        report.append(
            f'score {np.abs(got_scores - cases[name + "/scores"]).max():.2e}')
        report.append('logit %.2e' % np.abs(
            logits[columns].cpu().numpy() - cases[name + '/logits']).max())
        down = stages['downsampled'][:, columns].cpu().numpy()
        report.append('down %.2e' % np.abs(
            down - cases[name + '/downsampled']).max())
        for key, stage in (('mel', 'features'), ('input_layer', 'input_layer'),
                           ('encoder', 'encoder')):
            got = stages[stage][:, frame_columns].cpu().numpy()
            if f'{name}/{key}' in cases:
                want = cases[f'{name}/{key}']
            elif f'{name}/{key}_stride7' in cases:
                want = cases[f'{name}/{key}_stride7']
                got = got[:, ::7]
            else:
                continue
            report.append(f'{key} {np.abs(got - want).max():.2e}')
        print('  '.join(report), flush=True)

    # rough timing of the C2 workload
    count, frames = 64, 1000
    audios = [torch.from_numpy(synth.audio(i, frames)) for i in range(count)]
    aligns = [emphases_amd.Alignment.from_frames(
        synth.word_frames(i, frames)) for i in range(count)]
    lengths = [frames * 160] * count
    offsets = np.arange(count) * frames * 160
    segments = []
    for i, a in enumerate(aligns):
        segments.extend(batch.chunk_utterance(a, lengths[i], None, i))
    plan = batch.Plan(segments, offsets, lengths)
    packed = torch.cat([a.reshape(-1) for a in audios]).to(eng.device)
    for tile in (16, 32, 64):
        meta = eng.upload(plan, tile)
        for _ in range(3):
            eng.forward(packed, plan, meta)
        torch.cuda.synchronize()
        start = time.perf_counter()
        steps = 20
        for _ in range(steps):
            eng.forward(packed, plan, meta)
        torch.cuda.synchronize()
        elapsed = (time.perf_counter() - start) / steps
        print(f'C2 tile {tile}: {elapsed * 1e3:.3f} ms/step  '
              f'{count / elapsed:.0f} utt/s  {count * frames / elapsed / 1e6:.1f} Mframes/s')


if __name__ == '__main__':
    main()
