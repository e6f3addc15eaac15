"""The product path against the reference's goldens, as a table (the tests assert the same
numbers; this prints them).  Runs on a GPU box: python tools/parity_report.py > gpurun_out/<tag>_parity.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import emphases_amd  # noqa: E402
from emphases_amd import batch, engine as engine_module, synth, weights  # noqa: E402
from conftest import variant_config, variant_state  # noqa: E402


def run(engine, audio, bounds, batch_size):
    words = emphases_amd.Alignment.from_frames(bounds)
    segments = batch.chunk_utterance(words, audio.shape[1], batch_size)
    plan = batch.Plan(segments, [0], [audio.shape[1]])
    tracks = None
    if engine.config.pitch_feature or engine.config.periodicity_feature:
        tracks = torch.from_numpy(batch.pack_tracks(plan, [
            synth.pitch_tracks(batch.chunk_audio(torch.from_numpy(audio[0]), segment))
            for segment in plan.segments])).to(engine.device)
    scores, logits = engine.forward(
        torch.from_numpy(audio[0]).to(engine.device), plan, tracks=tracks)
    columns = plan.word_columns()
    return scores.cpu().numpy()[columns], logits.cpu().numpy()[columns]


def main():
    print(f'{torch.cuda.get_device_name(0)}; goldens: tests/golden/*.npz, captured from the '
          'unmodified reference on its fp32 path (tests/golden/generate.py); tolerance 1e-4 '
          'on scores (BASELINE.json)')
    cases = np.load(os.path.join(ROOT, 'tests', 'golden', 'cases.npz'))
    engine = emphases_amd.get_engine()
    print(f'\ndefault configuration, the shipped checkpoint (conv stack: {engine.stack}, '
          f'per-word sums folded: {engine.fold}, one-launch decoder: {engine.fused_words})')
    print(f'{"case":24s} {"words":>6s} {"max |score - ref|":>18s} {"max |logit - ref|":>18s}')
    worst = 0.
    for name in sorted({key.split('/')[0] for key in cases.files}):
        if f'{name}/pcm' in cases:
            audio = synth.pcm_to_float(cases[f'{name}/pcm'])
        else:
            audio = cases[f'{name}/audio'][None]
        bounds = cases[f'{name}/bounds_frames'].astype(np.int64)
        size = int(cases[f'{name}/batch_size'])
        scores, logits = run(engine, audio, bounds, None if size < 0 else size)
        delta = float(np.abs(scores - cases[f'{name}/scores']).max())
        worst = max(worst, delta)
        print(f'{name:24s} {len(scores):6d} {delta:18.2e} '
              f'{float(np.abs(logits - cases[name + "/logits"]).max()):18.2e}')
    print(f'worst score difference over the cases: {worst:.2e}')
    variants = np.load(os.path.join(ROOT, 'tests', 'golden', 'variants.npz'))
    audio = synth.pcm_to_float(variants['audio_pcm'])
    bounds = variants['bounds_frames'].astype(np.int64)
    print(f'\nvariant matrix, {len(variants["names"])} configurations with SEEDED RANDOM weights: '
          'the output layer of each carries a power-of-two gain\nthat keeps the largest |logit| '
          '(= scale) in (2, 4], so that no score is saturated')
    print(f'{"variant":74s} {"scale":>8s} {"max |score - ref|":>18s} {"max |logit - ref| / scale":>26s}')
    worst = 0.
    for name in variants['names']:
        config, _ = variant_config(name)
        engine = engine_module.Engine(config, variant_state(variants, name, config), 0)
        scores, logits = run(engine, audio, bounds, None)
        want = variants[f'{name}/logits']
        scale = max(1.0, float(np.abs(want).max()))
        delta = float(np.abs(scores - variants[f'{name}/scores']).max())
        worst = max(worst, delta)
        print(f'{str(name):74s} {scale:8.1f} {delta:18.2e} '
              f'{float(np.abs(logits - want).max()) / scale:26.2e}')
    print(f'worst score difference over the variants: {worst:.2e}')


if __name__ == '__main__':
    main()
