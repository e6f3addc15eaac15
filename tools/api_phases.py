"""Where a public-API call spends its host time, call by call:
python tools/api_phases.py corpus|longform|bench [calls]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emphases_amd  # noqa: E402
from emphases_amd import batch, config as cfg, dist, engine, session, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'corpus'
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch_size = None
if which == 'corpus':
    frames = synth.corpus_frames(10000, 200, 3000)
    frames = frames[dist.assign(dist.cost(frames), 8)[0]]
    pool = [torch.from_numpy(synth.audio(7000 + i, 3000)) for i in range(40)]
elif which == 'longform':
    frames = np.full(8, 30000)
    pool = [torch.from_numpy(synth.audio(7000 + i, 30000)) for i in range(8)]
    batch_size = 3000
else:
    frames = np.full(64, 1000)
    pool = [torch.from_numpy(synth.audio(i, 1000)) for i in range(64)]
audios = [pool[i % len(pool)][:, :int(n) * cfg.HOPSIZE]
          for i, n in enumerate(frames)]
alignments = [emphases_amd.Alignment.from_frames(
    synth.word_frames(5000 + i, int(n))) for i, n in enumerate(frames)]

spent = {}


def timed(owner, name):
    inner = getattr(owner, name)

    def wrapper(*args, **kwargs):
        start = time.perf_counter()
        try:
            return inner(*args, **kwargs)
        finally:
            spent[name] = spent.get(name, 0.) + time.perf_counter() - start
    setattr(owner, name, wrapper)


timed(session._Lane, 'stage')
timed(session._Lane, '_reserve')
timed(session, 'layout_key')
timed(batch, 'plan_batch')
timed(engine.Engine, 'upload')
timed(engine.Engine, 'capture')
timed(engine.Engine, 'forward')
timed(session.Pending, 'result')
timed(session.Session, 'submit')
for call in range(calls):
    spent.clear()
    start = time.perf_counter()
    emphases_amd.from_alignments_and_audios(
        alignments, audios, 16000, batch_size=batch_size, gpu=0)
    total = time.perf_counter() - start
    print(f'call {call}: {total * 1e3:7.2f} ms  ' + '  '.join(
        f'{k} {v * 1e3:.2f}' for k, v in spent.items()), flush=True)
