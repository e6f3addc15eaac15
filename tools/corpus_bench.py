"""BASELINE configs[3] and [4] at FULL size on the one MI355X of a gpurun box.

    python tools/corpus_bench.py [corpus] [longform] > gpurun_out/corpus.json

configs[3]: 10 000 utterances of 2-30 s (F_i ~ U{200..3000},
about 16 M frames and 0.47 M words).  Measured: the WHOLE corpus as one ragged
batch on one GPU (10 GB of audio and 2 x 5 GB of activations resident: what the
288 GB are for), and rank 0's share under the 8-rank LPT assignment of
`dist.assign` — device-only (graph replay, audio resident) and through the
public API (pageable host tensors in, scores out).

configs[4]: 5-minute utterances (30 000 frames) chunked at batch_size = 3000
frames: 8 per rank (64 per node) and 64 on one GPU, the same two protocols.

The audio of utterance i is a prefix of one of 40 distinct 30 s signals (the
generator of SURVEY §8d costs 50 ns per sample; 2.6 G samples would take
minutes); alignments are all distinct.  One JSON object per line.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emphases_amd  # noqa: E402
from emphases_amd import batch, config as cfg, dist, synth  # noqa: E402

BYTES_PER_FRAME = 641.        # SURVEY §8d compulsory traffic
PEAK_HBM = 8.0e12
FLOPS_PER_FRAME, FLOPS_PER_WORD = 0.2968e6, 0.2309e6
PEAK_MFMA = 157.3e12
POOL = 40


def device_only(engine, pool_device, picks, frames, alignments, batch_size):
    """Seconds per pass with the packed audio resident, replayed as a graph."""
    lengths = [int(n) * cfg.HOPSIZE for n in frames]
    plan = batch.plan_batch(alignments, lengths, batch_size)
    packed = torch.cat([pool_device[p][:n] for p, n in zip(picks, lengths)])
    meta = engine.upload(plan)
    replay, scores, _ = engine.capture(packed, plan, meta)
    for _ in range(2):
        replay()
    torch.cuda.synchronize()
    steps = max(3, int(3e6 // max(1, int(np.sum(frames)))))
    start = time.perf_counter()
    for _ in range(steps):
        replay()
    torch.cuda.synchronize()
    elapsed = (time.perf_counter() - start) / steps
    columns = plan.word_columns()
    values = scores[torch.as_tensor(columns, device=scores.device)]
    assert bool(torch.isfinite(values).all())
    checksum = float(values.double().sum())
    del replay, scores, packed, values
    return elapsed, plan.total_words, plan.total_frames, checksum


def through_api(pool_host, picks, frames, alignments, batch_size, rounds=5):
    audios = [pool_host[p][:, :int(n) * cfg.HOPSIZE]
              for p, n in zip(picks, frames)]
    # (a layout's graph is captured at its second sighting on a lane, and the
    # sub-batches of a large call alternate over two lanes)
    for _ in range(6):
        emphases_amd.from_alignments_and_audios(
            alignments, audios, 16000, batch_size=batch_size, gpu=0)
    laps = []
    for _ in range(rounds):
        start = time.perf_counter()
        scores = emphases_amd.from_alignments_and_audios(
            alignments, audios, 16000, batch_size=batch_size, gpu=0)
        laps.append(time.perf_counter() - start)
    return float(np.median(laps)), float(
        sum(float(s.double().sum()) for s in scores))


def report(name, what, count, frames, words, seconds, extra=None):
    total = int(np.sum(frames))
    line = {
        'config': name, 'what': what, 'utterances': count,
        'frames': total, 'words': words,
        'audio_seconds': total / 100., 'ms': seconds * 1e3,
        'utterances_per_s': count / seconds,
        'frames_per_s': total / seconds,
        'hbm_frac_compulsory': total / seconds * BYTES_PER_FRAME / PEAK_HBM,
        'mfma_frac': (total * FLOPS_PER_FRAME + words * FLOPS_PER_WORD)
        / seconds / PEAK_MFMA,
        'realtime_factor': total / 100. / seconds}
    line.update(extra or {})
    print(json.dumps(line), flush=True)


def main():
    which = sys.argv[1:] or ['corpus', 'longform']
    device = torch.device('cuda', 0)
    engine = emphases_amd.engine.Engine(cfg.DEFAULT, None, device)
    host = [torch.from_numpy(synth.audio(7000 + i, 30000 if 'longform' in which
                                         else 3000)) for i in range(POOL)]
    on_device = [a.reshape(-1).to(device) for a in host]
    pcm = [torch.from_numpy(np.rint(a.numpy() * 32768.).astype(np.int16))
           for a in host]

    if 'corpus' in which:
        frames = synth.corpus_frames(10000, 200, 3000)
        alignments = [emphases_amd.Alignment.from_frames(
            synth.word_frames(5000 + i, int(n))) for i, n in enumerate(frames)]
        picks = np.arange(len(frames)) % POOL
        shards = dist.assign(dist.cost(frames), 8)
        loads = [int(frames[s].sum()) for s in shards]
        own = shards[0]
        sub = [alignments[i] for i in own]
        seconds, words, _, checksum = device_only(
            engine, on_device, picks[own], frames[own], sub, None)
        report('configs[3]', 'rank 0 of 8 (LPT share), device-only graph '
               'replay, audio resident', len(own), frames[own], words, seconds,
               {'shard_frames_min_max': [min(loads), max(loads)],
                'checksum': checksum})
        seconds, api_sum = through_api(host, picks[own], frames[own], sub, None)
        report('configs[3]', 'rank 0 of 8, public API (pageable float32 host '
               'tensors in, scores out)', len(own), frames[own], words, seconds,
               {'checksum': api_sum, 'pcie_floor_ms_at_55GBps':
                int(frames[own].sum()) * 640 / 55e9 * 1e3})
        seconds, api_sum = through_api(pcm, picks[own], frames[own], sub, None)
        report('configs[3]', 'rank 0 of 8, public API, 16-bit PCM tensors in',
               len(own), frames[own], words, seconds, {'checksum': api_sum})
        seconds, words, _, checksum = device_only(
            engine, on_device, picks, frames, alignments, None)
        report('configs[3]', 'whole corpus as ONE ragged batch on one GPU, '
               'device-only', len(frames), frames, words, seconds,
               {'checksum': checksum,
                'memory_allocated_GB': torch.cuda.max_memory_allocated() / 1e9})

    if 'longform' in which:
        for count in (8, 64):
            frames = np.full(count, 30000, dtype=np.int64)
            alignments = [emphases_amd.Alignment.from_frames(
                synth.word_frames(9000 + i, 30000)) for i in range(count)]
            picks = np.arange(count) % POOL
            seconds, words, _, checksum = device_only(
                engine, on_device, picks, frames, alignments, 3000)
            report('configs[4]', f'{count} x 5 min chunked at batch_size 3000, '
                   'device-only graph replay', count, frames, words, seconds,
                   {'checksum': checksum})
            seconds, api_sum = through_api(
                host, picks, frames, alignments, 3000)
            report('configs[4]', f'{count} x 5 min chunked at batch_size 3000, '
                   'public API (float32)', count, frames, words, seconds,
                   {'checksum': api_sum})
            seconds, api_sum = through_api(
                pcm, picks, frames, alignments, 3000)
            report('configs[4]', f'{count} x 5 min chunked at batch_size 3000, '
                   'public API, 16-bit PCM tensors in', count, frames, words,
                   seconds, {'checksum': api_sum})


if __name__ == '__main__':
    main()
