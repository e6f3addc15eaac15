"""CPU, world_size 2, gloo: the multi-GPU path's sharding and score gather.
The per-rank compute is the CPU oracle here (there is no GPU in this
container); on a GPU node the same code runs the HIP engine under nccl/RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT
from emphases_amd import dist as edist


def test_lpt_assignment_balances_and_covers():
    rng = np.random.default_rng(0)
    frames = rng.integers(200, 3000, size=1000)
    for world in (1, 2, 4, 8):
        shards = edist.assign(edist.cost(frames), world)
        covered = np.sort(np.concatenate(shards))
        assert np.array_equal(covered, np.arange(1000))
        loads = np.array([frames[s].sum() for s in shards], dtype=np.float64)
        assert loads.max() / loads.mean() < 1.01
    shards = edist.assign([5.0], 4)
    assert sum(len(s) for s in shards) == 1
    heavy = edist.cost([1000, 4000], 'transformer')
    assert heavy[1] / heavy[0] > 4


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, queue, frames=(300, 120, 450, 80, 200)):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import emphases_amd
        from emphases_amd import synth, weights
        from oracle import prominence as oracle
        state = {k: torch.from_numpy(v) for k, v in weights.load().items()}
        frames = list(frames)
        audios = [torch.from_numpy(synth.audio(i, n))
                  for i, n in enumerate(frames)]
        bounds = [synth.word_frames(i, n, 3, 40) for i, n in enumerate(frames)]
        aligns = [emphases_amd.Alignment.from_frames(b) for b in bounds]
        calls = []

        def compute(shard_alignments, shard_audios):
            calls.append(len(shard_audios))
            return [oracle.from_alignment_and_audio(
                [(w.start(), w.end()) for w in words], audio, state)
                for words, audio in zip(shard_alignments, shard_audios)]

        scores = edist.from_alignments_and_audios(
            aligns, audios, compute=compute)
        shard_calls = list(calls)
        reference = compute(aligns, audios)
        worst = max(
            float((a - b).abs().max()) for a, b in zip(scores, reference))
        shapes = [tuple(s.shape) for s in scores]
        queue.put((rank, shard_calls[0] if shard_calls else 0, worst, shapes,
                   [b.shape[1] for b in bounds]))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_equals_single_process():
    world = 2
    context = mp.get_context('spawn')
    queue = context.Queue()
    port = _free_port()
    workers = [context.Process(target=_worker, args=(r, world, port, queue))
               for r in range(world)]
    for worker in workers:
        worker.start()
    results = [queue.get(timeout=240) for _ in workers]
    for worker in workers:
        worker.join(timeout=60)
        assert worker.exitcode == 0
    shard_sizes = sorted(r[1] for r in results)
    assert sum(shard_sizes) == 5 and shard_sizes[0] >= 2
    for rank, _, worst, shapes, words in results:
        assert worst == 0.0            # no arithmetic crosses a rank boundary
        assert shapes == [(1, w) for w in words]


@pytest.mark.timeout(300)
def test_more_ranks_than_utterances():
    """LPT leaves ranks empty when world_size > utterances: the collective's
    device comes from the backend, not from the (absent) data."""
    world = 3
    context = mp.get_context('spawn')
    queue = context.Queue()
    port = _free_port()
    workers = [context.Process(
        target=_worker, args=(r, world, port, queue, (150, 90)))
        for r in range(world)]
    for worker in workers:
        worker.start()
    results = [queue.get(timeout=240) for _ in workers]
    for worker in workers:
        worker.join(timeout=60)
        assert worker.exitcode == 0
    for rank, _, worst, shapes, words in results:
        assert worst == 0.0
        assert shapes == [(1, w) for w in words]
