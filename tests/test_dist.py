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


###############################################################################
# Sharded file API (dist.from_files_to_files)
###############################################################################


FILE_FRAMES = (300, 120, 450, 80, 200, 260, 90)
FILE_RATES = (16000, 16000, 8000, 16000, 16000, 22050, 16000)


def _write_corpus(directory):
    """Seeded WAV + TextGrid files; two of them not at 16 kHz."""
    import emphases_amd
    from emphases_amd import load, synth
    texts, audios = [], []
    for index, (frames, rate) in enumerate(zip(FILE_FRAMES, FILE_RATES)):
        samples = frames * rate // 100
        audio = synth.weights(900 + index, (1, samples), 0.4)
        load.save_wav(directory / f'u{index}.wav', audio, rate)
        emphases_amd.Alignment.from_frames(
            synth.word_frames(index, frames, 3, 40)).save(
                directory / f'u{index}.TextGrid')
        texts.append(directory / f'u{index}.TextGrid')
        audios.append(directory / f'u{index}.wav')
    return texts, audios


def _oracle_file(text_file, audio_file, state):
    import emphases_amd
    from emphases_amd import load
    from oracle import prominence as oracle
    from oracle import resample as oracle_resample
    alignment = emphases_amd.Alignment(text_file)
    audio, rate = load.wav(audio_file)
    if rate != 16000:
        audio = torch.from_numpy(oracle_resample.resample(
            audio[0].numpy(), rate).astype(np.float32))[None]
    return alignment, oracle.from_alignment_and_audio(
        [(w.start(), w.end()) for w in alignment], audio, state)


def _file_worker(rank, world, port, queue, directory):
    from pathlib import Path
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from emphases_amd import load, weights
        directory = Path(directory)
        texts = [directory / f'u{i}.TextGrid' for i in range(len(FILE_FRAMES))]
        audios = [directory / f'u{i}.wav' for i in range(len(FILE_FRAMES))]
        prefixes = [directory / f'out_{world}_{i}'
                    for i in range(len(FILE_FRAMES))]
        state = {k: torch.from_numpy(v) for k, v in weights.load().items()}
        read = []
        original = load.wav

        def tracking_wav(file, raw=False):
            read.append(Path(file).name)
            return original(file, raw)
        load.wav = tracking_wav

        def compute(own_text, own_audio, deliver):
            for index, (text, audio) in enumerate(zip(own_text, own_audio)):
                alignment, scores = _oracle_file(text, audio, state)
                deliver(index, alignment, scores)

        scores = edist.from_files_to_files(
            texts, audios, prefixes, compute=compute)
        # (numpy: pickled by value; a torch tensor travels as a shared-memory
        # handle that dies with this process)
        queue.put((rank, sorted(read), [s.numpy().copy() for s in scores]))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_files_read_only_their_shard(tmp_path):
    """dist.from_files_to_files under 2 gloo ranks (the oracle as compute):
    planned from WAV headers alone, every rank reads the samples of its own
    shard only, writes its own outputs, and both ranks get all scores - the
    ones a single process computes."""
    from emphases_amd import load, weights
    texts, audios = _write_corpus(tmp_path)
    for file, frames, rate in zip(audios, FILE_FRAMES, FILE_RATES):
        assert load.wav_info(file) == (rate, 1, frames * rate // 100)
    world = 2
    context = mp.get_context('spawn')
    queue = context.Queue()
    port = _free_port()
    workers = [context.Process(
        target=_file_worker, args=(r, world, port, queue, str(tmp_path)))
        for r in range(world)]
    for worker in workers:
        worker.start()
    results = sorted([queue.get(timeout=240) for _ in workers],
                     key=lambda item: item[0])
    for worker in workers:
        worker.join(timeout=60)
        assert worker.exitcode == 0
    # the shards: planned by frames at 16 kHz, disjoint, covering
    frames = [edist.frames_at_16k(f * r // 100, r)
              for f, r in zip(FILE_FRAMES, FILE_RATES)]
    shards = edist.assign(edist.cost(frames), world)
    for (rank, read, _), shard in zip(results, shards):
        assert read == sorted(f'u{i}.wav' for i in shard)
    state = {k: torch.from_numpy(v) for k, v in weights.load().items()}
    torch.set_num_threads(1)       # the workers' setting: same summation order
    for index, (text, audio) in enumerate(zip(texts, audios)):
        alignment, want = _oracle_file(text, audio, state)
        for _, _, scores in results:
            assert scores[index].shape == (1, len(alignment))
            assert np.array_equal(scores[index], want.numpy())
        saved = torch.load(tmp_path / f'out_{world}_{index}.pt')
        assert torch.equal(saved, want)
        assert (tmp_path / f'out_{world}_{index}.TextGrid').exists()


def test_score_counts_follow_the_plan():
    """batch.score_counts: words minus those of dropped chunks
    (core.py:414-415), per utterance."""
    import emphases_amd
    from emphases_amd import batch
    bounds = np.array([[0, 40, 42, 90], [40, 42, 90, 130]])
    aligns = [emphases_amd.Alignment.from_frames(bounds),
              emphases_amd.Alignment.from_frames(bounds)]
    lengths = [130 * 160, 130 * 160]
    assert batch.score_counts(aligns, lengths).tolist() == [4, 4]
    # batch_size 0: every word its own chunk; the 2-frame word is shorter than
    # the reflect pad and is dropped (tests/golden/chunks.npz dropped_chunk_b0)
    counts = batch.score_counts(aligns, lengths, 0)
    plan = batch.plan_batch(aligns, lengths, 0)
    assert counts.tolist() == [int(plan.words[plan.utterance == u].sum())
                               for u in range(2)]
    assert counts.tolist() == [3, 3]


###############################################################################
# Device binding and failure propagation
###############################################################################


def _single_rank_group(tmp_path):
    torch.distributed.init_process_group(
        'gloo', rank=0, world_size=1,
        init_method=f'file://{tmp_path}/rendezvous')


def test_device_is_bound_before_the_first_collective(tmp_path, monkeypatch):
    """Under nccl (= RCCL) a collective takes its tensors on the CURRENT
    device, so both sharded entry points must make the rank's GPU current
    before collective 1 - also for a rank whose shard is empty, also when a
    `compute` is injected.  Recorded here with the backend reported as nccl,
    `local_device` and the collectives' device patched (no GPU in this
    container)."""
    import emphases_amd
    from emphases_amd import load, synth
    _single_rank_group(tmp_path)
    try:
        events = []
        monkeypatch.setattr(
            torch.distributed, 'get_backend', lambda group=None: 'nccl')
        monkeypatch.setattr(
            edist, 'local_device', lambda: events.append('bind') or 0)

        def device(group=None):
            events.append('collective')
            return torch.device('cpu')
        monkeypatch.setattr(edist, 'collective_device', device)

        frames = [120, 90]
        audios = [torch.from_numpy(synth.audio(i, n))
                  for i, n in enumerate(frames)]
        aligns = [emphases_amd.Alignment.from_frames(
            synth.word_frames(i, n, 3, 40)) for i, n in enumerate(frames)]

        def compute(shard_alignments, shard_audios):
            events.append('compute')
            return [torch.zeros(1, len(a)) for a in shard_alignments]
        edist.from_alignments_and_audios(aligns, audios, compute=compute)
        assert events[0] == 'bind' and events.count('bind') == 1
        assert events.index('bind') < events.index('collective') \
            < events.index('compute')

        # the file API: bound first, with files ...
        texts, waves = [], []
        for index, (audio, alignment) in enumerate(zip(audios, aligns)):
            load.save_wav(tmp_path / f'b{index}.wav', audio.numpy(), 16000)
            alignment.save(tmp_path / f'b{index}.TextGrid')
            texts.append(tmp_path / f'b{index}.TextGrid')
            waves.append(tmp_path / f'b{index}.wav')

        def file_compute(own_text, own_audio, deliver):
            events.append('compute')
            for index, text in enumerate(own_text):
                alignment = emphases_amd.Alignment(text)
                deliver(index, alignment, torch.zeros(1, len(alignment)))
        del events[:]
        edist.from_files_to_files(
            texts, waves, [tmp_path / f'o{i}' for i in range(2)],
            compute=file_compute)
        assert events[0] == 'bind' and events.count('bind') == 1
        assert 'collective' in events
        # ... and with an EMPTY shard (no files at all for this rank)
        del events[:]
        assert edist.from_files_to_files([], [], [], compute=file_compute) == []
        assert events[0] == 'bind'
        # gather=False (the command line): no collective at all
        del events[:]
        local = edist.from_files_to_files(
            texts, waves, [tmp_path / f'p{i}' for i in range(2)],
            compute=file_compute, gather=False)
        assert sorted(local) == [0, 1] and 'collective' not in events
        assert events[0] == 'bind'
    finally:
        torch.distributed.destroy_process_group()


def _failing_worker(rank, world, port, queue, directory, stage):
    """Rank 1 fails (`stage` 'plan': a corrupt alignment file; 'compute': its
    engine raises); rank 0 must get RankFailure instead of hanging."""
    from pathlib import Path
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import emphases_amd
        directory = Path(directory)
        count = len(FILE_FRAMES)
        texts = [directory / f'u{i}.TextGrid' for i in range(count)]
        audios = [directory / f'u{i}.wav' for i in range(count)]
        prefixes = [directory / f'fail_{stage}_{i}' for i in range(count)]

        def compute(own_text, own_audio, deliver):
            if rank == 1 and stage == 'compute':
                raise OSError('the engine of rank 1 broke')
            for index, text in enumerate(own_text):
                alignment = emphases_amd.Alignment(text)
                deliver(index, alignment, torch.zeros(1, len(alignment)))
        try:
            edist.from_files_to_files(texts, audios, prefixes, compute=compute)
            queue.put((rank, 'returned', ''))
        except edist.RankFailure as error:
            queue.put((rank, 'RankFailure', str(error)))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('stage', ['plan', 'compute'])
def test_a_failing_rank_does_not_hang_the_others(tmp_path, stage):
    texts, audios = _write_corpus(tmp_path)
    world = 2
    if stage == 'plan':
        # corrupt one alignment of rank 1's shard
        frames = [edist.frames_at_16k(f * r // 100, r)
                  for f, r in zip(FILE_FRAMES, FILE_RATES)]
        victim = int(edist.assign(edist.cost(frames), world)[1][0])
        texts[victim].write_text('not a TextGrid')
    context = mp.get_context('spawn')
    queue = context.Queue()
    port = _free_port()
    workers = [context.Process(
        target=_failing_worker,
        args=(r, world, port, queue, str(tmp_path), stage))
        for r in range(world)]
    for worker in workers:
        worker.start()
    results = sorted(queue.get(timeout=120) for _ in workers)
    for worker in workers:
        worker.join(timeout=60)
        assert worker.exitcode == 0
    assert [r[1] for r in results] == ['RankFailure', 'RankFailure']
    assert all('[1]' in r[2] for r in results)
    # the failing rank's error names the cause, the other's only the rank
    assert ('broke' in results[1][2]) == (stage == 'compute')
    assert 'broke' not in results[0][2]


def test_flat_score_exchange(tmp_path):
    """exchange_scores(flat=True): one tensor in input order + sizes, from a
    packed local tensor (what the strong-scaling bench hands over)."""
    _single_rank_group(tmp_path)
    try:
        shards = [np.array([0, 1, 2])]
        counts = edist.exchange_counts([2, 0, 3], shards)
        local = torch.arange(5, dtype=torch.float32)
        flat, sizes = edist.exchange_scores(local, counts, shards, flat=True)
        assert sizes.tolist() == [2, 0, 3] and torch.equal(flat, local)
        listed = edist.exchange_scores(
            [local[:2], local[2:2], local[2:]], counts, shards)
        assert [s.tolist() for s in listed] == [[0., 1.], [], [2., 3., 4.]]
        with pytest.raises(edist.RankFailure):
            edist.exchange_scores(local[:4], counts, shards)
    finally:
        torch.distributed.destroy_process_group()


def test_score_order_interleaves_ranks():
    """score_order: rank payloads are (most + 1) apart (scores + status word);
    utterances come back in input order whatever the assignment."""
    shards = [np.array([1, 3]), np.array([0, 2, 4])]
    counts = np.array([[2, 1, 0], [3, 0, 2]])
    order, sizes, most = edist.score_order(counts, shards)
    assert most == 5 and sizes.tolist() == [3, 2, 0, 1, 2]
    # rank 0's payload starts at 0, rank 1's at most + 1 = 6
    assert order.tolist() == [6, 7, 8, 0, 1, 2, 9, 10]
