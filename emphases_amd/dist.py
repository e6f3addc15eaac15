"""Multi-GPU execution: utterances shard embarrassingly, scores are gathered.

The reference has no distributed code at all; every utterance
— and every `batch_size` chunk of a long one — is computed independently
(`emphases/core.py:250-265`), so the only exchange a multi-GPU run needs is the
gather of per-word scores at the end.  One process per GPU
(`torch.distributed`, backend `nccl` = RCCL over xGMI on ROCm; `gloo` on CPU
for tests):

1. `assign` — longest-processing-time-first sharding of utterances by cost
   (frames for the conv model, frames^2-ish for the transformer), so that ranks
   finish together; planned from lengths alone (`from_files_to_files`: from the
   WAV headers — a rank never reads audio it does not compute);
2. `exchange_counts` — one all_gather of the per-utterance score counts, which
   follow from the plan (`batch.score_counts`), BEFORE anything is computed;
3. every rank runs its shard through its own `Engine` / `Session`, with the
   conv tile pinned (`CONV_TILE`) so that the kernel variant does not depend on
   the shard size;
4. `exchange_scores` — one all_gather of the padded score vectors (a few
   hundred KB in total: latency bound; which utterance sits where follows from
   the assignment every rank computed and the counts of step 2, so nothing is
   negotiated and no device-to-host copy sits between the kernels and the
   collective), after which every rank reorders the scores to input order.
   The result is bitwise what one GPU produces, because no arithmetic crosses
   a rank boundary.
"""
import numpy as np
import torch


def cost(frames, architecture='convolution'):
    """Relative cost of an utterance."""
    frames = np.asarray(frames, dtype=np.float64)
    if architecture == 'transformer':
        return frames * (1.0 + frames / 2000.0)    # attention grows ~ F^2
    return frames


def assign(costs, world_size):
    """Longest-processing-time-first assignment.

    Returns a list of `world_size` index arrays (ascending within a shard)."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind='stable')
    loads = np.zeros(world_size)
    shards = [[] for _ in range(world_size)]
    for index in order:
        rank = int(np.argmin(loads))
        shards[rank].append(int(index))
        loads[rank] += costs[index]
    return [np.array(sorted(shard), dtype=np.int64) for shard in shards]


def collective_device(group=None):
    """Device the process group's collectives take their tensors on: the
    rank's own GPU for nccl (= RCCL), the CPU for gloo.  Derived from the
    backend, never from the data (an empty shard has no tensor to ask)."""
    backend = str(torch.distributed.get_backend(group)).lower()
    if 'nccl' in backend:
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def exchange_counts(local_counts, shards, group=None, device=None):
    """Collective 1 of 2: how many scores each utterance of each rank will
    have -> int64 [world, widest] on the host (row r: rank r's utterances in
    the order of `shards[r]`, zero padded).  Called BEFORE anything is
    computed (the counts follow from the plan, `batch.score_counts`), so the
    device-to-host copy of its result waits for nothing and the score exchange
    that closes the run needs no size negotiation."""
    dist = torch.distributed
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = device or collective_device(group)
    if len(shards) != world or len(local_counts) != len(shards[rank]):
        raise ValueError('shards do not describe this process group')
    widest = max(max(len(shard) for shard in shards), 1)
    counts = torch.zeros(widest, dtype=torch.int64)
    counts[:len(local_counts)] = torch.as_tensor(
        np.asarray(local_counts, dtype=np.int64))
    counts = counts.to(device)
    all_counts = torch.empty(world * widest, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_counts, counts, group=group)
    return all_counts.cpu().reshape(world, widest)


def exchange_scores(local_scores, all_counts, shards, group=None, device=None):
    """Collective 2 of 2: every rank's scores back to back, padded to the
    largest rank's total (known to every rank from `all_counts`: no host
    synchronisation between the kernels and this collective).  Returns a list
    of 1-D tensors in input order, on `device` (every rank)."""
    dist = torch.distributed
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = device or collective_device(group)
    expected = all_counts[rank, :len(shards[rank])].tolist()
    if [int(score.numel()) for score in local_scores] != expected:
        raise RuntimeError(
            'scores do not have the planned sizes: '
            f'{[int(s.numel()) for s in local_scores]} vs {expected}')
    most = max(int(all_counts.sum(dim=1).max()), 1)
    payload = torch.zeros(most, dtype=torch.float32, device=device)
    if local_scores:
        flat = torch.cat(
            [score.reshape(-1).to(torch.float32) for score in local_scores])
        payload[:flat.numel()] = flat.to(device)
    payloads = torch.empty(world * most, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(payloads, payload, group=group)

    total = sum(len(shard) for shard in shards)
    result = [None] * total
    for source in range(world):
        cursor = source * most
        for index, size in zip(
                shards[source], all_counts[source].tolist()):
            result[int(index)] = payloads[cursor:cursor + size]
            cursor += size
    if any(item is None for item in result):
        missing = [i for i, item in enumerate(result) if item is None]
        raise RuntimeError(f'no rank produced scores for utterances {missing}')
    return result


def gather_scores(local_scores, shards, group=None, device=None):
    """Both collectives for scores that already exist: the
    counts are taken from the scores themselves.

    local_scores: list of 1-D float32 tensors, this rank's utterances in the
        order of `shards[rank]`
    shards: the LPT assignment every rank computed from the same inputs
        (`assign`): `shards[r]` = global indices of rank r's utterances
    Returns a list of 1-D tensors in input order, on `device` (every rank)."""
    all_counts = exchange_counts(
        [score.numel() for score in local_scores], shards, group, device)
    return exchange_scores(local_scores, all_counts, shards, group, device)


def local_device():
    """This rank's GPU: `LOCAL_RANK` (torchrun) modulo the visible devices,
    made the current device before any engine or collective call."""
    import os
    count = torch.cuda.device_count()
    if count < 1:
        from . import runtime
        runtime.require_gpu()          # raises: no CPU fallback
    rank = torch.distributed.get_rank() \
        if torch.distributed.is_initialized() else 0
    index = int(os.environ.get('LOCAL_RANK', rank)) % count
    torch.cuda.set_device(index)
    return index


# Positions per frame-rate conv tile of a sharded run.  `Engine.frame_tile`
# would pick by shard size (a shard of a handful of utterances takes
# 16-position direct-form tiles instead of the F(4,3) kernel's 64), and scores
# of different kernels agree to 1e-6, not bitwise: a sharded job pins the tile
# so that its scores do not depend on the world size.
CONV_TILE = 64


def length_at_16k(samples, sample_rate):
    """Samples an utterance has once it is at 16 kHz (`core.py:613-619`)."""
    from . import config as cfg
    from . import load
    if int(sample_rate) == cfg.SAMPLE_RATE:
        return int(samples)
    _, orig, new, _ = load.resample_kernel(sample_rate)
    return load.resampled_length(int(samples), orig, new)


def frames_at_16k(samples, sample_rate):
    from . import config as cfg
    return length_at_16k(samples, sample_rate) // cfg.HOPSIZE


def from_alignments_and_audios(alignments, audios, sample_rate=16000,
                               checkpoint=None, batch_size=None, config=None,
                               compute=None, group=None, conv_tile=CONV_TILE):
    """Sharded version of `core.from_alignments_and_audios`: every rank passes
    the SAME full lists (tensors already in memory; for a corpus on disk use
    `from_files_to_files`, which loads only the shard); each computes its LPT
    shard on its own GPU (`LOCAL_RANK`, bound here with
    `torch.cuda.set_device`) and all ranks return all scores in input order
    (on the collective's device: the GPU for nccl/RCCL, the CPU for gloo).
    `conv_tile` is pinned (`CONV_TILE`), so the result is bitwise the same for
    every world size, 1 included.

    `compute(alignments, audios) -> list of [1, W] tensors` can replace the
    HIP engine (the gloo CPU test injects the oracle there)."""
    from . import batch
    dist = torch.distributed
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    architecture = getattr(config, 'architecture', 'convolution')
    frames = [frames_at_16k(audio.shape[-1], sample_rate) for audio in audios]
    shards = assign(cost(frames, architecture), world)
    mine = shards[rank]
    if compute is None:
        from . import core
        gpu = local_device()

        def compute(shard_alignments, shard_audios):
            return core.from_alignments_and_audios(
                shard_alignments, shard_audios, sample_rate, checkpoint,
                batch_size, gpu, config, conv_tile=conv_tile)
    lengths = [length_at_16k(audios[i].shape[-1], sample_rate) for i in mine]
    all_counts = exchange_counts(
        batch.score_counts([alignments[i] for i in mine], lengths, batch_size),
        shards, group)
    local = compute([alignments[i] for i in mine], [audios[i] for i in mine]) \
        if len(mine) else []
    gathered = exchange_scores(
        [score.reshape(-1) for score in local], all_counts, shards, group)
    return [score[None] for score in gathered]


def from_files_to_files(text_files, audio_files, output_prefixes=None,
                        checkpoint=None, batch_size=None, config=None,
                        group=None, utterances_per_batch=64,
                        conv_tile=CONV_TILE, gather=True, compute=None):
    """`emphases.from_files_to_files` (`core.py:115-179`) over the ranks of a
    process group, one process per GPU.  No rank reads what it does not
    compute:

    1. every rank reads the WAV *headers* of all files (`load.wav_info`:
       seeks, no samples) and computes the same LPT assignment by frames at
       16 kHz;
    2. it reads the alignments of ITS shard, plans them
       (`batch.score_counts`) and joins collective 1 (`exchange_counts`);
    3. it loads, stages and runs its shard through its own `Session` in
       batches of `utterances_per_batch` with two batches in flight
       (`core.files_to_scores`) and writes `<prefix>.TextGrid` / `<prefix>.pt`
       for its own files as the reference does (`core.py:111-112`);
    4. `gather=True`: collective 2 (`exchange_scores`) returns all scores, in
       input order, to every rank; `gather=False`: returns this rank's
       {index: scores} and no second collective runs.

    The conv tile is pinned (`CONV_TILE`): the files a rank writes are bitwise
    those a single process writes.  `compute(text_files, audio_files,
    deliver)` replaces step 3's engine (the gloo CPU test passes the
    oracle)."""
    from pathlib import Path
    from . import alignment as alignment_module
    from . import batch
    from . import load
    dist = torch.distributed
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    text_files, audio_files = list(text_files), list(audio_files)
    if len(text_files) != len(audio_files):
        raise ValueError('one audio file per text file')
    for file in text_files:
        if not str(file).endswith(('.TextGrid', '.json')):
            from . import core
            core.from_text_and_audio(None, None, None)   # raises: no forced alignment
    if output_prefixes is None:
        output_prefixes = [Path(file).stem for file in text_files]
    output_prefixes = list(output_prefixes)
    architecture = getattr(config, 'architecture', 'convolution')
    headers = [load.wav_info(file) for file in audio_files]
    frames = [frames_at_16k(samples, rate) for rate, _, samples in headers]
    shards = assign(cost(frames, architecture), world)
    mine = [int(i) for i in shards[rank]]
    own_text = [text_files[i] for i in mine]
    own_audio = [audio_files[i] for i in mine]
    lengths = [length_at_16k(headers[i][2], headers[i][0]) for i in mine]
    all_counts = exchange_counts(
        batch.score_counts(
            [alignment_module.Alignment(file) for file in own_text], lengths,
            batch_size), shards, group)
    local = {}

    def deliver(index, item, scores):
        from . import core
        core._save(item, scores, output_prefixes[mine[index]])
        local[mine[index]] = scores

    if compute is not None:
        compute(own_text, own_audio, deliver)
    elif mine:
        from . import core
        gpu = local_device()
        session = core.get_session(checkpoint, gpu, config, conv_tile)
        core.files_to_scores(
            own_text, own_audio, session, batch_size, utterances_per_batch,
            deliver)
    if not gather:
        return local
    gathered = exchange_scores(
        [local[i].reshape(-1) for i in mine], all_counts, shards, group)
    return [score[None] for score in gathered]
