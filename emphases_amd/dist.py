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
3. every rank runs its shard through its own `Engine` / `Session`; the
   kernel variant of a layer follows from the configuration, not from the
   shard size (`Engine.frame_tile`);
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


class RankFailure(RuntimeError):
    """Raised on EVERY rank when some rank could not do its part: a rank that
    fails (an unreadable file, a corrupt alignment, a kernel error) still joins
    the collective with a sentinel payload instead of leaving the others
    blocked in it until the process group times out."""


def _raise_for(failed, rank, stage, failure):
    if not failed:
        return
    message = f'rank(s) {failed} failed {stage}'
    if rank in failed and failure is not None:
        raise RankFailure(f'{message}: {failure!r}') from failure
    raise RankFailure(message)


def exchange_counts(local_counts, shards, group=None, device=None,
                    failure=None):
    """Collective 1 of 2: how many scores each utterance of each rank will
    have -> int64 [world, widest] on the host (row r: rank r's utterances in
    the order of `shards[r]`, zero padded).  Called BEFORE anything is
    computed (the counts follow from the plan, `batch.score_counts`), so the
    device-to-host copy of its result waits for nothing and the score exchange
    that closes the run needs no size negotiation.

    `failure`: the exception that kept this rank from planning its shard; the
    rank then sends -1 counts and every rank raises `RankFailure`."""
    dist = torch.distributed
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = device or collective_device(group)
    if len(shards) != world:
        raise ValueError('shards do not describe this process group')
    if failure is None and len(local_counts) != len(shards[rank]):
        failure = ValueError(
            f'{len(local_counts)} counts for {len(shards[rank])} utterances')
    widest = max(max(len(shard) for shard in shards), 1)
    counts = torch.zeros(widest, dtype=torch.int64)
    if failure is None:
        counts[:len(local_counts)] = torch.as_tensor(
            np.asarray(local_counts, dtype=np.int64))
    else:
        counts[:] = -1
    counts = counts.to(device)
    all_counts = torch.empty(world * widest, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_counts, counts, group=group)
    all_counts = all_counts.cpu().reshape(world, widest)
    _raise_for([r for r in range(world) if int(all_counts[r].min()) < 0],
               rank, 'before the score-count exchange', failure)
    return all_counts


def score_order(all_counts, shards):
    """Where every score of the gathered payload goes: `(order, sizes)` with
    `flat = payloads[order]` the scores in input order and `sizes[i]` the
    number of scores of utterance i.  Host arithmetic on the counts of
    collective 1 and the assignment (numpy, once per job)."""
    counts = np.asarray(all_counts, dtype=np.int64)
    world = counts.shape[0]
    most = max(int(counts.sum(axis=1).max()), 1)
    total = sum(len(shard) for shard in shards)
    sizes = np.full(total, -1, dtype=np.int64)
    starts = np.zeros(total, dtype=np.int64)
    for source in range(world):
        shard = np.asarray(shards[source], dtype=np.int64)
        own = counts[source, :len(shard)]
        sizes[shard] = own
        starts[shard] = source * (most + 1) + np.cumsum(own) - own
    if (sizes < 0).any():
        missing = np.flatnonzero(sizes < 0).tolist()
        raise RuntimeError(f'no rank produced scores for utterances {missing}')
    offsets = np.cumsum(sizes) - sizes
    order = np.repeat(starts - offsets, sizes) + np.arange(int(sizes.sum()))
    return order, sizes, most


def exchange_scores(local_scores, all_counts, shards, group=None, device=None,
                    failure=None, flat=False):
    """Collective 2 of 2: every rank's scores back to back, padded to the
    largest rank's total (known to every rank from `all_counts`: no host
    synchronisation between the kernels and this collective) plus ONE status
    word, so that a rank whose compute failed (`failure`, or scores that do
    not have the planned sizes) still joins and every rank raises
    `RankFailure` afterwards instead of hanging.

    local_scores: list of 1-D tensors in the order of `shards[rank]`, or ONE
        1-D tensor holding them back to back (the packed word axis).
    Returns the scores in input order on `device` (every rank): a list of 1-D
    tensors, or with `flat=True` `(scores [total], sizes int64 [utterances])`
    - one gather on the device and no per-utterance work on the host."""
    dist = torch.distributed
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = device or collective_device(group)
    order, sizes, most = score_order(all_counts, shards)
    expected = np.asarray(all_counts[rank, :len(shards[rank])], dtype=np.int64)
    payload = torch.zeros(most + 1, dtype=torch.float32, device=device)
    if failure is None:
        try:
            if torch.is_tensor(local_scores):
                got = int(local_scores.numel())
                flat_local = local_scores.reshape(-1)
            else:
                got = [int(score.numel()) for score in local_scores]
                flat_local = torch.cat(
                    [score.reshape(-1).to(torch.float32)
                     for score in local_scores]) if local_scores else None
            if (got != int(expected.sum())) if torch.is_tensor(local_scores) \
                    else (got != expected.tolist()):
                raise RuntimeError(
                    'scores do not have the planned sizes: '
                    f'{got} vs {expected.tolist()}')
            if flat_local is not None and flat_local.numel():
                payload[:flat_local.numel()] = flat_local.to(
                    device=device, dtype=torch.float32)
        except Exception as error:       # noqa: BLE001
            failure = error
    if failure is not None:
        payload[most] = 1.
    payloads = torch.empty(
        world * (most + 1), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(payloads, payload, group=group)
    status = payloads[most::most + 1].cpu()
    _raise_for([r for r in range(world) if float(status[r]) != 0.],
               rank, 'to compute its shard', failure)
    scores = payloads[torch.as_tensor(order, device=device)]
    if flat:
        return scores, torch.as_tensor(sizes)
    return list(torch.split(scores, sizes.tolist()))


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


# Conv tile of a sharded run.  Since round 4 the engine's default
# (`conv_tile=None`) fixes the kernel family of every frame-rate layer by the
# configuration, never by the batch (`Engine.frame_tile`), so a shard of a
# handful of utterances runs the kernels a 10 000-utterance batch runs and the
# scores do not depend on the world size; sharded calls simply use that
# default.  (Round 3 had to pin 64 here because the default picked by shard
# size.)
CONV_TILE = None


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


def bind_device(group=None, compute=None):
    """Make this rank's GPU the current device BEFORE its first collective or
    engine call: `nccl` (= RCCL) takes collective tensors on the current
    device, so a rank that has not bound yet would put them on `cuda:0` next
    to every other rank's ("duplicate GPU").  Also for a rank whose shard is
    empty.  With an injected `compute` under a host backend (the gloo CPU
    tests) there is no GPU to bind."""
    backend = str(torch.distributed.get_backend(group)).lower()
    if 'nccl' in backend or compute is None:
        return local_device()
    return None


def from_alignments_and_audios(alignments, audios, sample_rate=16000,
                               checkpoint=None, batch_size=None, config=None,
                               compute=None, group=None, conv_tile=CONV_TILE,
                               precision='f32'):
    """Sharded version of `core.from_alignments_and_audios`: every rank passes
    the SAME full lists (tensors already in memory; for a corpus on disk use
    `from_files_to_files`, which loads only the shard); each computes its LPT
    shard on its own GPU (`LOCAL_RANK`, bound here with
    `torch.cuda.set_device` before the first collective) and all ranks return
    all scores in input order (on the collective's device: the GPU for
    nccl/RCCL, the CPU for gloo).  The kernel family does not depend on the
    batch (`Engine.frame_tile`), so the result is bitwise the same for every
    world size, 1 included - at every `precision` ('f32', or an opt-in name
    of `engine.PRECISIONS`).  A rank that
    fails still joins both collectives and every rank raises `RankFailure`.

    `compute(alignments, audios) -> list of [1, W] tensors` can replace the
    HIP engine (the gloo CPU test injects the oracle there)."""
    from . import batch
    dist = torch.distributed
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    gpu = bind_device(group, compute)
    architecture = getattr(config, 'architecture', 'convolution')
    frames = [frames_at_16k(audio.shape[-1], sample_rate) for audio in audios]
    shards = assign(cost(frames, architecture), world)
    mine = shards[rank]
    if compute is None:
        from . import core

        def compute(shard_alignments, shard_audios):
            return core.from_alignments_and_audios(
                shard_alignments, shard_audios, sample_rate, checkpoint,
                batch_size, gpu, config, conv_tile=conv_tile,
                precision=precision)
    counts, failure = [], None
    try:
        lengths = [length_at_16k(audios[i].shape[-1], sample_rate)
                   for i in mine]
        counts = batch.score_counts(
            [alignments[i] for i in mine], lengths, batch_size)
    except Exception as error:       # noqa: BLE001
        failure = error
    all_counts = exchange_counts(counts, shards, group, failure=failure)
    local = []
    try:
        if len(mine):
            local = compute(
                [alignments[i] for i in mine], [audios[i] for i in mine])
    except Exception as error:       # noqa: BLE001
        failure = error
    gathered = exchange_scores(
        [score.reshape(-1) for score in local], all_counts, shards, group,
        failure=failure)
    return [score[None] for score in gathered]


def from_files_to_files(text_files, audio_files, output_prefixes=None,
                        checkpoint=None, batch_size=None, config=None,
                        group=None, utterances_per_batch=256,
                        conv_tile=CONV_TILE, gather=True, compute=None,
                        precision='f32'):
    """`emphases.from_files_to_files` (`core.py:115-179`) over the ranks of a
    process group, one process per GPU.  No rank reads what it does not
    compute:

    0. the rank's GPU (`LOCAL_RANK`) becomes the current device - before the
       first collective, and for a rank with an empty shard too;
    1. every rank reads the WAV *headers* of all files (`load.wav_info`:
       seeks, no samples) and computes the same LPT assignment by frames at
       16 kHz;
    2. `gather=True`: it reads the alignments of ITS shard, plans them
       (`batch.score_counts`) and joins collective 1 (`exchange_counts`);
    3. it loads, stages and runs its shard through its own `Session` in
       batches of `utterances_per_batch` with two batches in flight
       (`core.files_to_scores`) and writes `<prefix>.TextGrid` / `<prefix>.pt`
       for its own files as the reference does (`core.py:111-112`);
    4. `gather=True`: collective 2 (`exchange_scores`) returns all scores, in
       input order, to every rank; `gather=False` (the command line): returns
       this rank's {index: scores} and NO collective runs at all - nothing
       would consume the counts.

    A rank that fails in step 2 or 3 still joins the collectives (sentinel
    payload) and every rank raises `RankFailure`: nobody is left blocked in an
    all_gather.  The files a rank writes are bitwise those a single process
    writes (`core.from_files_to_files`, the single-process command line): the
    kernel family does not depend on the batch (`Engine.frame_tile`).  `compute(text_files, audio_files, deliver)` replaces
    step 3's engine (the gloo CPU test passes the oracle)."""
    from pathlib import Path
    from . import alignment as alignment_module
    from . import batch
    from . import load
    dist = torch.distributed
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    gpu = bind_device(group, compute)
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
    failure, all_counts = None, None
    if gather:
        counts = []
        try:
            lengths = [length_at_16k(headers[i][2], headers[i][0])
                       for i in mine]
            # (the library's parser, a batch of files per call - the one that
            # produces the scores' alignments below, so the counts and the
            # scores cannot disagree; a Python reader per file was 0.5 ms each,
            # in front of the first collective)
            from . import files
            counts = []
            for lo in range(0, len(own_text), 1024):
                opened = files.FileBatch(
                    own_text[lo:lo + 1024], own_audio[lo:lo + 1024])
                counts.extend(batch.score_counts(
                    opened.all_times(), lengths[lo:lo + 1024],
                    batch_size).tolist())
                opened.close()
        except Exception as error:       # noqa: BLE001
            failure = error
        all_counts = exchange_counts(counts, shards, group, failure=failure)
    local = {}

    def deliver(index, item, scores):
        from . import core
        core._save(item, scores, output_prefixes[mine[index]])
        local[mine[index]] = scores

    try:
        if compute is not None:
            compute(own_text, own_audio, deliver)
        elif mine:
            from . import core
            session = core.get_session(
                checkpoint, gpu, config, conv_tile, precision)
            def deliver_batch(opened, chosen, indices, scores):
                opened.write(
                    chosen, [output_prefixes[mine[i]] for i in indices],
                    scores)
                for index, item in zip(indices, scores):
                    local[mine[index]] = item
            core.files_to_scores(
                own_text, own_audio, session, batch_size, utterances_per_batch,
                deliver_batch=deliver_batch)
    except Exception as error:       # noqa: BLE001
        if not gather:
            raise
        failure = error
    if not gather:
        return local
    gathered = exchange_scores(
        [local[i].reshape(-1) for i in mine if i in local], all_counts,
        shards, group, failure=failure)
    return [score[None] for score in gathered]
