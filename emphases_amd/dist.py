"""Multi-GPU execution: utterances shard embarrassingly, scores are gathered.

The reference has no distributed code at all; every utterance
— and every `batch_size` chunk of a long one — is computed independently
(`emphases/core.py:250-265`), so the only exchange a multi-GPU run needs is the
gather of per-word scores at the end.  One process per GPU
(`torch.distributed`, backend `nccl` = RCCL over xGMI on ROCm; `gloo` on CPU
for tests):

1. `assign` — longest-processing-time-first sharding of utterances by cost
   (frames for the conv model, frames^2-ish for the transformer), so that ranks
   finish together;
2. every rank runs its shard through its own `Engine`;
3. `gather_scores` — one all_gather of the per-utterance word counts and one
   all_gather of the padded score vectors (a few hundred KB in total: latency
   bound, a single collective each; which utterance sits where follows from
   the assignment every rank computed), after which every rank reorders the
   scores to input order.  The result is bitwise what one GPU would have produced,
   because no arithmetic crosses a rank boundary.
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


def gather_scores(local_scores, shards, group=None, device=None):
    """All-gather per-utterance score vectors: one collective of counts and one
    of scores.

    local_scores: list of 1-D float32 tensors, this rank's utterances in the
        order of `shards[rank]`
    shards: the LPT assignment every rank computed from the same inputs
        (`assign`): `shards[r]` = global indices of rank r's utterances
    Returns a list of 1-D tensors in input order, on `device` (every rank)."""
    dist = torch.distributed
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = device or collective_device(group)
    if len(shards) != world or len(local_scores) != len(shards[rank]):
        raise ValueError('shards do not describe this process group')
    widest = max(max(len(shard) for shard in shards), 1)

    # (1) counts: words of each of this rank's utterances, padded to the
    # largest shard (known to every rank from the assignment)
    counts = torch.zeros(widest, dtype=torch.int64)
    counts[:len(local_scores)] = torch.tensor(
        [score.numel() for score in local_scores], dtype=torch.int64)
    counts = counts.to(device)
    all_counts = torch.empty(world * widest, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_counts, counts, group=group)
    all_counts = all_counts.cpu().reshape(world, widest)

    # (2) scores: every rank's words back to back, padded to the largest
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


def from_alignments_and_audios(alignments, audios, sample_rate=16000,
                               checkpoint=None, batch_size=None, config=None,
                               compute=None, group=None):
    """Sharded version of `core.from_alignments_and_audios`: every rank passes
    the SAME full lists; each computes its LPT shard on its own GPU
    (`LOCAL_RANK`, bound here with `torch.cuda.set_device`) and all ranks
    return all scores in input order (on the collective's device: the GPU for
    nccl/RCCL, the CPU for gloo).

    `compute(alignments, audios) -> list of [1, W] tensors` can replace the
    HIP engine (the gloo CPU test injects the oracle there)."""
    dist = torch.distributed
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    architecture = getattr(config, 'architecture', 'convolution')
    frames = [int(audio.shape[-1]) // 160 for audio in audios]
    shards = assign(cost(frames, architecture), world)
    mine = shards[rank]
    if compute is None:
        from . import core
        gpu = local_device()

        def compute(shard_alignments, shard_audios):
            return core.from_alignments_and_audios(
                shard_alignments, shard_audios, sample_rate, checkpoint,
                batch_size, gpu, config)
    local = compute([alignments[i] for i in mine], [audios[i] for i in mine]) \
        if len(mine) else []
    gathered = gather_scores(
        [score.reshape(-1) for score in local], shards, group)
    return [score[None] for score in gathered]
