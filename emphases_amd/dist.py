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
3. `gather_scores` — one all_gather of the per-rank word counts and one
   all_gather of the padded score vectors (a few hundred KB in total: latency
   bound, a single collective each), after which every rank reorders the scores
   to input order.  The result is bitwise what one GPU would have produced,
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


def gather_scores(local_scores, local_indices, total, group=None):
    """All-gather per-utterance score vectors.

    local_scores: list of 1-D float32 tensors (this rank's utterances, in the
        order of `local_indices`), on the device of the process group's backend
    local_indices: indices of those utterances in the global input order
    total: number of utterances overall
    Returns a list of `total` 1-D tensors in input order (on every rank)."""
    dist = torch.distributed
    world = dist.get_world_size(group)
    device = local_scores[0].device if local_scores else torch.device('cpu')
    lengths = torch.tensor(
        [score.numel() for score in local_scores], dtype=torch.int64)
    header = torch.tensor(
        [len(local_scores), int(lengths.sum())], dtype=torch.int64,
        device=device)
    headers = [torch.zeros_like(header) for _ in range(world)]
    dist.all_gather(headers, header, group=group)
    headers = torch.stack(headers).cpu()
    max_count = int(headers[:, 0].max())
    max_words = int(headers[:, 1].max())

    # one integer collective (indices + lengths) and one float collective
    meta = torch.full((2, max(max_count, 1)), -1, dtype=torch.int64)
    meta[0, :len(local_scores)] = torch.as_tensor(
        np.asarray(local_indices, dtype=np.int64))
    meta[1, :len(local_scores)] = lengths
    meta = meta.to(device)
    payload = torch.zeros(max(max_words, 1), dtype=torch.float32, device=device)
    if local_scores:
        payload[:int(lengths.sum())] = torch.cat(
            [score.reshape(-1).to(torch.float32) for score in local_scores])
    metas = [torch.zeros_like(meta) for _ in range(world)]
    payloads = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    dist.all_gather(payloads, payload, group=group)

    result = [None] * total
    for rank in range(world):
        indices, sizes = metas[rank].cpu()
        cursor = 0
        for index, size in zip(indices.tolist(), sizes.tolist()):
            if index < 0:
                continue
            result[index] = payloads[rank][cursor:cursor + size]
            cursor += size
    if any(item is None for item in result):
        missing = [i for i, item in enumerate(result) if item is None]
        raise RuntimeError(f'no rank produced scores for utterances {missing}')
    return result


def from_alignments_and_audios(alignments, audios, sample_rate=16000,
                               checkpoint=None, batch_size=None, config=None,
                               compute=None, group=None):
    """Sharded version of `core.from_alignments_and_audios`: every rank passes
    the SAME full lists; each computes its LPT shard on its own GPU
    (`LOCAL_RANK`) and all ranks return all scores in input order.

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

        def compute(shard_alignments, shard_audios):
            return core.from_alignments_and_audios(
                shard_alignments, shard_audios, sample_rate, checkpoint,
                batch_size, torch.cuda.current_device(), config)
    local = compute([alignments[i] for i in mine], [audios[i] for i in mine]) \
        if len(mine) else []
    gathered = gather_scores(
        [score.reshape(-1) for score in local], mine, len(audios), group)
    return [score[None] for score in gathered]
