"""One rank of a multi-process run of the HIP engine (started by
tests/test_gpu_dist.py as a fresh child process; never imported by pytest).

    python tests/dist_worker.py <backend> <count> <low> <high> <out.pt>

Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT like a
torchrun worker, runs `emphases_amd.dist.from_alignments_and_audios` with the
DEFAULT compute (the HIP engine on this rank's GPU) on a seeded corpus and
saves what it returned.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def corpus(count, low, high):
    import emphases_amd
    from emphases_amd import synth
    frames = synth.corpus_frames(count, low, high)
    audios = [torch.from_numpy(synth.audio(500 + i, int(n)))
              for i, n in enumerate(frames)]
    aligns = [emphases_amd.Alignment.from_frames(
        synth.word_frames(500 + i, int(n))) for i, n in enumerate(frames)]
    return aligns, audios


def main():
    backend, count, low, high, out = sys.argv[1:6]
    rank = int(os.environ['RANK'])
    world = int(os.environ['WORLD_SIZE'])
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    from emphases_amd import dist as edist
    device_id = None
    if backend == 'nccl':
        index = int(os.environ.get('LOCAL_RANK', rank)) % \
            torch.cuda.device_count()
        torch.cuda.set_device(index)
        device_id = torch.device('cuda', index)
    torch.distributed.init_process_group(
        backend, rank=rank, world_size=world, device_id=device_id)
    try:
        aligns, audios = corpus(int(count), int(low), int(high))
        scores = edist.from_alignments_and_audios(aligns, audios)
        from emphases_amd import runtime
        runtime.library()          # the native library is what ran
        torch.save({
            'scores': [s.cpu() for s in scores],
            'device': str(scores[0].device),
            'current_device': torch.cuda.current_device(),
            'shard': len(edist.assign(edist.cost(
                [a.shape[-1] // 160 for a in audios]), world)[rank]),
        }, out)
    finally:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
