"""One rank of a multi-process run of the HIP engine (started by
tests/test_gpu_dist.py as a fresh child process; never imported by pytest).

    python tests/dist_worker.py <backend> <count> <low> <high> <out.pt>
    python tests/dist_worker.py <backend> files <directory> <count> <out.pt>

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

# (tests/test_gpu_dist.py: the opt-in precisions under N > 1 ranks)
PRECISION = os.environ.get('EMPHASES_TEST_PRECISION', 'f32')


def corpus(count, low, high):
    import emphases_amd
    from emphases_amd import synth
    frames = synth.corpus_frames(count, low, high)
    audios = [torch.from_numpy(synth.audio(500 + i, int(n)))
              for i, n in enumerate(frames)]
    aligns = [emphases_amd.Alignment.from_frames(
        synth.word_frames(500 + i, int(n))) for i, n in enumerate(frames)]
    return aligns, audios


def write_files(directory, count, low=200, high=3000):
    """The seeded corpus as <directory>/u<i>.wav (16-bit PCM; every 7th file
    at 8 kHz, every 11th at 22.05 kHz) + u<i>.TextGrid."""
    import emphases_amd
    from emphases_amd import load, synth
    frames = synth.corpus_frames(count, low, high)
    for index, n in enumerate(frames):
        rate = 8000 if index % 7 == 3 else 22050 if index % 11 == 5 else 16000
        samples = int(n) * rate // 100
        if rate == 16000:
            audio = synth.audio(500 + index, int(n))
        else:
            audio = synth.weights(700 + index, (1, samples), 0.3)
        load.save_wav(os.path.join(directory, f'u{index}.wav'), audio, rate)
        emphases_amd.Alignment.from_frames(
            synth.word_frames(500 + index, int(n))).save(
                os.path.join(directory, f'u{index}.TextGrid'))


def file_lists(directory, count, tag):
    texts = [os.path.join(directory, f'u{i}.TextGrid') for i in range(count)]
    audios = [os.path.join(directory, f'u{i}.wav') for i in range(count)]
    prefixes = [os.path.join(directory, f'{tag}_{i}') for i in range(count)]
    return texts, audios, prefixes


def files_main(backend, directory, count, out):
    """`dist.from_files_to_files` on this rank; records which audio files this
    process read the samples of."""
    rank = int(os.environ['RANK'])
    world = int(os.environ['WORLD_SIZE'])
    from emphases_amd import dist as edist, load
    torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    try:
        # which audio files this process reads the SAMPLES of: through the
        # library's batched reader (files.FileBatch.read) or load.wav
        from emphases_amd import files
        read = []
        original = load.wav

        def tracking_wav(file, raw=False):
            read.append(os.path.basename(str(file)))
            return original(file, raw)
        load.wav = tracking_wav
        original_read = files.FileBatch.read

        def tracking_read(self, indices, where, nbytes, destination):
            read.extend(os.path.basename(self.audio_files[i]) for i in indices)
            return original_read(self, indices, where, nbytes, destination)
        files.FileBatch.read = tracking_read
        tag = f'w{world}' if PRECISION == 'f32' else f'w{world}_{PRECISION}'
        texts, audios, prefixes = file_lists(directory, count, tag)
        scores = edist.from_files_to_files(
            texts, audios, prefixes, precision=PRECISION)
        torch.save({'scores': [s.cpu() for s in scores], 'read': sorted(read)},
                   out)
    finally:
        torch.distributed.destroy_process_group()


def main():
    if sys.argv[2] == 'files':
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        return files_main(sys.argv[1], sys.argv[3], int(sys.argv[4]),
                          sys.argv[5])
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
        scores = edist.from_alignments_and_audios(
            aligns, audios, precision=PRECISION)
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
