"""`from_files_to_files` on alignments never seen before (the bench's `files_api` regime) for
several `utterances_per_batch`: the per-batch Python work is what bounds the call.
usage (GPU box): python tools/files_batchsize.py [files]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emphases_amd  # noqa: E402
from emphases_amd import load, synth  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    # (utterances_per_batch, threads of an open call, threads of a write call)
    sizes = ((512, 8, 4), (512, 6, 4), (1024, 8, 4), (384, 8, 4), (256, 8, 4))
    laps_per_size = 4
    directory = tempfile.mkdtemp(prefix='emph_bs_', dir='/dev/shm')
    try:
        waves = []
        for index in range(32):
            wave = os.path.join(directory, f'a{index}.wav')
            load.save_wav(wave, synth.audio(index, 1000))
            waves.append(wave)
        sets = []
        for lap in range(len(sizes) * laps_per_size + 1):
            texts = []
            for index in range(count):
                text = os.path.join(directory, f'u{lap}_{index}.TextGrid')
                emphases_amd.Alignment.from_frames(
                    synth.word_frames(100000 * lap + index, 1000)).save(text)
                texts.append(text)
            sets.append(texts)
        audio = [waves[i % 32] for i in range(count)]
        prefixes = [os.path.join(directory, f'o{i}') for i in range(count)]
        emphases_amd.from_files_to_files(sets[-1], audio, prefixes, gpu=0, utterances_per_batch=1024)
        emphases_amd.from_files_to_files(sets[-1], audio, prefixes, gpu=0)
        lap = 0
        import gc
        from emphases_amd import session as session_module
        if len(sys.argv) > 2:
            session_module.FILE_BUFFERS = int(sys.argv[2])
        print('FILE_BUFFERS', session_module.FILE_BUFFERS)
        for size, opening, writing in sizes:
            os.environ['EMPHASES_OPEN_THREADS'] = str(opening)
            os.environ['EMPHASES_WRITE_THREADS'] = str(writing)
            times = []
            for _ in range(laps_per_size):
                gc.collect()
                start = time.perf_counter()
                emphases_amd.from_files_to_files(sets[lap], audio, prefixes, gpu=0, utterances_per_batch=size)
                times.append(time.perf_counter() - start)
                lap += 1
            print(f'utterances_per_batch {size:5d}, file pools open {opening:2d} / write {writing}: laps (ms) '
                  + ' '.join(f'{t * 1e3:.1f}' for t in times)
                  + f' -> best {count / min(times):.0f} files/s, median {count / sorted(times)[len(times) // 2]:.0f}')
    finally:
        shutil.rmtree(directory, ignore_errors=True)


if __name__ == '__main__':
    main()
