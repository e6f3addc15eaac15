"""Where a `from_files_to_files` call spends its host time (cProfile on the GPU box).
python tools/files_profile.py [files]"""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emphases_amd  # noqa: E402
from emphases_amd import load, synth  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
directory = tempfile.mkdtemp(prefix='emph_prof_', dir='/dev/shm')
try:
    texts, waves, prefixes = [], [], []
    for index in range(count):
        wave = os.path.join(directory, f'a{index % 32}.wav')
        if index < 32:
            load.save_wav(wave, synth.audio(index, 1000))
        else:
            link = os.path.join(directory, f'a{index}.wav')
            os.link(wave, link)
            wave = link
        text = os.path.join(directory, f'u{index}.TextGrid')
        emphases_amd.Alignment.from_frames(
            synth.word_frames(3000 + index, 1000)).save(text)
        texts.append(text), waves.append(wave)
        prefixes.append(os.path.join(directory, f'o{index}'))
    emphases_amd.from_files_to_files(texts[:512], waves[:512], prefixes[:512], gpu=0)
    from emphases_amd import files
    for threads in (4, 8, 16):
        files.THREADS = threads
        laps = []
        for _ in range(3):
            start = time.perf_counter()
            emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
            laps.append(time.perf_counter() - start)
        print(f'{threads:2d} file threads: {min(laps) * 1e3:7.1f} ms '
              f'= {count / min(laps):8.0f} files/s  (laps {laps})')
    files.THREADS = 8
    profiler = cProfile.Profile()
    profiler.enable()
    emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
    profiler.disable()
    pstats.Stats(profiler).sort_stats('cumulative').print_stats(45)
finally:
    shutil.rmtree(directory, ignore_errors=True)
