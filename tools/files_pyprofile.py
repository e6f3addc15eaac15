"""cProfile of the Python side of one `from_files_to_files` call on fresh alignments, on one
thread's clock per stage: where the interpreter-lock time goes.
usage (GPU box): python tools/files_pyprofile.py"""
import cProfile
import io
import os
import pstats
import shutil
import sys
import tempfile
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emphases_amd  # noqa: E402
from emphases_amd import load, synth  # noqa: E402


def main():
    count = 4096
    directory = tempfile.mkdtemp(prefix='emph_pp_', dir='/dev/shm')
    try:
        waves = []
        for index in range(32):
            wave = os.path.join(directory, f'a{index}.wav')
            load.save_wav(wave, synth.audio(index, 1000))
            waves.append(wave)
        sets = []
        for lap in range(3):
            texts = []
            for index in range(count):
                text = os.path.join(directory, f'u{lap}_{index}.TextGrid')
                emphases_amd.Alignment.from_frames(
                    synth.word_frames(100000 * lap + index, 1000)).save(text)
                texts.append(text)
            sets.append(texts)
        audio = [waves[i % 32] for i in range(count)]
        prefixes = [os.path.join(directory, f'o{i}') for i in range(count)]
        emphases_amd.from_files_to_files(sets[0], audio, prefixes, gpu=0)
        emphases_amd.from_files_to_files(sets[1], audio, prefixes, gpu=0)
        # every thread gets its own profiler
        profilers = {}
        original = threading.Thread.run

        def run(self):
            profiler = cProfile.Profile()
            profilers[self.name] = profiler
            profiler.enable()
            try:
                original(self)
            finally:
                profiler.disable()
        threading.Thread.run = run
        main_profiler = cProfile.Profile()
        main_profiler.enable()
        emphases_amd.from_files_to_files(sets[2], audio, prefixes, gpu=0)
        main_profiler.disable()
        threading.Thread.run = original
        profilers['caller'] = main_profiler
        for name, profiler in profilers.items():
            out = io.StringIO()
            stats = pstats.Stats(profiler, stream=out).sort_stats('tottime')
            stats.print_stats(14)
            text = out.getvalue()
            print(f'==== thread {name}')
            print('\n'.join(line for line in text.splitlines() if line.strip())[:2600])
    finally:
        shutil.rmtree(directory, ignore_errors=True)


if __name__ == '__main__':
    main()
