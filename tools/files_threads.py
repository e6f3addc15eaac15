"""Which threads of the process burn CPU during `from_files_to_files`, and is the cgroup
throttling it?  Per-thread utime + stime (/proc/self/task/*/stat) and the cgroup's
nr_throttled / throttled_usec (/sys/fs/cgroup/cpu.stat) around ten calls.

usage (GPU box): python tools/files_threads.py [files]
"""
import collections
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emphases_amd  # noqa: E402
from emphases_amd import load, synth  # noqa: E402

TICK = os.sysconf('SC_CLK_TCK')


def threads():
    result = {}
    for task in os.listdir('/proc/self/task'):
        try:
            with open(f'/proc/self/task/{task}/stat') as file:
                text = file.read()
        except OSError:
            continue
        name = text[text.index('(') + 1:text.rindex(')')]
        fields = text[text.rindex(')') + 2:].split()
        result[int(task)] = (name, (int(fields[11]) + int(fields[12])) / TICK)
    return result


def cgroup():
    try:
        with open('/sys/fs/cgroup/cpu.stat') as file:
            return {line.split()[0]: int(line.split()[1]) for line in file}
    except OSError:
        return {}


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    directory = tempfile.mkdtemp(prefix='emph_threads_', dir='/dev/shm')
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
            emphases_amd.Alignment.from_frames(synth.word_frames(3000 + index, 1000)).save(text)
            texts.append(text), waves.append(wave)
            prefixes.append(os.path.join(directory, f'o{index}'))
        emphases_amd.from_files_to_files(texts[:512], waves[:512], prefixes[:512], gpu=0)
        emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
        before, stat_before = threads(), cgroup()
        start = time.perf_counter()
        laps = []
        for _ in range(10):
            lap = time.perf_counter()
            emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
            laps.append(time.perf_counter() - lap)
        wall = time.perf_counter() - start
        after, stat_after = threads(), cgroup()
        print('laps (ms):', ' '.join(f'{lap * 1e3:.0f}' for lap in laps))
        import gc
        from emphases_amd import core
        pauses, began = [], [0]

        def watch(phase, info):
            if phase == 'start':
                began[0] = time.perf_counter()
            else:
                pauses.append((info['generation'], time.perf_counter() - began[0], info['collected']))
        gc.callbacks.append(watch)
        for _ in range(10):
            emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
        gc.callbacks.remove(watch)
        print(f'{len(gc.get_objects())} objects tracked by the collector; ten more laps: '
              + ', '.join(f'{sum(1 for g, _, _ in pauses if g == generation)} collections of generation {generation} '
                          f'({sum(t for g, t, _ in pauses if g == generation) * 1e3:.0f} ms, longest '
                          f'{max([t for g, t, _ in pauses if g == generation] or [0]) * 1e3:.0f} ms)'
                          for generation in (0, 1, 2)))
        gc.collect()
        gc.disable()
        quiet = []
        for _ in range(10):
            lap = time.perf_counter()
            emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
            quiet.append(time.perf_counter() - lap)
        gc.enable()
        print('laps with the garbage collector off (ms):', ' '.join(f'{lap * 1e3:.0f}' for lap in quiet))
        # the slowest of ten stamped laps: where is the gap?
        worst, worst_events = 0., None
        for _ in range(10):
            core.TIMELINE = []
            lap = time.perf_counter_ns()
            emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
            took = time.perf_counter_ns() - lap
            if took > worst:
                worst, worst_events, worst_start = took, core.TIMELINE, lap
        core.TIMELINE = None
        print(f'slowest of ten stamped laps: {worst * 1e-6:.0f} ms; stages longer than 12 ms:')
        for stage, position, a, b in sorted(worst_events, key=lambda e: e[2]):
            if b - a > 12e6:
                print(f'   {stage:7s} batch {position:2d}: {(a - worst_start) * 1e-6:7.1f} .. {(b - worst_start) * 1e-6:7.1f} ms')
        print(f'wall {wall * 1e3:.0f} ms; cgroup: ' + ', '.join(
            f'{key} +{stat_after[key] - stat_before.get(key, 0)}' for key in
            ('nr_periods', 'nr_throttled', 'throttled_usec', 'usage_usec') if key in stat_after))
        by_name = collections.defaultdict(lambda: [0, 0.])
        for task, (name, seconds) in after.items():
            spent = seconds - before.get(task, (name, 0.))[1]
            by_name[name][0] += 1
            by_name[name][1] += spent
        total = sum(v[1] for v in by_name.values())
        print(f'{len(after)} threads, {total * 1e3:.0f} ms of CPU = {total / wall:.1f} CPUs busy on average')
        for name, (number, seconds) in sorted(by_name.items(), key=lambda item: -item[1][1])[:12]:
            print(f'  {name:18s} x{number:4d}  {seconds * 1e3:8.0f} ms')
    finally:
        shutil.rmtree(directory, ignore_errors=True)


if __name__ == '__main__':
    main()
