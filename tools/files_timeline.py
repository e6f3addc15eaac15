"""Where do the three stages of `from_files_to_files` run?  (round-4 review, weak #5: 5.3 ms per
256-file batch measured against stages of 2.5 / 1.5 / 1.3 ms.)  `core.TIMELINE` collects
(stage, batch, start, end) stamps of the opener thread, the caller (submit, scores) and the
writer thread; this prints them per batch next to each other, the busy time of every stage,
how much of it overlaps, and the laps of five runs.

usage (GPU box): python tools/files_timeline.py [files] > gpurun_out/r5_files_timeline.txt
"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emphases_amd  # noqa: E402
from emphases_amd import core, files, load, synth  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    directory = tempfile.mkdtemp(prefix='emph_timeline_', dir='/dev/shm')
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
        # (every pinned buffer of the session and every output file exists afterwards: the
        # laps overwrite, bench.py's create)
        emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
        import gc
        gc.collect()
        from emphases_amd import session as session_module
        openers = session_module.FILE_BUFFERS - 2
        if len(sys.argv) > 2 and sys.argv[2] == 'sweep':
            # laps only, for several pool sizes (threads per open call, per write call)
            for opening, writing in ((2, 2), (3, 3), (3, 5), (4, 2), (4, 4), (5, 2), (6, 2), (6, 4), (8, 4)):
                os.environ['EMPHASES_OPEN_THREADS'] = str(opening)
                os.environ['EMPHASES_WRITE_THREADS'] = str(writing)
                laps = []
                for _ in range(5):
                    start = time.perf_counter()
                    emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
                    laps.append(time.perf_counter() - start)
                print(f'{openers} openers x {opening} threads, writer x {writing}: laps (ms) '
                      + ' '.join(f'{lap * 1e3:.1f}' for lap in laps)
                      + f' -> median {count / sorted(laps)[2]:.0f} files/s')
            return
        print(f'{count} files of 10 s (16-bit PCM, /dev/shm), 256 per batch; CPU budget '
              f'{files._cpu_budget()}, {openers} openers, file-pool threads per call (open, write) = '
              f'{files.stage_threads(openers)}')
        laps = []
        for _ in range(5):
            start = time.perf_counter()
            emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
            laps.append(time.perf_counter() - start)
        print('laps (ms):', ' '.join(f'{lap * 1e3:.1f}' for lap in laps),
              f'-> median {sorted(laps)[2] * 1e3:.1f} ms = {count / sorted(laps)[2]:.0f} files/s, '
              f'worst {max(laps) / sorted(laps)[2] - 1:+.0%} of the median')
        core.TIMELINE = []
        start = time.perf_counter_ns()
        emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0)
        total = time.perf_counter_ns() - start
        events, core.TIMELINE = core.TIMELINE, None
        print(f'one more run with stamps: {total * 1e-6:.1f} ms')
        print(f'{"batch":>5s} ' + ' '.join(f'{stage:>22s}' for stage in ('open', 'submit', 'scores', 'write'))
              + '   (start .. end, ms from the call)')
        batches = sorted({position for _, position, _, _ in events})
        for position in batches:
            row = []
            for stage in ('open', 'submit', 'scores', 'write'):
                spans = [(a - start, b - start) for s, p, a, b in events if s == stage and p == position]
                row.append(' '.join(f'{a * 1e-6:8.2f} ..{b * 1e-6:8.2f}' for a, b in spans) or ' ' * 20)
            print(f'{position:5d}   ' + '   '.join(row))
        busy = {}
        for stage in ('open', 'open.parse', 'open.objects', 'open.read', 'open.plan', 'submit', 'scores', 'write'):
            spans = [(a, b) for s, _, a, b in events if s == stage]
            busy[stage] = sum(b - a for a, b in spans)
            print(f'{stage:>7s}: busy {busy[stage] * 1e-6:7.1f} ms = {busy[stage] * 1e-6 / len(batches):5.2f} ms per batch '
                  f'({100. * busy[stage] / total:4.1f} % of the call)')
        # how much of the opener's time runs beside the caller or the writer
        def overlap(first, second):
            total_overlap = 0
            for s1, _, a1, b1 in events:
                if s1 != first:
                    continue
                for s2, _, a2, b2 in events:
                    if s2 in second:
                        total_overlap += max(0, min(b1, b2) - max(a1, a2))
            return total_overlap
        print(f'open beside submit / scores / write: {overlap("open", ("submit", "scores", "write")) * 1e-6:.1f} ms '
              f'of {busy["open"] * 1e-6:.1f}; write beside open / submit / scores: '
              f'{overlap("write", ("open", "submit", "scores")) * 1e-6:.1f} ms of {busy["write"] * 1e-6:.1f}')
        print(f'the longest stage alone would take {max(busy.values()) * 1e-6:.1f} ms; the call took {total * 1e-6:.1f}')
    finally:
        shutil.rmtree(directory, ignore_errors=True)


if __name__ == '__main__':
    main()
