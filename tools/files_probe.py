"""Per-call durations of the library's file calls inside a real from_files_to_files run
(GPU box).  python tools/files_probe.py [files] [threads]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import emphases_amd  # noqa: E402
from emphases_amd import files, load, synth  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
if len(sys.argv) > 2:
    files.THREADS = int(sys.argv[2])
distinct = int(os.environ.get('DISTINCT', 32))
directory = tempfile.mkdtemp(prefix='emph_probe_', dir='/dev/shm')
log = {}


def timed(cls, name):
    original = getattr(cls, name)

    def wrapper(*args, **kwargs):
        start = time.perf_counter()
        try:
            return original(*args, **kwargs)
        finally:
            log.setdefault(f'{cls.__name__}.{name}', []).append(
                time.perf_counter() - start)
    setattr(cls, name, wrapper)


try:
    texts, waves, prefixes = [], [], []
    for index in range(count):
        wave = os.path.join(directory, f'a{index % distinct}.wav')
        if index < distinct:
            load.save_wav(wave, synth.audio(index % 32, 1000))
        else:
            link = os.path.join(directory, f'a{index}.wav')
            os.link(wave, link)
            wave = link
        text = os.path.join(directory, f'u{index}.TextGrid')
        emphases_amd.Alignment.from_frames(
            synth.word_frames(3000 + index, 1000)).save(text)
        texts.append(text), waves.append(wave)
        prefixes.append(os.path.join(directory, f'o{index}'))
    emphases_amd.from_files_to_files(texts[:1024], waves[:1024], prefixes[:1024], gpu=0, utterances_per_batch=int(os.environ.get('PER_BATCH', 256)))
    from emphases_amd import session as session_module
    timed(files.FileBatch, '__init__')
    timed(files.FileBatch, 'read')
    timed(files.FileBatch, 'write')
    timed(session_module.Session, 'submit')
    timed(session_module.Pending, 'result')
    timed(session_module._Lane, 'stage')
    from emphases_amd import batch as batch_module, engine as engine_module
    timed(engine_module.Engine, 'prepare')
    timed(engine_module.Engine, 'upload')
    timed(engine_module.Engine, 'forward')
    def cpu_stat():
        try:
            return dict(line.split() for line in open('/sys/fs/cgroup/cpu.stat'))
        except OSError:
            return {}
    if os.environ.get('TORCH_THREADS'):
        torch.set_num_threads(int(os.environ['TORCH_THREADS']))
    print('torch threads', torch.get_num_threads(), 'interop', torch.get_num_interop_threads(),
          'process threads', len(os.listdir('/proc/self/task')))
    before = cpu_stat()
    start = time.perf_counter()
    emphases_amd.from_files_to_files(texts, waves, prefixes, gpu=0, utterances_per_batch=int(os.environ.get('PER_BATCH', 256)))
    total = time.perf_counter() - start
    after = cpu_stat()
    print('  cgroup:', {k: int(after[k]) - int(before[k]) for k in after
                        if k in ('usage_usec', 'nr_periods', 'nr_throttled', 'throttled_usec')})
    print(f'{count} files, {files.THREADS} threads, {distinct} distinct wavs: '
          f'{total * 1e3:.1f} ms = {count / total:.0f} files/s')
    for name, values in log.items():
        values = np.array(values) * 1e3
        print(f'  {name:22s} x{len(values):3d}  sum {values.sum():7.1f} ms  '
              f'median {np.median(values):6.2f}  max {values.max():6.2f}  '
              f'first {values[:4].round(2).tolist()}')
    # the same reads with the GPU idle
    opened = files.FileBatch(texts[:256], waves[:256])
    pinned = torch.empty(256 * 160000, dtype=torch.int16).pin_memory()
    where = np.arange(256) * 320000
    for _ in range(3):
        start = time.perf_counter()
        for piece in range(4):
            chosen = list(range(64 * piece, 64 * piece + 64))
            files.FileBatch.read.__wrapped__ if False else None
            opened.read(chosen, where[chosen], [320000] * 64, pinned.data_ptr())
        print('  idle: 4 reads of 64 files', (time.perf_counter() - start) * 1e3, 'ms')
finally:
    shutil.rmtree(directory, ignore_errors=True)
