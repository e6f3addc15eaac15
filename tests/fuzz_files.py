"""Differential fuzzing of the library's file readers (csrc/files.hip) against the Python
readers (alignment.py, load.py): mutated TextGrid and WAVE files must never crash the library,
and whatever the library accepts must be what the Python reader makes of the same bytes.

    python tests/fuzz_files.py [cases] [seed]        (the test suite runs a few hundred)
"""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emphases_amd                                     # noqa: E402
from emphases_amd import files, load                    # noqa: E402


def mutate(data, rng):
    """One of: flip bytes, cut, duplicate a slice, delete a slice, splice in digits / quotes."""
    data = bytearray(data)
    kind = rng.integers(0, 7)
    if not data:
        return bytes(data)
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
    elif kind == 1:
        del data[int(rng.integers(0, len(data))):]
    elif kind == 2:
        a = int(rng.integers(0, len(data)))
        b = min(len(data), a + int(rng.integers(1, 40)))
        data[a:a] = data[a:b]
    elif kind == 3:
        a = int(rng.integers(0, len(data)))
        del data[a:a + int(rng.integers(1, 40))]
    elif kind == 4:
        a = int(rng.integers(0, len(data)))
        token = [b'"', b'""', b'-1', b'1e308', b'nan', b'=', b'\n', b'\x00', b'9' * 30,
                 b'intervals: size = 99999999', b'\r\n', b'0x10'][int(rng.integers(0, 12))]
        data[a:a] = token
    elif kind == 5:                                     # swap two lines
        lines = bytes(data).split(b'\n')
        if len(lines) > 2:
            i, j = rng.integers(0, len(lines), 2)
            lines[i], lines[j] = lines[j], lines[i]
        data = bytearray(b'\n'.join(lines))
    else:                                               # a 32-bit field of a binary header
        if len(data) >= 8:
            a = int(rng.integers(0, len(data) - 4))
            value = [0, 1, 0xFFFFFFFF, 0x7FFFFFFF, len(data), len(data) + 1,
                     int(rng.integers(0, 1 << 32))][int(rng.integers(0, 7))]
            data[a:a + 4] = struct.pack('<I', value)
    return bytes(data)


def python_alignment(path):
    try:
        words = emphases_amd.Alignment(path)
        return ('ok', [(str(w), w.start(), w.end(), tuple(
            (str(p), p.start(), p.end()) for p in (w.phonemes or []))) for w in words],
                words.tiers)
    except Exception as error:                          # noqa: BLE001
        return ('error', type(error).__name__)


def library_alignment(opened, index):
    try:
        words = opened.alignment(index)
        return ('ok', [(str(w), w.start(), w.end(), tuple(
            (str(p), p.start(), p.end()) for p in (w.phonemes or []))) for w in words],
                words.tiers)
    except Exception as error:                          # noqa: BLE001
        return ('error', type(error).__name__)


def python_audio(path):
    try:
        samples, rate = load.wav(path, raw=True)
        return ('ok', tuple(samples.shape), str(samples.dtype), int(rate),
                samples.numpy().tobytes())
    except Exception as error:                          # noqa: BLE001
        return ('error', type(error).__name__)


def library_audio(opened, index):
    try:
        audio, rate = opened.audio(index)
        if isinstance(audio, files.FileAudio):
            itemsize = 2 if 'int16' in str(audio.dtype) else 4
            buffer = np.zeros(audio.shape[0] * itemsize + 16, dtype=np.uint8)
            opened.read([index], [0], [audio.shape[0] * itemsize], buffer.ctypes.data)
            return ('ok', (1, audio.shape[0]), str(audio.dtype), int(rate),
                    buffer[:audio.shape[0] * itemsize].tobytes())
        return ('ok', tuple(audio.shape), str(audio.dtype), int(rate), audio.numpy().tobytes())
    except Exception as error:                          # noqa: BLE001
        return ('error', type(error).__name__)


def run(directory, grids, waves, cases, seed):
    """-> list of disagreements (empty = pass)."""
    rng = np.random.default_rng(seed)
    texts, audios = [], []
    for case in range(cases):
        grid = grids[int(rng.integers(0, len(grids)))]
        wave = waves[int(rng.integers(0, len(waves)))]
        data, sound = open(grid, 'rb').read(), open(wave, 'rb').read()
        for _ in range(int(rng.integers(1, 3))):
            data = mutate(data, rng)
        for _ in range(int(rng.integers(0, 3))):
            sound = mutate(sound[:64], rng) + sound[64:] if rng.integers(0, 2) else \
                mutate(sound, rng)
        text, audio = os.path.join(directory, f'f{case}.TextGrid'), \
            os.path.join(directory, f'f{case}.wav')
        open(text, 'wb').write(data)
        open(audio, 'wb').write(sound)
        texts.append(text)
        audios.append(audio)
    opened = files.FileBatch(texts, audios, threads=4)
    problems = []
    for index in range(cases):
        want, got = python_alignment(texts[index]), library_alignment(opened, index)
        if want != got:
            problems.append((texts[index], int(opened.status[index]), want[:2], got[:2],
                             open(texts[index], 'rb').read()))
        want, got = python_audio(audios[index]), library_audio(opened, index)
        if want != got:
            problems.append((audios[index], int(opened.status[index]), want[:4], got[:4],
                             open(audios[index], 'rb').read()))
    return problems


def run_writer(directory, cases, seed):
    """Random alignments (labels with quotes, blanks of every kind, text beyond the BMP; times
    that exercise repr(float)) saved by alignment.py, read by the library, written back by the
    library: the bytes `Alignment.save` writes, the tensor `torch.save` would hold."""
    import torch
    from emphases_amd import alignment as al
    rng = np.random.default_rng(seed)
    alphabet = ['a', 'B', '"', '""', ' ', '\t', 'é', 'ü', '日', '\U0001F600', '=', '<x>', '1.5',
                '\\', 'sp', '<silent>', '\u00a0', '\u2003', "'"]
    times = [0., 1e-7, 0.1 + 0.2, 1 / 3, 2.5, 1e5 / 3, 123456789.125, 1e15 + 0.5, 3e-5, 7.0,
             0.30000000000000004, 1e16, 3e16]
    texts = []
    for case in range(cases):
        count = int(rng.integers(1, 8))
        edges = np.sort(rng.choice(times, size=count + 1, replace=False))
        with_phones = bool(rng.integers(0, 2))
        words = []
        for i in range(count):
            label = ''.join(rng.choice(alphabet, size=int(rng.integers(0, 4))))
            phones = None
            if with_phones:
                pieces = int(rng.integers(1, 3))
                cuts = np.linspace(edges[i], edges[i + 1], pieces + 1)
                phones = [al.Phoneme(''.join(rng.choice(alphabet, size=int(rng.integers(0, 3)))),
                                     cuts[j], cuts[j + 1]) for j in range(pieces)]
            words.append(al.Word(label, edges[i], edges[i + 1], phones))
        path = os.path.join(directory, f'w{case}.TextGrid')
        al.Alignment(words).save(path)
        texts.append(path)
    wave = os.path.join(directory, 'silence.wav')
    load.save_wav(wave, np.zeros((1, 100), dtype='f4'))
    opened = files.FileBatch(texts, [wave] * cases)
    problems = []
    scores = [torch.from_numpy(rng.standard_normal((1, len(opened.alignment(i)))).astype('f4'))
              for i in range(cases)]
    prefixes = [os.path.join(directory, f'out{i}') for i in range(cases)]
    opened.write(list(range(cases)), prefixes, scores)
    for index in range(cases):
        if python_alignment(texts[index]) != library_alignment(opened, index):
            problems.append((texts[index], 'read'))
        emphases_amd.Alignment(texts[index]).save(os.path.join(directory, 'python.TextGrid'))
        if open(prefixes[index] + '.TextGrid', 'rb').read() != \
                open(os.path.join(directory, 'python.TextGrid'), 'rb').read():
            problems.append((texts[index], 'TextGrid written'))
        if not torch.equal(torch.load(prefixes[index] + '.pt'), scores[index]):
            problems.append((texts[index], '.pt written'))
    return problems


def corpus(directory):
    """Seed files: every TextGrid form the readers take, a few WAVE layouts."""
    import test_host
    from pathlib import Path
    grids = [str(p) for p in test_host._grid_variants(Path(directory))]
    waves = []
    rng = np.random.default_rng(5)
    for index, (rate, kind) in enumerate([(16000, 'pcm'), (8000, 'pcm'), (22050, 'float'),
                                          (16000, 'stereo'), (44100, 'pcm8')]):
        path = os.path.join(directory, f'seed{index}.wav')
        samples = (rng.standard_normal((2 if kind == 'stereo' else 1, 300)) * 0.3).astype('f4')
        if kind == 'float':
            body = samples.astype('<f4').tobytes()
            fmt = struct.pack('<HHIIHH', 3, 1, rate, rate * 4, 4, 32)
        elif kind == 'pcm8':
            body = ((samples[0] * 100 + 128).astype('u1')).tobytes()
            fmt = struct.pack('<HHIIHH', 1, 1, rate, rate, 1, 8)
        else:
            load.save_wav(path, samples, rate)
            waves.append(path)
            continue
        chunks = b'fmt ' + struct.pack('<I', 16) + fmt + b'LIST' + struct.pack('<I', 3) + \
            b'abc\0' + b'data' + struct.pack('<I', len(body)) + body
        open(path, 'wb').write(b'RIFF' + struct.pack('<I', 4 + len(chunks)) + b'WAVE' + chunks)
        waves.append(path)
    return grids, waves


if __name__ == '__main__':
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    with tempfile.TemporaryDirectory(dir='/dev/shm') as directory:
        grids, waves = corpus(directory)
        found = []
        for start in range(0, cases, 500):
            found += run(directory, grids, waves, min(500, cases - start), seed + start)
        os.makedirs('/tmp/fuzz_found', exist_ok=True)
        for number, problem in enumerate(found[:40]):
            print(problem[:4])
            name = f'{number}_' + os.path.basename(problem[0])
            open(os.path.join('/tmp/fuzz_found', name), 'wb').write(problem[4])
        print(f'{cases} cases, {len(found)} disagreement(s)')
        written = run_writer(directory, min(cases, 1000), seed)
        print(f'writer: {min(cases, 1000)} cases, {len(written)} problem(s)', written[:5])
