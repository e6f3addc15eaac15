"""CPU: host-side logic of the product (chunking, bounds, layout, file I/O),
and the C-ABI library's exported surface.  No kernel is launched here."""
import hashlib
import os
import re
import struct

import numpy as np
import pytest
import torch

import emphases_amd
from conftest import ROOT, case_names, seconds
from emphases_amd import (alignment, batch, config as cfg, convert, load,
                          melbasis, runtime, synth, weights)
from oracle import librosa_mel
from oracle import prominence as oracle


###############################################################################
# Chunking (emphases/core.py:345-418)
###############################################################################


def plans(chunk_goldens):
    return case_names(chunk_goldens)


def test_chunk_plans_match_reference(chunk_goldens):
    names = plans(chunk_goldens)
    assert 'dropped_chunk_b0' in names and 'floor_b300' in names
    for name in names:
        bounds = chunk_goldens[f'{name}/bounds_frames'].astype(np.int64)
        batch_size = int(chunk_goldens[f'{name}/batch_size'])
        batch_size = None if batch_size < 0 else batch_size
        samples = int(bounds[1, -1]) * 160
        want_frames = chunk_goldens[f'{name}/chunk_frames']
        want_words = chunk_goldens[f'{name}/chunk_words']
        want_bounds = chunk_goldens[f'{name}/chunk_bounds']

        # product
        words = emphases_amd.Alignment.from_frames(bounds)
        segments = batch.chunk_utterance(words, samples, batch_size)
        assert [s.frames for s in segments] == want_frames.tolist(), name
        assert [s.bounds.shape[1] for s in segments] == \
            want_words.tolist(), name
        np.testing.assert_array_equal(
            np.concatenate([s.bounds for s in segments], axis=1), want_bounds)
        if batch_size is None:
            # ... and the library's one pass over a batch (emph_plan_batch), here a
            # batch of this utterance three times over
            plan = batch.plan_batch([words] * 3, [samples] * 3)
            assert plan.frames.tolist() == want_frames.tolist() * 3, name
            np.testing.assert_array_equal(
                plan.segment_bounds, np.concatenate([want_bounds] * 3, axis=1))

        # oracle
        kept = [c for c in oracle.chunks(seconds(bounds), samples, batch_size)
                if not c['dropped']]
        assert [(c['end_sample'] - c['start_sample']) // 160 for c in kept] \
            == want_frames.tolist(), name
        np.testing.assert_array_equal(
            np.concatenate([c['bounds'] for c in kept], axis=1), want_bounds)


def test_dropped_chunk_has_no_scores(chunk_goldens):
    """core.py:414-415 swallows the reflect-pad error: the 2-frame chunk's
    word gets no score."""
    bounds = chunk_goldens['dropped_chunk_b0/bounds_frames']
    assert bounds.shape[1] == 4
    assert chunk_goldens['dropped_chunk_b0/chunk_words'].sum() == 3
    words = emphases_amd.Alignment.from_frames(bounds)
    segments = batch.chunk_utterance(words, int(bounds[1, -1]) * 160, 0)
    assert [(s.start_word, s.end_word) for s in segments] == \
        [(0, 1), (2, 3), (3, 4)]


def test_vectorised_chunking_equals_the_scalar_loops():
    """batch.chunk_utterance (array form) against the oracle's word-by-word
    restatement of `core.py:361-400` on random alignments with gaps, float
    times that hit the floor-division quirk, and small batch sizes."""
    from oracle import prominence as oracle
    rng = np.random.default_rng(11)
    for trial in range(300):
        count = int(rng.integers(1, 60))
        step = rng.choice([0.01, 0.013, 0.0803, 0.25])
        lengths = rng.integers(1, 70, count) * step
        gaps = rng.integers(0, 3, count) * step * (trial % 3 == 0)
        starts = np.cumsum(lengths + gaps) - lengths
        words = [(float(s), float(s + d)) for s, d in zip(starts, lengths)]
        samples = int(words[-1][1] * 16000) + int(rng.integers(0, 400))
        size = [None, 50, 200, 1000, 0][trial % 5]
        want = [c for c in oracle.chunks(words, samples, size)
                if not c['dropped']]
        got = batch.chunk_utterance(list(words), samples, size)
        assert len(got) == len(want), (trial, size)
        for mine, theirs in zip(got, want):
            assert (mine.start_word, mine.end_word, mine.start_sample) == (
                theirs['start_word'], theirs['end_word'],
                theirs['start_sample'])
            assert mine.length == theirs['end_sample'] - theirs['start_sample']
            assert np.array_equal(mine.bounds, theirs['bounds'])


def test_float_floor_quirk():
    """convert.py:29-31 floor-divides in float64: 8.03 s is frame 802."""
    assert convert.seconds_to_frames(8.03) == 802.0
    assert convert.seconds_to_frames(8.02) == 802.0
    assert convert.seconds_to_frames(16.06) == 1605.0
    assert convert.seconds_to_frames(1.0) == 100.0
    assert isinstance(convert.seconds_to_frames(1.0), float)
    assert convert.frames_to_samples(802) == 128320
    assert oracle.seconds_to_frames(8.03) == 802.0


def test_chunk_frames_follow_stft_length():
    times = [(0.0, 0.5), (0.5, 1.0)]
    (segment,) = batch.chunk_utterance(times, 16000)
    assert segment.frames == 100 and segment.length == 16000
    assert segment.start_sample == 0


###############################################################################
# Packed layout
###############################################################################


def test_plan_layout():
    segments = []
    lengths = []
    for index, frames in enumerate([1000, 37, 250]):
        bounds = synth.word_frames(index, frames, 3, 40)
        segments.extend(batch.chunk_utterance(
            emphases_amd.Alignment.from_frames(bounds), frames * 160, None,
            index))
        lengths.append(frames * 160)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    plan = batch.Plan(segments, offsets, lengths)
    assert plan.total_frames == 1287
    assert np.all(plan.frame_off % batch.ALIGN == 0)
    assert np.all(plan.word_off % batch.ALIGN == 0)
    assert plan.frame_off[0] >= batch.LEAD
    assert plan.ld_frames % 4 == 0
    assert plan.ld_frames >= plan.frame_off[-1] + plan.frames[-1] + batch.TAIL
    # ranges do not overlap
    ends = plan.frame_off + plan.frames
    assert np.all(plan.frame_off[1:] >= ends[:-1])
    tiles = plan.tiles(runtime.AXIS_FRAMES, 64)
    assert tiles.dtype == np.int32
    assert tiles.shape[1] == runtime.TILE_FIELDS
    assert tiles[:, 0].tolist() == [0] * 16 + [1] + [2] * 4
    assert tiles[16].tolist() == [1, 0, int(plan.frame_off[1]), 37]
    assert tiles[-1].tolist() == [2, 192, int(plan.frame_off[2]), 250]
    # every word column is labelled with its segment, padding with -1
    columns = plan.word_columns()
    assert len(columns) == plan.total_words
    assert np.all(plan.word_segment[columns] >= 0)
    assert (plan.word_segment >= 0).sum() == plan.total_words
    meta, where = plan.pack_metadata([(runtime.AXIS_FRAMES, 64)])
    start, size = where['table']
    assert start == 0 and size == 3 * runtime.SEG_FIELDS * 2
    table = meta[start:start + size].view(np.int64).reshape(3, -1)
    np.testing.assert_array_equal(table, plan.table)
    assert table[1, runtime.SEG_AUDIO_OFF] == 160000


###############################################################################
# Mel basis, weights, synthetic data
###############################################################################


def test_mel_basis_matches_oracle_bitwise():
    ours = melbasis.filterbank()
    theirs = librosa_mel.mel(sr=16000, n_fft=1024, n_mels=80)
    assert np.array_equal(ours, theirs)
    plain = melbasis.SparseBasis(ours, aligned=False)
    assert plain.nnz == plain.values.size == 1001 and plain.read_cycles is None
    # the packing the front-end takes: every run starts at a 16-byte aligned bin
    # at or before the row's first non-zero one, zero weights in between, chosen
    # so that the kernel's LDS reads have no bank conflicts (4 cycles for each of
    # the 6 + 3 sixteen-byte reads of a frame; 69 with `first & ~3`)
    sparse = melbasis.default()
    assert sparse.nnz == 1001 and sparse.read_cycles == 36
    assert (sparse.row_start % 4 == 0).all()
    assert (sparse.row_start <= plain.row_start).all()
    assert sparse.row_count[:64].max() <= 24 and sparse.row_count[64:].max() <= 48
    assert np.array_equal(sparse.row_start + sparse.row_count,
                          plain.row_start + plain.row_count)
    starts, cycles = melbasis.conflict_free_starts(
        plain.row_start, plain.row_count)
    assert cycles == 36 and np.array_equal(starts, sparse.row_start)
    for packing in (plain, sparse):
        dense = np.zeros_like(ours)
        for row in range(80):
            lo, n, off = (packing.row_start[row], packing.row_count[row],
                          packing.row_offset[row])
            dense[row, lo:lo + n] = packing.values[off:off + n]
        assert np.array_equal(dense, ours)


def test_bundled_checkpoint_is_the_reference_weights():
    state = weights.load()
    assert len(state) == 28
    assert sum(v.size for v in state.values()) == 250881
    flat = np.concatenate([v.ravel() for v in state.values()])
    digest = hashlib.sha256(flat.astype('<f4').tobytes()).hexdigest()
    # SURVEY.md App. E
    assert digest == \
        'be2fb6555ba5fab57a4abb3c4e55a6cefc9dbd03c3ccbdcc2c735784aa277e69'


def test_parameter_layouts():
    assert sum(int(np.prod(s)) for s in weights.parameter_shapes().values()) \
        == 250881
    transformer = cfg.Config(architecture='transformer')
    assert sum(int(np.prod(s)) for s in
               weights.parameter_shapes(transformer).values()) == 489921
    inference = cfg.Config(downsample_location='inference')
    assert sum(int(np.prod(s)) for s in
               weights.parameter_shapes(inference).values()) == 135201
    with pytest.raises(ValueError):
        cfg.Config(architecture='lstm')
    with pytest.raises(ValueError):
        cfg.Config(downsample_method='median')
    with pytest.raises(KeyError):
        weights.load({'input_layer.weight': np.zeros((80, 80, 3))})


def test_synthetic_data_is_deterministic():
    assert synth.splitmix64(0, 2).tolist() == \
        [16294208416658607535, 7960286522194355700]
    audio = synth.audio(9, 600)
    assert audio.shape == (1, 96000) and audio.dtype == np.float32
    assert np.all(audio[0, 32000:40000] == 0)          # the silent stretch
    assert np.array_equal(np.rint(audio * 32768) / 32768, audio)
    bounds = synth.word_frames(0, 1000)
    assert bounds[0, 0] == 0 and bounds[1, -1] == 1000
    assert np.all(bounds[0, 1:] == bounds[1, :-1])
    a = weights.random_state(cfg.DEFAULT, 7)['input_layer.weight']
    b = weights.random_state(cfg.DEFAULT, 7)['input_layer.weight']
    assert np.array_equal(a, b) and abs(float(a.mean())) < 0.01


def test_golden_audio_regenerates(cases):
    """The committed int16 audio equals what synth produces here."""
    np.testing.assert_array_equal(
        np.rint(synth.audio(0, 1000)[0] * 32768).astype(np.int16),
        cases['utt_10s/pcm'])


###############################################################################
# Alignment protocol and file I/O
###############################################################################


def test_alignment_protocol(tmp_path):
    bounds = synth.word_frames(3, 250)
    names = synth.word_names(bounds.shape[1])
    words = emphases_amd.Alignment.from_frames(bounds, names)
    assert len(words) == bounds.shape[1]
    assert words[1].start() == bounds[0, 1] / 100.
    assert abs(words[1].duration() -
               (bounds[1, 1] - bounds[0, 1]) / 100.) < 1e-12
    assert words.word_bounds(16000, 160, silences=True) == \
        [tuple(int(v) for v in pair) for pair in bounds.T]
    # a slice reports bounds relative to its first word
    assert words[2:4].word_bounds(16000, 160, silences=True)[0][0] == 0
    silent = sum(1 for n in names if n == alignment.SILENCE)
    assert len(words.word_bounds(16000, 160)) == len(words) - silent
    for suffix in ('.TextGrid', '.json'):
        file = tmp_path / f'utt{suffix}'
        words.save(file)
        loaded = emphases_amd.Alignment(file)
        assert [str(w) for w in loaded] == [str(w) for w in words]
        assert [(w.start(), w.end()) for w in loaded] == \
            [(w.start(), w.end()) for w in words]
    with pytest.raises(ValueError):
        emphases_amd.Alignment(tmp_path / 'utt.lab')


def test_textgrid_with_phone_tier(tmp_path):
    text = '''File type = "ooTextFile"
Object class = "TextGrid"

xmin = 0
xmax = 1.0
tiers? <exists>
size = 2
item []:
    item [1]:
        class = "IntervalTier"
        name = "phones"
        xmin = 0
        xmax = 1.0
        intervals: size = 3
        intervals [1]:
            xmin = 0
            xmax = 0.3
            text = "HH"
        intervals [2]:
            xmin = 0.3
            xmax = 0.5
            text = "AY"
        intervals [3]:
            xmin = 0.5
            xmax = 1.0
            text = "sp"
    item [2]:
        class = "IntervalTier"
        name = "words"
        xmin = 0
        xmax = 1.0
        intervals: size = 2
        intervals [1]:
            xmin = 0
            xmax = 0.5
            text = "hi"
        intervals [2]:
            xmin = 0.5
            xmax = 1.0
            text = ""
'''
    file = tmp_path / 'two_tier.TextGrid'
    file.write_text(text)
    words = emphases_amd.Alignment(file)
    assert [str(w) for w in words] == ['hi', alignment.SILENCE]
    assert words.word_bounds(16000, 160, silences=True) == [(0, 50), (50, 100)]
    # the phoneme tier is kept (the reference saves the alignment it loaded,
    # both tiers: emphases/core.py:105-112) ...
    assert [[str(p) for p in w.phonemes] for w in words] == \
        [['HH', 'AY'], [alignment.SILENCE]]
    assert [(p.start(), p.end()) for p in words.phonemes()] == \
        [(0., .3), (.3, .5), (.5, 1.)]
    assert words.tiers == ('words', 'phones', True)
    # ... and written back: tier names, tier order, every interval
    words.save(tmp_path / 'again.TextGrid')
    again = emphases_amd.Alignment(tmp_path / 'again.TextGrid')
    assert again.tiers == words.tiers
    assert [str(w) for w in again] == [str(w) for w in words]
    assert np.array_equal(again.times(), words.times())
    assert again.phonemes() == words.phonemes()
    first, second = (tmp_path / 'again.TextGrid').read_text().split('item [2]')
    assert 'name = "phones"' in first and 'name = "words"' in second
    assert 'size = 2' in first
    # json keeps them too
    words.save(tmp_path / 'again.json')
    assert emphases_amd.Alignment(tmp_path / 'again.json').phonemes() == \
        words.phonemes()
    # the same file in Praat's SHORT text format, and as UTF-16 (what Praat
    # writes when a label is not ASCII), with and without BOM
    short = '\n'.join([
        'File type = "ooTextFile"', 'Object class = "TextGrid"', '',
        '0', '1.0', '<exists>', '2',
        '"IntervalTier"', '"phones"', '0', '1.0', '3',
        '0', '0.3', '"HH"', '0.3', '0.5', '"AY"', '0.5', '1.0', '"sp"',
        '"IntervalTier"', '"words"', '0', '1.0', '2',
        '0', '0.5', '"hi"', '0.5', '1.0', '""', ''])
    for name, data in (
            ('short', short.encode('utf-8')),
            ('short_bom', b'\xef\xbb\xbf' + short.encode('utf-8')),
            ('long16', text.encode('utf-16')),
            ('long16be', text.encode('utf-16-be')),
            ('short16le', short.encode('utf-16-le'))):
        (tmp_path / f'{name}.TextGrid').write_bytes(data)
        other = emphases_amd.Alignment(tmp_path / f'{name}.TextGrid')
        assert [str(w) for w in other] == [str(w) for w in words], name
        assert np.array_equal(other.times(), words.times()), name
        assert other.phonemes() == words.phonemes(), name
        assert other.tiers == words.tiers, name


def test_textgrid_odd_content(tmp_path):
    """Labels with doubled quotes, digits and '=' inside, non-ASCII text,
    exponent notation, a point tier beside the interval tiers, a file that
    ends early."""
    text = '\n'.join([
        'File type = "ooTextFile"', 'Object class = "TextGrid"', '',
        'xmin = 0', 'xmax = 2.5e0', 'tiers? <exists>', 'size = 2', 'item []:',
        '    item [1]:', '        class = "TextTier"',
        '        name = "marks"', '        xmin = 0', '        xmax = 2.5',
        '        points: size = 1', '        points [1]:',
        '            number = 1.25', '            mark = "x = 3"',
        '    item [2]:', '        class = "IntervalTier"',
        '        name = "word"', '        xmin = 0', '        xmax = 2.5',
        '        intervals: size = 3',
        '        intervals [1]:', '            xmin = 0',
        '            xmax = 5e-1', '            text = "say ""2.5"" = x"',
        '        intervals [2]:', '            xmin = 0.5',
        '            xmax = 1.75', '            text = "naïve café"',
        '        intervals [3]:', '            xmin = 1.75',
        '            xmax = 2.5', '            text = "42"', ''])
    file = tmp_path / 'odd.TextGrid'
    file.write_text(text, encoding='utf-8')
    words = emphases_amd.Alignment(file)
    assert [str(w) for w in words] == [
        'say "2.5" = x', 'naïve café', '42']
    assert words.times().tolist() == [[0., .5], [.5, 1.75], [1.75, 2.5]]
    assert words.tiers[0] == 'word' and not words.phonemes()
    words.save(tmp_path / 'odd_again.TextGrid')
    again = emphases_amd.Alignment(tmp_path / 'odd_again.TextGrid')
    assert [str(w) for w in again] == [str(w) for w in words]
    assert 'name = "word"' in (tmp_path / 'odd_again.TextGrid').read_text(
        encoding='utf-8')
    (tmp_path / 'cut.TextGrid').write_text(
        text[:text.index('intervals [3]')], encoding='utf-8')
    with pytest.raises(ValueError):
        emphases_amd.Alignment(tmp_path / 'cut.TextGrid')
    (tmp_path / 'not.TextGrid').write_text('hello')
    with pytest.raises(ValueError):
        emphases_amd.Alignment(tmp_path / 'not.TextGrid')


def test_wav_roundtrip_and_resample(tmp_path):
    audio = synth.audio(2, 50)
    file = tmp_path / 'a.wav'
    load.save_wav(file, audio)
    loaded, rate = load.wav(file)
    assert rate == 16000 and np.array_equal(loaded.numpy(), audio)
    assert np.array_equal(load.audio(file).numpy(), audio)
    # the resampling TABLE is host arithmetic (the kernel that applies it is
    # `emph_resample`, tests/test_gpu_ops.py): every phase of the polyphase
    # filter passes DC with the windowed sinc's gain
    n = np.arange(8000)
    tone = torch.from_numpy(
        np.sin(2 * np.pi * 440 * n / 8000).astype(np.float32))[None]
    for rate in (8000, 22050, 44100, 48000):
        kernel, orig, new, width = load.resample_kernel(rate)
        assert kernel.shape == (new, 1, 2 * width + orig)
        gains = kernel.double().sum(dim=(1, 2))
        assert float((gains - 1).abs().max()) < 2e-3
        assert load.resampled_length(rate, orig, new) == 16000
    assert emphases_amd.resample(tone, 16000) is tone


def test_wav_headers_alone(tmp_path):
    """load.wav_info (what a sharded run plans from) agrees with the full
    reader on every format it accepts, reads no samples, and copes with odd
    chunk lists: a LIST chunk (odd size, padded) in front of the data, a
    WAVE_FORMAT_EXTENSIBLE header, a data chunk that claims more bytes than
    the file holds."""
    import struct

    def riff(fmt, data, extra=b'', claimed=None):
        chunks = b'fmt ' + struct.pack('<I', len(fmt)) + fmt + extra + \
            b'data' + struct.pack('<I', len(data) if claimed is None
                                  else claimed) + data
        return b'RIFF' + struct.pack('<I', 4 + len(chunks)) + b'WAVE' + chunks

    def pcm_format(code, channels, rate, bits, extensible=False):
        block = channels * bits // 8
        head = struct.pack('<HHIIHH', 0xFFFE if extensible else code, channels,
                           rate, rate * block, block, bits)
        if extensible:
            head += struct.pack('<HHI', 22, bits, 0) + \
                struct.pack('<H', code) + bytes(14)
        return head

    cases = {
        'pcm16_stereo': (pcm_format(1, 2, 22050, 16), 2, 22050, 16, 300),
        'pcm8': (pcm_format(1, 1, 8000, 8), 1, 8000, 8, 77),
        'pcm24': (pcm_format(1, 1, 44100, 24), 1, 44100, 24, 41),
        'float32': (pcm_format(3, 1, 16000, 32), 1, 16000, 32, 1234),
        'extensible': (pcm_format(1, 1, 48000, 16, True), 1, 48000, 16, 99),
    }
    for name, (fmt, channels, rate, bits, samples) in cases.items():
        data = bytes(range(256)) * (samples * channels * bits // 8 // 256 + 1)
        data = data[:samples * channels * bits // 8]
        file = tmp_path / f'{name}.wav'
        extra = b'LIST' + struct.pack('<I', 5) + b'abcde\0' \
            if name == 'pcm24' else b''
        file.write_bytes(riff(fmt, data, extra))
        assert load.wav_info(file) == (rate, channels, samples), name
        audio, loaded_rate = load.wav(file)
        assert loaded_rate == rate and tuple(audio.shape) == (channels, samples)
    # truncated: the data chunk claims 1000 samples, 600 are there
    file = tmp_path / 'short.wav'
    file.write_bytes(riff(pcm_format(1, 1, 16000, 16), bytes(1200), claimed=2000))
    assert load.wav_info(file) == (16000, 1, 600)
    assert load.wav(file)[0].shape == (1, 600)
    for junk in (b'RIFX' + bytes(40), riff(pcm_format(1, 1, 16000, 16), b'')[:36]):
        file = tmp_path / 'junk.wav'
        file.write_bytes(junk)
        with pytest.raises(ValueError):
            load.wav_info(file)
    # one walker behind both readers: the data chunk BEFORE the fmt chunk, two
    # data chunks (the last one counts), unsupported codes / bit depths and a
    # short fmt chunk are the same answer (or the same ValueError) from both
    fmt = pcm_format(1, 1, 16000, 16)
    first, last = bytes(range(200)), bytes(range(100, 160))
    body = b'data' + struct.pack('<I', len(first)) + first + \
        b'fmt ' + struct.pack('<I', len(fmt)) + fmt + \
        b'data' + struct.pack('<I', len(last)) + last
    file = tmp_path / 'order.wav'
    file.write_bytes(b'RIFF' + struct.pack('<I', 4 + len(body)) + b'WAVE' + body)
    assert load.wav_info(file) == (16000, 1, 30)
    audio, _ = load.wav(file, raw=True)
    assert audio.shape == (1, 30) and \
        audio.numpy().tobytes() == last
    for bad in (pcm_format(2, 1, 16000, 4), pcm_format(1, 1, 16000, 12),
                pcm_format(7, 1, 8000, 8), pcm_format(1, 1, 16000, 16)[:12],
                pcm_format(1, 0, 16000, 16)):
        file.write_bytes(riff(bad, bytes(64)))
        for reader in (load.wav_info, load.wav):
            with pytest.raises(ValueError):
                reader(file)


###############################################################################
# C ABI
###############################################################################


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'emphases_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(emph_\w+)\s*\(', header))
    assert declared == set(runtime.SIGNATURES), \
        declared ^ set(runtime.SIGNATURES)
    library = runtime.library()          # raises if any symbol is missing
    assert library.emph_abi_version() == runtime.ABI_VERSION
    assert library.emph_frontend_table_size() > 1024


def test_host_side_abi_helpers():
    table = runtime.frontend_table()
    n = np.arange(1024)
    # periodic Hann, halved (the 1/2 of the real-FFT split rides on it)
    np.testing.assert_allclose(
        2. * table[:1024], 0.5 - 0.5 * np.cos(2 * np.pi * n / 1024), atol=1e-7)
    weight = np.arange(80 * 80 * 3, dtype=np.float32).reshape(80, 80, 3)
    pack = runtime.conv_pack(weight)
    # pack[step][m][lane] = W[16 m + (lane & 15)][4 group + (lane >> 4)][tap]
    step, m, lane = 7, 3, 37
    group, tap = divmod(step, 3)
    assert pack.reshape(-1, 5, 64)[step, m, lane] == \
        weight[16 * m + (lane & 15), 4 * group + (lane >> 4), tap]
    odd = runtime.conv_pack(np.ones((1, 81, 3), dtype=np.float32))
    assert odd.sum() == 81 * 3          # padding rows/channels are zero
    library = runtime.library()
    assert library.emph_conv_pack(None, 1, 1, 1, None) == -1
    assert b'null' in library.emph_last_error()


def test_winograd_and_decoder_packs():
    """Host-side weight layouts of the Winograd conv and the word decoder,
    checked entry by entry against their definition."""
    library = runtime.library()
    weight = synth.weights(9, (80, 81, 3), 1.0)
    pack = runtime.conv_winograd_pack(weight)
    groups, mb = 21, 5
    assert pack.shape == (groups * 4 * mb * 64,)
    image = pack.reshape(1, groups, 4, mb, 64)
    wide = weight.astype(np.float64)
    transformed = [
        wide[:, :, 0], (wide[:, :, 0] + wide[:, :, 1] + wide[:, :, 2]) / 2,
        (wide[:, :, 0] - wide[:, :, 1] + wide[:, :, 2]) / 2, wide[:, :, 2]]
    for group, j, m, lane in [(0, 0, 0, 0), (3, 1, 2, 17), (20, 2, 4, 5),
                              (20, 3, 4, 63), (7, 2, 1, 48)]:
        row, column = 16 * m + (lane & 15), 4 * group + (lane >> 4)
        want = transformed[j][row, column] if column < 81 else 0.
        assert image[0, group, j, m, lane] == np.float32(want)
    assert runtime.conv_winograd_lds_bytes(80, 80) < 160 * 1024 < \
        runtime.conv_winograd_lds_bytes(80, 200)
    # 128 output channels: two m-blocks of four tiles
    assert runtime.conv_winograd_pack(
        np.ones((128, 128, 3), dtype=np.float32)).size == 2 * 32 * 4 * 4 * 64

    square = synth.weights(10, (80, 80, 3), 1.0)
    packed = runtime.word_decoder_pack(square).reshape(5, 5, 64, 4, 3)
    for trip, m, lane, t, tap in [(0, 0, 0, 0, 0), (4, 4, 63, 3, 2),
                                  (2, 1, 37, 1, 1)]:
        assert packed[trip, m, lane, t, tap] == square[
            16 * m + (lane & 15), 16 * trip + 4 * t + (lane >> 4), tap]
    linear = synth.weights(12, (80, 80), 1.0)
    natural = runtime.linear_chain_pack(linear, True).reshape(20, 5, 64)
    chained = runtime.linear_chain_pack(linear, False).reshape(20, 5, 64)
    for step, m, lane in [(0, 0, 0), (19, 4, 63), (7, 2, 21)]:
        row, k = 16 * m + (lane & 15), lane >> 4
        assert natural[step, m, lane] == linear[row, 4 * step + k]
        # k-step 4 m' + r multiplies the channels an accumulator register r of
        # m-tile m' holds: 16 m' + 4 k + r
        assert chained[step, m, lane] == linear[
            row, 16 * (step >> 2) + 4 * k + (step & 3)]
    assert library.emph_word_decoder_pack(None, 80, 3, None) == -1
    scratch = np.zeros(8, dtype=np.float32)
    assert library.emph_word_decoder_pack(
        scratch.ctypes.data, 72, 3, scratch.ctypes.data) == -2
    assert library.emph_conv_winograd_pack(None, 1, 1, None) == -1


def test_no_silent_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    bounds = synth.word_frames(0, 100)
    words = emphases_amd.Alignment.from_frames(bounds)
    with pytest.raises(runtime.LibraryError, match='no CPU fallback'):
        emphases_amd.from_alignment_and_audio(
            words, torch.zeros(1, 16000), 16000)
    with pytest.raises(NotImplementedError):
        emphases_amd.from_text_and_audio('hi', torch.zeros(1, 16000), 16000)


def test_product_does_not_import_the_oracle():
    for directory, _, files in os.walk(os.path.join(ROOT, 'emphases_amd')):
        for file in files:
            if file.endswith('.py'):
                source = open(os.path.join(directory, file)).read()
                assert not re.search(
                    r'^\s*(from|import)\s+oracle', source, flags=re.M), file


def test_word_pieces_layout():
    """DOWNSAMPLE_LOCATION='input': one padded piece per word, pooled back
    into the word's own column (`model/core.py:41-87`)."""
    bounds_a = np.array([[0, 5, 9], [5, 9, 30]], dtype=np.int64)
    bounds_b = np.array([[2], [7]], dtype=np.int64)
    plan = batch.Plan(
        [batch.Segment(0, 0, 3, 0, 0, 30, bounds_a),
         batch.Segment(1, 0, 1, 0, 0, 12, bounds_b)], [0, 0], [0, 0])
    pieces = plan.pieces('sum')
    assert pieces is plan.pieces('sum')
    assert pieces.plan.frames.tolist() == [21, 21, 21, 5]
    lengths = [5, 4, 21, 5]
    for piece, length in enumerate(lengths):
        source, count, target, padded = pieces.gather[piece]
        assert count == length and padded == pieces.plan.frames[piece]
        assert target == pieces.plan.frame_off[piece]
    assert pieces.gather[1, 0] == plan.frame_off[0] + 5
    assert pieces.gather[3, 0] == plan.frame_off[1] + 2
    columns = plan.word_columns()
    assert pieces.word_piece[columns].tolist() == [0, 1, 2, 3]
    assert (np.delete(pieces.word_piece, columns) == -1).all()
    assert pieces.bounds[:, columns].tolist() == [[0] * 4, [21, 21, 21, 5]]
    center = plan.pieces('center')
    assert center.bounds[:, columns].tolist() == [[0] * 4, lengths]
    bad = batch.Plan(
        [batch.Segment(0, 0, 1, 0, 0, 10, np.array([[4], [12]]))], [0], [0])
    with pytest.raises(ValueError):
        bad.pieces('sum')


def test_foreign_alignment_supplies_its_own_bounds():
    """An alignment object that is not ours (a real `pypar.Alignment`) keeps
    the last word on what the model sees: the reference takes the chunk's
    bounds from `alignment[start:end].word_bounds(16000, 160, silences=True)`
    (`emphases/core.py:384-392`), so the planner must call exactly that - and
    still do its own float-floor arithmetic for the chunking and the audio
    slice (`core.py:365-381,395-401`)."""
    from emphases_amd import batch

    class Word:
        def __init__(self, start, end):
            self._start, self._end = start, end

        def start(self):
            return self._start

        def end(self):
            return self._end

        def duration(self):
            return self._end - self._start

    calls = []

    class Foreign:
        """pypar-like; word_bounds deliberately differs from ours (+1 on every
        end frame) so that the test can tell who computed the bounds."""

        def __init__(self, words):
            self.words = words

        def __len__(self):
            return len(self.words)

        def __getitem__(self, index):
            if isinstance(index, slice):
                return Foreign(self.words[index])
            return self.words[index]

        def word_bounds(self, sample_rate, hopsize=1, silences=False):
            calls.append((sample_rate, hopsize, silences, len(self.words)))
            origin = int(self.words[0].start() * sample_rate / hopsize)
            return [(int(w.start() * sample_rate / hopsize) - origin,
                     int(w.end() * sample_rate / hopsize) - origin + 1)
                    for w in self.words]

    frames = synth.word_frames(5, 900, 8, 40)
    words = [Word(int(s) / 100., int(e) / 100.) for s, e in frames.T]
    ours = emphases_amd.Alignment.from_frames(frames)
    for batch_size in (None, 300):
        calls.clear()
        mine = batch.chunk_utterance(ours, 900 * 160, batch_size)
        theirs = batch.chunk_utterance(Foreign(words), 900 * 160, batch_size)
        assert len(mine) == len(theirs) == len(calls)
        assert all(call[:3] == (16000, 160, True) for call in calls)
        for a, b in zip(mine, theirs):
            assert (a.start_word, a.end_word, a.start_sample, a.length,
                    a.frames) == (b.start_word, b.end_word, b.start_sample,
                                  b.length, b.frames)
            assert np.array_equal(b.bounds[0], a.bounds[0])
            assert np.array_equal(b.bounds[1], a.bounds[1] + 1)
    plan = batch.plan_batch([Foreign(words), ours], [900 * 160] * 2)
    assert plan.words.tolist() == [frames.shape[1]] * 2
    first = plan.segment_bounds[:, :frames.shape[1]]
    second = plan.segment_bounds[:, frames.shape[1]:]
    assert np.array_equal(first[1], second[1] + 1)


def test_batch_planner_equals_per_utterance_planner():
    """plan_batch's vectorised pass against chunk_utterance + Plan, incl. an
    empty alignment, a dropped chunk and an alignment longer than its audio
    (which must fall back to real chunking)."""
    from emphases_amd import batch
    frames = [1000, 612, 37, 250, 1000, 999, 161, 16, 3]
    aligns = [emphases_amd.Alignment.from_frames(synth.word_frames(i, n, 2, 30))
              for i, n in enumerate(frames)]
    aligns.append(emphases_amd.Alignment.from_frames(
        np.zeros((2, 0), dtype=np.int64)))
    lengths = [n * 160 for n in frames] + [1600]
    # the alignment of utterance 1 describes far more audio than there is
    lengths[1] = 200 * 160
    for batch_size in (None, 400):
        fast = batch.plan_batch(aligns, lengths, batch_size)
        segments = []
        for index, (item, length) in enumerate(zip(aligns, lengths)):
            segments.extend(
                batch.chunk_utterance(item, length, batch_size, index))
        offsets = np.cumsum(lengths) - np.array(lengths)
        slow = batch.Plan(segments, offsets, lengths)
        for name in ('table', 'bounds', 'word_segment', 'frames', 'words',
                     'frame_off', 'word_off', 'utterance'):
            assert np.array_equal(getattr(fast, name), getattr(slow, name)), \
                (name, batch_size)
        assert (fast.ld_frames, fast.ld_words) == (slow.ld_frames, slow.ld_words)
        assert [s.frames for s in fast.segments] == \
            [s.frames for s in slow.segments]


def test_dropout_checkpoints_load():
    """Checkpoints of the dropout configs keep layer i at Sequential index 3 i
    (`convolution.py:25-33`, `config/hparam-search/dropout-*.py`)."""
    from emphases_amd import config as cfg
    state = weights.random_state(cfg.DEFAULT, 1)
    raw = {}
    for name, value in state.items():
        parts = name.split('.')
        if parts[0] in ('frame_encoder', 'word_decoder'):
            parts[1] = str(int(parts[1]) // 2 * 3)
        raw['.'.join(parts)] = value
    assert 'frame_encoder.3.weight' in raw
    loaded = weights.load(raw)
    assert all(np.array_equal(loaded[name], state[name]) for name in state)


def test_host_gather_and_fork():
    """emph_host_gather (the persistent pool of copy threads behind
    `session._Lane.stage`): pieces of any size and alignment land where they
    should, and a forked child - which inherits the library but not its
    worker threads - gathers with a pool of its own instead of waiting for
    the parent's."""
    import os
    from emphases_amd import runtime
    lib = runtime.library()
    generator = np.random.default_rng(3)

    def gather(threads):
        sizes = [1, 4095, 4096, 70001, 3 << 20, 17, 0, 1 << 20]
        sources = [generator.integers(0, 255, n, dtype=np.uint8) for n in sizes]
        offsets = np.cumsum([5] + [n + 3 for n in sizes[:-1]]).astype(np.int64)
        destination = np.full(int(offsets[-1]) + sizes[-1] + 9, 0xAB, np.uint8)
        pointers = np.array([s.ctypes.data for s in sources], dtype=np.int64)
        nbytes = np.array(sizes, dtype=np.int64)
        runtime.check(lib.emph_host_gather(
            pointers.ctypes.data, nbytes.ctypes.data, offsets.ctypes.data,
            len(sizes), destination.ctypes.data, threads), 'emph_host_gather')
        expect = np.full_like(destination, 0xAB)
        for source, offset in zip(sources, offsets):
            expect[offset:offset + source.size] = source
        return bool(np.array_equal(destination, expect))

    assert gather(1) and gather(6) and gather(3)
    assert lib.emph_host_gather(None, None, None, 1, None, 4) != 0
    pid = os.fork()
    if pid == 0:                       # the child: must not hang or crash
        os._exit(0 if gather(4) and gather(2) else 1)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    assert gather(5)


###############################################################################
# The batched file boundary (csrc/files.hip) against alignment.py / load.py
###############################################################################


def _grid_variants(tmp_path):
    """TextGrid files in every form the readers accept -> list of paths."""
    long_two = '\n'.join([
        'File type = "ooTextFile"', 'Object class = "TextGrid"', '',
        'xmin = 0', 'xmax = 1.0', 'tiers? <exists>', 'size = 2', 'item []:',
        '    item [1]:', '        class = "IntervalTier"',
        '        name = "phones"', '        xmin = 0', '        xmax = 1.0',
        '        intervals: size = 3',
        '        intervals [1]:', '            xmin = 0',
        '            xmax = 0.3', '            text = "HH"',
        '        intervals [2]:', '            xmin = 0.3',
        '            xmax = 0.5', '            text = "AY"',
        '        intervals [3]:', '            xmin = 0.5',
        '            xmax = 1.0', '            text = "sp"',
        '    item [2]:', '        class = "IntervalTier"',
        '        name = "words"', '        xmin = 0', '        xmax = 1.0',
        '        intervals: size = 2',
        '        intervals [1]:', '            xmin = 0',
        '            xmax = 0.5', '            text = "hi"',
        '        intervals [2]:', '            xmin = 0.5',
        '            xmax = 1.0', '            text = ""', ''])
    short = '\n'.join([
        'File type = "ooTextFile"', 'Object class = "TextGrid"', '',
        '0', '1.0', '<exists>', '2',
        '"IntervalTier"', '"phones"', '0', '1.0', '3',
        '0', '0.3', '"HH"', '0.3', '0.5', '"AY"', '0.5', '1.0', '"sp"',
        '"IntervalTier"', '"words"', '0', '1.0', '2',
        '0', '0.5', '"hi"', '0.5', '1.0', '""', ''])
    odd = '\n'.join([
        'File type = "ooTextFile"', 'Object class = "TextGrid"', '',
        'xmin = 0', 'xmax = 2.5e0', 'tiers? <exists>', 'size = 2', 'item []:',
        '    item [1]:', '        class = "TextTier"',
        '        name = "marks"', '        xmin = 0', '        xmax = 2.5',
        '        points: size = 1', '        points [1]:',
        '            number = 1.25', '            mark = "x = 3"',
        '    item [2]:', '        class = "IntervalTier"',
        '        name = "word"', '        xmin = 0', '        xmax = 2.5',
        '        intervals: size = 3',
        '        intervals [1]:', '            xmin = 0.00001',
        '            xmax = 5e-1', '            text = "say ""2.5"" = x"',
        '        intervals [2]:', '            xmin = 0.75',
        '            xmax = 1.7500000000000002',
        '            text = "naïve café"',
        '        intervals [3]:', '            xmin = 1.7500000000000002',
        '            xmax = 2.5', '            text = "42"', ''])
    # words and a finer unnamed tier, in that order; a gap between two words
    gaps = '\n'.join([
        'File type = "ooTextFile"', 'Object class = "TextGrid"', '',
        '0', '3', '<exists>', '2',
        '"IntervalTier"', '"utterance"', '0', '3', '2',
        '0.25', '1', '"a b"', '1.5', '3', '"c"',
        '"IntervalTier"', '"segments"', '0', '3', '4',
        '0.25', '0.5', '"a"', '0.5', '1', '"b"', '1.5', '2', '"c1"',
        '2', '3', '"c2"', ''])
    variants = {
        'long': long_two.encode(), 'short': short.encode(),
        'short_bom': b'\xef\xbb\xbf' + short.encode(),
        'long16': long_two.encode('utf-16'),
        'long16be': long_two.encode('utf-16-be'),
        'short16le': short.encode('utf-16-le'),
        'odd': odd.encode('utf-8'), 'odd16': odd.encode('utf-16'),
        'gaps': gaps.encode()}
    paths = []
    for name, data in variants.items():
        path = tmp_path / f'{name}.TextGrid'
        path.write_bytes(data)
        paths.append(path)
    words = emphases_amd.Alignment.from_frames(
        synth.word_frames(3, 700), synth.word_names(
            synth.word_frames(3, 700).shape[1]))
    words.save(tmp_path / 'synthetic.TextGrid')
    paths.append(tmp_path / 'synthetic.TextGrid')
    return paths


def test_file_batch_matches_the_python_readers(tmp_path):
    """csrc/files.hip (the batched file boundary of from_files_to_files)
    against alignment.py and load.py, file by file: word and phoneme labels
    and times, tier names and order, gap filling; WAVE headers over odd chunk
    lists; the samples it reads; and the files it writes - the TextGrid byte
    for byte what `Alignment.save` writes, the .pt what `torch.save` would
    hold (`torch.load` gives the same float32 tensor)."""
    from emphases_amd import files
    grids = _grid_variants(tmp_path)
    waves = []
    for index, grid in enumerate(grids):
        wave = tmp_path / f'{grid.stem}.wav'
        samples = 400 + 37 * index
        audio = synth.weights(50 + index, (1, samples), 0.5)
        if index % 3 == 2:
            # float32 body, a LIST chunk (odd size) in front of it
            body = audio.astype('<f4').tobytes()
            fmt = struct.pack('<HHIIHH', 3, 1, 22050, 22050 * 4, 4, 32)
            extra = b'LIST' + struct.pack('<I', 5) + b'abcde' + b'\0'
            chunks = b'fmt ' + struct.pack('<I', 16) + fmt + extra + \
                b'data' + struct.pack('<I', len(body)) + body
            wave.write_bytes(b'RIFF' + struct.pack('<I', 4 + len(chunks)) +
                             b'WAVE' + chunks)
        else:
            load.save_wav(wave, audio, 16000 if index % 3 == 0 else 8000)
        waves.append(wave)
    opened = files.FileBatch(grids, waves, threads=4)
    assert not opened.status.any(), [opened.error(i) for i in range(len(grids))]
    total = 0
    for index, (grid, wave) in enumerate(zip(grids, waves)):
        want = emphases_amd.Alignment(grid)
        got = opened.alignment(index)
        assert len(got) == len(want), grid.name
        assert np.array_equal(got.times(), want.times()), grid.name
        assert got.word_bounds(16000, 160, silences=True) == \
            want.word_bounds(16000, 160, silences=True)
        assert [str(w) for w in got] == [str(w) for w in want], grid.name
        assert got.tiers == want.tiers, grid.name
        assert got.phonemes() == want.phonemes(), grid.name
        assert [len(w.phonemes or []) for w in got] == \
            [len(w.phonemes or []) for w in want], grid.name
        # audio: headers and samples
        samples, rate = load.wav(wave, raw=True)
        audio, got_rate = opened.audio(index)
        assert got_rate == rate and isinstance(audio, files.FileAudio)
        assert audio.shape == (samples.shape[1],) and \
            audio.dtype == samples.dtype
        buffer = np.zeros(samples.shape[1] + 8, dtype=samples.numpy().dtype)
        opened.read([index], [16], [samples.shape[1] * buffer.itemsize],
                    buffer.ctypes.data)
        assert np.array_equal(buffer[16 // buffer.itemsize:][
            :samples.shape[1]], samples[0].numpy()[
                :len(buffer) - 16 // buffer.itemsize])
        total += len(want)
    assert total == len(opened.times)
    # outputs
    scores = [torch.from_numpy(synth.weights(
        90 + i, (1, len(opened.alignment(i))), 1.)) for i in range(len(grids))]
    scores[1] = torch.zeros(1, len(opened.alignment(1)))
    prefixes = [tmp_path / f'out_{grid.stem}' for grid in grids]
    opened.write(list(range(len(grids))), prefixes, scores)
    for grid, prefix, item in zip(grids, prefixes, scores):
        loaded = torch.load(f'{prefix}.pt')
        assert loaded.dtype == torch.float32 and torch.equal(loaded, item)
        emphases_amd.Alignment(grid).save(tmp_path / 'python.TextGrid')
        assert open(f'{prefix}.TextGrid', 'rb').read() == \
            (tmp_path / 'python.TextGrid').read_bytes(), grid.name
    # a long score vector (pickle integer widths), an empty one
    for count in (0, 255, 256, 70000):
        item = torch.arange(count, dtype=torch.float32)[None] / 7
        opened.write([0], [tmp_path / f'n{count}'], [item])
        assert torch.equal(torch.load(tmp_path / f'n{count}.pt'), item)


def test_file_batch_reports_bad_files_like_the_python_readers(tmp_path):
    """A file the library cannot take is a per-file status, and asking for it
    raises what alignment.py / load.py raise."""
    from emphases_amd import files
    good = _grid_variants(tmp_path)[0]
    load.save_wav(tmp_path / 'good.wav', synth.weights(1, (1, 300), .3))
    (tmp_path / 'bad.TextGrid').write_text('hello')
    (tmp_path / 'cut.TextGrid').write_bytes(good.read_bytes()[:300])
    (tmp_path / 'bad.wav').write_bytes(b'RIFX' + bytes(40))
    words = emphases_amd.Alignment(good)
    words.save(tmp_path / 'words.json')
    stereo = synth.weights(2, (2, 200), .3)
    load.save_wav(tmp_path / 'stereo.wav', stereo)
    texts = [good, tmp_path / 'bad.TextGrid', tmp_path / 'cut.TextGrid',
             tmp_path / 'missing.TextGrid', tmp_path / 'words.json', good]
    waves = [tmp_path / 'good.wav', tmp_path / 'good.wav',
             tmp_path / 'good.wav', tmp_path / 'good.wav',
             tmp_path / 'stereo.wav', tmp_path / 'bad.wav']
    opened = files.FileBatch(texts, waves)
    assert opened.status.tolist() == [0, 1, 1, 1, 1, 2]
    assert 'TextGrid' in opened.error(1) and opened.error(0) == ''
    for index in (1, 2):
        with pytest.raises(ValueError):
            opened.alignment(index)
    with pytest.raises(OSError):
        opened.alignment(3)
    # JSON alignments and multi-channel audio take the Python readers
    assert [str(w) for w in opened.alignment(4)] == [str(w) for w in words]
    audio, rate = opened.audio(4)
    assert torch.is_tensor(audio) and audio.shape == (2, 200)
    with pytest.raises(ValueError):
        opened.audio(5)


def test_textgrid_numbers_do_not_follow_the_locale_or_overflow(tmp_path):
    """The library reads numbers with std::from_chars: `strtod` follows the
    process's LC_NUMERIC (under a comma-decimal locale '0.5' reads as 0) and
    Python's float(), the oracle, never does; a time no double holds (1e999)
    is not vouched for - the Python reader gets the file - instead of being
    written back as 'i.nf'."""
    import ctypes
    from emphases_amd import files
    good = _grid_variants(tmp_path)[0]
    load.save_wav(tmp_path / 'good.wav', synth.weights(1, (1, 300), .3))
    text = good.read_text()
    # every available comma-decimal locale (the image may have none: then the
    # library's parse is at least shown not to call the C locale machinery by
    # giving the same times under LC_NUMERIC='' as under 'C')
    libc = ctypes.CDLL(None)
    libc.setlocale.restype = ctypes.c_char_p
    LC_NUMERIC = 1
    before = libc.setlocale(LC_NUMERIC, None)
    want = files.FileBatch([good], [tmp_path / 'good.wav']).times.copy()
    assert want.size and np.any(want != np.floor(want))       # (fractions in there)
    try:
        for name in (b'de_DE.UTF-8', b'de_DE.utf8', b'fr_FR.UTF-8', b'de_DE', b''):
            if libc.setlocale(LC_NUMERIC, name) is None:
                continue
            got = files.FileBatch([good], [tmp_path / 'good.wav']).times
            assert np.array_equal(got, want), name
    finally:
        libc.setlocale(LC_NUMERIC, before)
    # a time beyond the doubles
    huge = tmp_path / 'huge.TextGrid'
    huge.write_text(text.replace(text.split('xmax = ')[1].split('\n')[0], '1e999', 1))
    opened = files.FileBatch([huge, good], [tmp_path / 'good.wav'] * 2)
    assert opened.status.tolist()[0] & 1 and opened.status.tolist()[1] == 0
    plus = tmp_path / 'plus.TextGrid'
    plus.write_text(text.replace('xmin = 0', 'xmin = +0', 1))
    assert files.FileBatch([plus], [tmp_path / 'good.wav']).status.tolist() == [0]


def test_file_batch_reads_a_whole_batch_without_an_object_per_file(tmp_path):
    """`FileBatch.staged_format` / `read_staged`: when every file is 16 kHz mono
    in one sample format, the samples of all of them go back to back (4-byte
    aligned) into one buffer and come with (byte offsets, sample counts) - the
    bytes are `load.wav`'s."""
    import torch
    from emphases_amd import files, load
    texts, waves, counts = [], [], [16000, 4801, 3199, 8003, 1600]
    for index, samples in enumerate(counts):
        emphases_amd.Alignment.from_frames(
            synth.word_frames(index, samples // 160, 3, 20)).save(
                tmp_path / f'u{index}.TextGrid')
        audio = np.resize(synth.audio(index, 100), samples)
        load.save_wav(tmp_path / f'u{index}.wav', audio)
        texts.append(tmp_path / f'u{index}.TextGrid')
        waves.append(tmp_path / f'u{index}.wav')
    opened = files.FileBatch(texts, waves, 2)
    assert opened.staged_format(16000) == torch.int16
    assert opened.staged_format(22050) is None
    staging = torch.zeros(opened.audio_bytes() + 64, dtype=torch.uint8)
    where, lengths = opened.read_staged(staging)
    assert lengths.tolist() == counts
    assert np.all(where % 4 == 0) and where[0] == 0
    assert where.tolist() == np.concatenate(
        [[0], np.cumsum([(2 * n + 3) // 4 * 4 for n in counts])[:-1]]).tolist()
    for index, (start, n) in enumerate(zip(where.tolist(), counts)):
        got = staging[start:start + 2 * n].view(torch.int16)
        want, rate = load.wav(waves[index], raw=True)
        assert rate == 16000 and torch.equal(got, want[0])
    with pytest.raises(ValueError, match='too small'):
        opened.read_staged(torch.zeros(100, dtype=torch.uint8))
    # a stereo file, or one the library does not vouch for: not this way
    load.save_wav(tmp_path / 'stereo.wav', np.zeros((2, 800), dtype=np.float32))
    assert files.FileBatch(
        texts[:2], [waves[0], tmp_path / 'stereo.wav']).staged_format(16000) is None
    assert files.FileBatch([], []).staged_format(16000) is None


def test_file_pool_threads_can_be_moved_next_to_the_gpu(tmp_path):
    """emph_files_affinity moves the library's OWN pool threads (the ones there are
    and the ones to come), never the caller: what `core.files_to_scores` does with
    the CPUs of the GPU's NUMA node (`files.cpus_near`, None without a GPU)."""
    from emphases_amd import files, load
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 3:
        pytest.skip('needs three CPUs')
    assert files.cpus_near(0) is None            # (no GPU here: nothing to be near)
    texts, waves = [], []
    for index in range(12):
        emphases_amd.Alignment.from_frames(
            synth.word_frames(index, 50, 3, 20)).save(tmp_path / f'u{index}.TextGrid')
        load.save_wav(tmp_path / f'u{index}.wav', synth.audio(index, 50))
        texts.append(tmp_path / f'u{index}.TextGrid')
        waves.append(tmp_path / f'u{index}.wav')

    def affinities():
        result = {}
        for task in os.listdir('/proc/self/task'):
            try:
                result[int(task)] = sorted(os.sched_getaffinity(int(task)))
            except OSError:
                pass
        return result

    files.FileBatch(texts, waves, 4).close()              # the pool has threads now
    before = affinities()
    try:
        files.pool_near(allowed[:2])
        files.FileBatch(texts, waves, 6).close()          # ... and grows pinned
        after = affinities()
        moved = [task for task, cpus in after.items() if cpus == allowed[:2]]
        assert len(moved) >= 5                            # helpers: 3 old + 2 new
        assert after[os.getpid()] == allowed              # the caller stays
        assert all(after[task] == cpus for task, cpus in before.items()
                   if task in after and task not in moved)
        with pytest.raises(runtime.LibraryError):
            files.pool_near([])
        with pytest.raises(runtime.LibraryError):
            files.pool_near([1 << 20])
    finally:
        files.pool_near(allowed)


def test_plan_tables_of_the_library_match_numpy():
    """emph_plan_tiles / emph_plan_word_sums (host arithmetic in the library,
    what `batch.Plan` calls) against the numpy restatement in
    tests/plan_reference.py, bit for bit, on ragged plans whose bounds stress
    the tables: empty words, ends and starts beyond the chunk, overlapping
    words, words over many restarts, one-frame segments; both restart schemes
    (64-frame tiles, the conv stack's spans); filtered tile tables."""
    import plan_reference
    from emphases_amd import runtime
    rng = np.random.default_rng(11)
    frames = [1000, 37, 130, 64, 65, 1, 447, 256, 257, 3000, 505]
    segments = []
    for index, n in enumerate(frames):
        starts = np.sort(rng.integers(0, n + 1, size=max(2, n // 9)))
        ends = np.minimum(starts + rng.integers(0, 300, size=starts.size), n + 40)
        ends[0] = starts[0]
        starts[-1], ends[-1] = n + 3, n + 9
        bounds = np.stack([starts, ends]).astype(np.int64)
        segments.append(batch.Segment(index, 0, bounds.shape[1], 432, n * 160,
                                      n, bounds))
    plan = batch.Plan(segments, [0] * len(frames), [0] * len(frames))
    for block in (8, 16, 64, 256):
        assert np.array_equal(
            plan.tiles(runtime.AXIS_FRAMES, block),
            plan_reference.tiles(plan.frames, plan.frame_off, block))
        whole = plan_reference.tiles(plan.words, plan.word_off, block)
        assert np.array_equal(plan.tiles(runtime.AXIS_WORDS, block), whole)
        keep = (whole[:, 3] >= 5) & (whole[:, 3] <= 60)
        assert np.array_equal(
            plan.tiles(runtime.AXIS_WORDS, block, 5, 60), whole[keep])
    for restarts in (plan.sum_restarts(),
                     plan.sum_restarts(plan.conv_spans())):
        got = plan.word_sum_tables(restarts)
        want = plan_reference.word_sum_tables(plan, restarts)
        assert got['n_slots'] == want['n_slots']
        for name in ('slot_map', 'terms', 'first', 'lengths'):
            assert got[name].dtype == np.int32
            assert np.array_equal(got[name], want[name]), name
    empty = batch.Plan([], [], [])
    assert empty.word_sum_tables()['n_slots'] == 0


def test_plan_batch_of_the_library_matches_numpy_and_the_chunker():
    """emph_plan_batch (the chunk of every utterance of a batch, in the library)
    against the numpy restatement in tests/plan_reference.py bit for bit, and
    `batch.plan_batch` over it against the per-utterance chunker
    (`chunk_utterance`, emphases/core.py:345-418): word times on and around the
    hop grid (where x // 160 and floor(x / 160) part ways), empty alignments,
    utterances too short for a chunk, words past the audio's end, gaps and
    overlaps; "plan slowly" for several chunks, negative durations, NaN."""
    import plan_reference
    rng = np.random.default_rng(5)
    tables, lengths = [], []
    for index in range(300):
        count = int(rng.integers(0, 40)) if index % 17 else 0
        kind = index % 5
        if kind == 0:       # a multiple of the hop, plus or minus an ulp
            edges = np.sort(rng.integers(0, 1500, size=count + 1)) * 0.01
            edges = np.nextafter(edges, rng.choice([-np.inf, np.inf], size=edges.shape))
        elif kind == 1:     # the decimal times of a TextGrid
            edges = np.round(np.sort(rng.uniform(0, 12, size=count + 1)), 2)
        elif kind == 2:     # exactly on the hop grid
            edges = np.sort(rng.integers(0, 1200, size=count + 1)) * 0.01
        else:
            edges = np.sort(rng.uniform(0, 10, size=count + 1))
        times = np.stack([edges[:-1], edges[1:]], axis=1) if count else \
            np.zeros((0, 2))
        if kind == 3 and count > 2:     # a gap and an overlap
            times[1, 1] -= min(0.004, (times[1, 1] - times[1, 0]) / 2)
            times[2, 0] -= 0.008
        if kind == 4 and count:         # too short for a chunk
            times = times * 0.002
        if kind == 2 and count:         # the last word runs past the audio
            times[-1, 1] += 3.0
        tables.append(np.ascontiguousarray(times, dtype=np.float64))
        end = float(edges[-1]) * (0.002 if kind == 4 else 1.)
        lengths.append(int(end * 16000) + int(rng.choice([0, 1, 159, 160, 433, 5000])))
    lengths = np.array(lengths, dtype=np.int64)
    counts = np.array([len(t) for t in tables], dtype=np.int64)
    table = np.concatenate(tables)
    want = plan_reference.plan_columns(table, counts, lengths)
    got = batch._plan_columns(table, counts, lengths)
    assert want is not None and got is not None
    assert len(want[0]) > 200
    for a, b in zip(got, want):
        assert a.dtype == np.int64 and np.array_equal(a, b)
    # the Plan over it is the Plan of the per-utterance chunker
    fast = batch.plan_batch(tables, lengths)
    also = batch.plan_batch(None, lengths, tables=(table, counts))
    segments = []
    for index, (times, length) in enumerate(zip(tables, lengths)):
        segments.extend(batch.chunk_utterance(times, int(length), None, index))
    slow = batch.Plan(segments, np.cumsum(lengths) - lengths, lengths)
    for plan in (fast, also):
        for name in ('utterance', 'start_word', 'frames', 'words', 'frame_off',
                     'word_off', 'table', 'bounds', 'word_segment',
                     'segment_bounds'):
            assert np.array_equal(getattr(plan, name), getattr(slow, name)), name
        assert (plan.ld_frames, plan.ld_words) == (slow.ld_frames, slow.ld_words)
    # what the one pass must hand back to the chunker
    for spoil in ('chunks', 'negative', 'nan', 'huge'):
        changed = [t.copy() for t in tables]
        big = int(np.argmax(counts))
        if spoil == 'huge':
            # finite, but its frame index is beyond 2^52 (a corrupt TextGrid): the
            # library must not cast it (undefined behaviour; UBSan float-cast-overflow)
            changed[big][-1, 1] = 1e300
            spoiled = np.concatenate(changed)
            assert batch._plan_columns(spoiled, counts, lengths) is None
            continue
        if spoil == 'chunks':       # words that outlast the audio's frames
            changed[big][:, 1] += 30.0
            changed[big][1:, 0] += 30.0
        elif spoil == 'negative':
            changed[big][3, 1] = changed[big][3, 0] - 0.5
        else:
            changed[big][2, 0] = np.nan
        spoiled = np.concatenate(changed)
        assert batch._plan_columns(spoiled, counts, lengths) is None
        assert plan_reference.plan_columns(spoiled, counts, lengths) is None
        if spoil == 'chunks':
            # the caller that holds one table gets the chunker's plan all the same:
            # the list of alignments is built only now (`files.FileBatch.all_times`)
            asked = []
            lazily = batch.plan_batch(
                lambda: asked.append(1) or changed, lengths,
                tables=(spoiled, counts))
            listed = batch.plan_batch(changed, lengths)
            assert asked == [1] and len(lazily) == len(listed)
            assert np.array_equal(lazily.table, listed.table)
            assert np.array_equal(lazily.bounds, listed.bounds)
    with pytest.raises(ValueError, match='disagree'):
        batch._plan_columns(table[:-1], counts, lengths)
    # ... and pack_metadata's tables built in place are word_sum_tables' own
    packed, offsets = fast.pack_metadata(
        [(0, 64)], word_sums=True, spans=True, sum_step=32)
    fresh = batch.plan_batch(tables, lengths)
    alone = fresh.word_sum_tables(fresh.stack_restarts(32))
    for name in ('slot_map', 'terms', 'first', 'lengths'):
        start, size = offsets[('word_sums', name)]
        assert np.array_equal(packed[start:start + size], alone[name]), name
    assert packed.size == sum(
        -(-max(size, 1) // 4) * 4 for _, size in offsets.values())


def test_hand_scheduled_loads_are_not_touched_in_flight():
    """conv_stack.hip's loader waves issue their row loads from inline asm and wait with
    explicit s_waitcnt; hipcc does not know those registers are still being written.  The
    checker replays the kernel's ISA and fails on any instruction that touches them early."""
    import shutil
    import subprocess
    import sys
    if shutil.which('hipcc') is None:
        pytest.skip('hipcc not on PATH')
    done = subprocess.run([sys.executable, os.path.join(str(ROOT), 'tools', 'check_inflight.py')],
                          capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout + done.stderr
    assert re.search(r'[1-9]\d* inline-asm loads checked, 0 problem', done.stdout), done.stdout


def test_file_readers_differential_fuzz(tmp_path):
    """Mutated TextGrid and WAVE files (flipped bytes, cuts, doubled or dropped
    slices, stray quotes and numbers, swapped lines, rewritten header fields)
    through the library's readers and through alignment.py / load.py: the
    library never crashes, and whatever it accepts is what the Python reader
    makes of the same bytes (a file it does not vouch for goes to the Python
    reader).  tests/fuzz_files.py runs more cases from the command line; the
    round ran 27 000 without a disagreement after the fixes it prompted
    (regex backtracking over an unclosed string, strict UTF-8 / UTF-16,
    counts beyond int, Unicode blanks, a line break in a tier name)."""
    import fuzz_files
    grids, waves = fuzz_files.corpus(str(tmp_path))
    problems = fuzz_files.run(str(tmp_path), grids, waves, 400, seed=7)
    assert not problems, [problem[:4] for problem in problems[:5]]
    assert not fuzz_files.run_writer(str(tmp_path), 150, seed=7)
    # the cases the fuzzer found, by hand
    cases = {
        'pair.TextGrid': open(grids[0], 'rb').read().replace(
            b'text = "hi"\n', b'text = "hi"\n"', 1),
        'utf8.TextGrid': open(grids[0], 'rb').read().replace(b'xmax', b'xm\xddx', 1),
        'count.TextGrid': open(grids[1], 'rb').read().replace(
            b'"words"\n0\n1.0\n2\n', b'"words"\n0\n1.0\n' + b'9' * 31 + b'\n', 1),
        'name.TextGrid': open(grids[1], 'rb').read().replace(b'"words"', b'"wor\nds"', 1),
        'odd16.TextGrid': open(grids[3], 'rb').read()[:-1],
        'dot.TextGrid': open(grids[1], 'rb').read().replace(b'\n0.3\n', b'\n0.3x\n', 1),
    }
    texts = []
    for name, data in cases.items():
        (tmp_path / name).write_bytes(data)
        texts.append(tmp_path / name)
    from emphases_amd import files
    opened = files.FileBatch(texts, [waves[0]] * len(texts))
    for index, text in enumerate(texts):
        assert fuzz_files.python_alignment(text) == \
            fuzz_files.library_alignment(opened, index), text.name


def test_conv_span_table_invariants():
    """emph_conv_stack_spans for every segment length up to 2100 and a few long
    ones: the spans of a segment tile it without gap or overlap from quad
    boundaries, each fits the 256 computed positions of a workgroup together
    with the quad of halo it needs on every side that continues inside the
    segment, and there are as few of them as those capacities allow."""
    counts = np.array(list(range(1, 2101)) + [2999, 3000, 3001, 30000, 30001],
                      dtype=np.int64)
    plan = batch.Plan([], [], [])
    plan.frames = counts
    plan.frame_off = np.concatenate([[16], 16 + np.cumsum(counts + 24)[:-1]])
    plan._tiles = {}
    spans = plan.conv_spans()
    assert spans.dtype == np.int32 and spans.shape[1] == 8
    for segment, count in enumerate(counts.tolist()):
        rows = spans[spans[:, 0] == segment]
        assert (rows[:, 2] == plan.frame_off[segment]).all()
        assert (rows[:, 3] == count).all() and not rows[:, 6:].any()
        first, owned, computed = rows[:, 1], rows[:, 4], rows[:, 5]
        assert first[0] == 0 and (first[1:] == first[:-1] + owned[:-1]).all()
        assert owned.sum() == count and (owned > 0).all()
        assert not (first % 4).any() and not (computed % 4).any()
        assert (computed == np.maximum(first - 4, 0)).all()
        end = first + owned
        need = np.where(end < count, end + 4, end)      # a quad of halo inside
        assert (need <= computed + 256).all(), count
        pieces = 1 if count <= 256 else max(2, 2 + -(-(count - 504) // 248))
        assert len(rows) == pieces, (count, len(rows), pieces)


def test_package_never_switches_torch_threads():
    """`torch.set_num_threads` is process-global: a drop-in library must not
    call it (round 4 did, around every API call)."""
    import glob
    package = os.path.join(ROOT, 'emphases_amd')
    for path in glob.glob(os.path.join(package, '**', '*.py'), recursive=True):
        with open(path) as file:
            for number, line in enumerate(file, 1):
                code = line.split('#')[0]
                assert 'set_num_threads(' not in code, (path, number)
    from emphases_amd import runtime
    assert not hasattr(runtime, 'few_host_threads')


def test_file_pipeline_threads_fit_the_cpu_budget(monkeypatch):
    from emphases_amd import files
    for budget in (1, 2, 4, 8, 16, 64, 256):
        monkeypatch.setattr(files, '_cpu_budget', lambda budget=budget: budget)
        monkeypatch.delenv('EMPHASES_FILE_THREADS', raising=False)
        opening, writing = files.stage_threads()
        assert opening >= 1 and writing >= 1 and opening >= writing
        # opener + writer pools, the calling thread and HIP's own threads
        assert opening + writing + 4 <= max(budget, 6), (budget, opening, writing)


def test_torch_library_ops_are_registered_and_refuse_the_cpu():
    """SURVEY 8b's operator seams exist as torch.ops.emphases_amd.* with the
    varlen signatures, and there is no CPU implementation behind them."""
    import emphases_amd  # noqa: F401
    from emphases_amd import runtime
    ops = torch.ops.emphases_amd
    assert str(ops.logmel.default._schema) == \
        'emphases_amd::logmel(Tensor audio_packed, Tensor cu_samples) -> Tensor'
    assert 'Tensor cu_T, str activation' in str(ops.conv1d_same_act.default._schema)
    assert 'Tensor cu_frames, Tensor cu_words, str mode' in \
        str(ops.segment_reduce.default._schema)
    assert 'Tensor cu_T, SymInt heads' in str(ops.encoder_layer.default._schema)
    assert 'Tensor bounds, Tensor cu_words' in \
        str(ops.prominence_forward.default._schema)
    edges = torch.tensor([0, 16000], dtype=torch.int32)
    with pytest.raises(runtime.LibraryError, match='no CPU'):
        ops.logmel(torch.zeros(16000), edges)
    with pytest.raises(runtime.LibraryError, match='no CPU'):
        ops.segment_reduce(torch.zeros(80, 10), torch.zeros(2, 1), edges, edges, 'sum')


def test_conv_split_pack_is_two_bf16_pieces_in_fragment_order():
    """emph_conv_split_pack (host): every weight as two bf16 pieces whose sum is
    within 2^-17 of it, laid out [tap][block][m-tile][piece][lane][8] with lane =
    (output channel 32 m + lane % 32, input channels 16 block + 8 (lane / 32) ..);
    rows 80 .. 95 are zeros."""
    from emphases_amd import runtime
    lib = runtime.library()
    weight = synth.weights(77, (80, 80, 3), 0.3)
    pack = runtime.conv_split_pack(weight)
    assert pack.nbytes == lib.emph_conv_split_pack_size() == 3 * 5 * 3 * 2 * 1024
    halves = pack.view(np.uint16).reshape(3, 5, 3, 2, 64, 8)
    values = (halves.astype(np.uint32) << 16).view(np.float32)
    rebuilt = np.zeros((96, 80, 3), dtype=np.float64)
    for tap in range(3):
        for block in range(5):
            for m in range(3):
                for lane in range(64):
                    row = 32 * m + lane % 32
                    channels = 16 * block + 8 * (lane // 32) + np.arange(8)
                    rebuilt[row, channels, tap] = \
                        values[tap, block, m, 0, lane].astype(np.float64) + \
                        values[tap, block, m, 1, lane]
    assert np.all(rebuilt[80:] == 0.)
    error = np.abs(rebuilt[:80] - weight)
    assert float((error / np.maximum(np.abs(weight), 1e-30)).max()) < 2.0 ** -16
    # the leading piece is the weight rounded to nearest (even) bf16
    bits = weight.view(np.uint32)
    nearest = ((bits + 0x7fff + ((bits >> 16) & 1)) >> 16).astype(np.uint16)
    lane, m, block, tap = 37, 1, 3, 2
    row, channels = 32 * m + lane % 32, 16 * block + 8 * (lane // 32) + np.arange(8)
    assert np.array_equal(halves[tap, block, m, 0, lane], nearest[row, channels, tap])


def test_linear_split_pack_pieces_in_chain_order():
    """emph_linear_split_pack (host): a [80, 80] Linear weight as two (rounded) or
    three (truncated: exact) bf16 pieces, [k-step][m-tile][piece][lane][8] with
    lane = (output channel 32 m + lane % 32; input channels 16 j + 8 (e / 4) +
    4 (lane / 32) + e % 4 - the order in which one GEMM's result lies in the
    registers of the next); rows 80 .. 95 are zeros."""
    from emphases_amd import runtime
    lib = runtime.library()
    weight = synth.weights(78, (80, 80), 0.3)
    assert lib.emph_linear_split_pack_size(4) == 0
    for pieces in (2, 3):
        pack = runtime.linear_split_pack(weight, pieces)
        assert pack.nbytes == lib.emph_linear_split_pack_size(pieces) == \
            5 * 3 * pieces * 1024
        halves = pack.view(np.uint16).reshape(5, 3, pieces, 64, 8)
        values = (halves.astype(np.uint32) << 16).view(np.float32)
        rebuilt = np.zeros((96, 80), dtype=np.float64)
        e = np.arange(8)
        for j in range(5):
            for m in range(3):
                for lane in range(64):
                    channels = 16 * j + 8 * (e // 4) + 4 * (lane // 32) + e % 4
                    rebuilt[32 * m + lane % 32, channels] = \
                        values[j, m, :, lane].astype(np.float64).sum(0)
        assert np.all(rebuilt[80:] == 0.)
        error = np.abs(rebuilt[:80] - weight) / np.maximum(np.abs(weight), 1e-30)
        if pieces == 3:
            assert float(error.max()) == 0.         # 8 + 8 + 8 bits: all of fp32's
            leading = (weight.view(np.uint32) >> 16).astype(np.uint16)
        else:
            assert float(error.max()) < 2.0 ** -16
            bits = weight.view(np.uint32)
            leading = ((bits + 0x7fff + ((bits >> 16) & 1)) >> 16).astype(
                np.uint16)
        lane, m, j = 37, 1, 3
        channels = 16 * j + 8 * (e // 4) + 4 * (lane // 32) + e % 4
        assert np.array_equal(halves[j, m, 0, lane],
                              leading[32 * m + lane % 32, channels])


def test_linear_split_pack16_pieces_in_chain_order():
    """emph_linear_split_pack16 (host; the kernels on tiles of 16 positions):
    [k-step j][m-tile m][piece][lane][8] with lane = (output channel 16 m + lane % 16;
    input channels 32 j + 16 (e / 4) + 4 (lane / 16) + e % 4 - the order in which a
    16 x 16 result lies in the registers of the next GEMM); five m-tiles (no padded
    rows), three k-steps whose input channels 80 .. 95 are zeros."""
    from emphases_amd import runtime
    lib = runtime.library()
    weight = synth.weights(78, (80, 80), 0.3)
    assert lib.emph_linear_split_pack16_size(4) == 0
    e = np.arange(8)
    for pieces in (2, 3):
        pack = runtime.linear_split_pack(weight, pieces, 16)
        assert pack.nbytes == lib.emph_linear_split_pack16_size(pieces) == \
            3 * 5 * pieces * 1024
        halves = pack.view(np.uint16).reshape(3, 5, pieces, 64, 8)
        values = (halves.astype(np.uint32) << 16).view(np.float32)
        rebuilt = np.zeros((80, 96), dtype=np.float64)
        seen = np.zeros((80, 96), dtype=np.int64)
        for j in range(3):
            for m in range(5):
                for lane in range(64):
                    channels = 32 * j + 16 * (e // 4) + 4 * (lane // 16) + e % 4
                    rebuilt[16 * m + lane % 16, channels] = \
                        values[j, m, :, lane].astype(np.float64).sum(0)
                    seen[16 * m + lane % 16, channels] += 1
        assert np.all(seen == 1)                    # every (row, channel) exactly once
        assert np.all(rebuilt[:, 80:] == 0.)
        error = np.abs(rebuilt[:, :80] - weight) / np.maximum(np.abs(weight), 1e-30)
        if pieces == 3:
            assert float(error.max()) == 0.         # 8 + 8 + 8 bits: all of fp32's
        else:
            assert float(error.max()) < 2.0 ** -16
    with pytest.raises(AssertionError):
        runtime.linear_split_pack(weight, 2, 64)


def test_split_kv_scratch_sizes_and_piece_codes():
    from emphases_amd import runtime
    lib = runtime.library()
    key, value = 6 * 64 * 16, 8 * 42 * 16      # a piece of a 64-key stage: K, V
    ld, segments = 64 * 100, 7
    slots = ld // 64 + segments + 1
    for code, (pk, pv) in ((2, (2, 2)), (3, (3, 3)), (32, (3, 2))):
        assert lib.emph_split_kv_bytes(ld, segments, 80, 2, code) == \
            slots * 2 * (pk * key + pv * value)
    assert lib.emph_split_kv_bytes(ld, segments, 80, 2, 4) == -1      # no such split
    assert lib.emph_split_kv_bytes(ld, segments, 64, 2, 2) == -1      # head dimension 32


def test_bench_line_stays_under_the_driver_limit(tmp_path, monkeypatch):
    """The contract's ONE line: whatever the side measurements hold (every
    record present, each far larger than life), `bench.compact_line` + `emit`
    stay under 10 KB - round 5's 25 KB line was not parsed by the driver - and
    the full record goes to the side-records file the line names (sha256)."""
    import io
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    fat = {f'field_{i}': 'x' * 200 for i in range(40)}
    roof = {key: 0.123456789012345 for key in bench.ROOFLINE_KEYS}
    roof.update(bound='mfma', kernel='conv1d_stack_frames_80x80_k3',
                unit='TFLOP/s', traffic=50049626, executed=fat,
                avg_launch_us_source='kernel-exact events, this run',
                rocprof_file='profiles/r6_bench_kernel_stats_1stream.csv',
                traffic_source='profiles/r6_pmc_summary.json')
    entry = dict(fat, ms_per_step=1.234567890123, utterances_per_s=1e5,
                 max_abs_dscore_vs_f32=1e-6, roofline=dict(roof))
    result = {
        'metric': 'utterances/s (10 s @16 kHz) whole-node',
        'value': 436123.4567890123, 'unit': 'utterances/s', 'n_gpus': 1,
        'steps': 200, 'warmup': 20, 'ms_per_step': 0.14712345678901,
        'ms_per_step_min': 0.1461234567890, 'ms_per_step_max': 0.1491234567,
        'timed_region_s': 0.0293456789012, 'regions': 9, 'timing': 'x' * 300,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'w' * 200, 'utterances_per_gpu': 64,
                   'parallelism': 'utterance-sharded x1', 'exchange': 'e' * 120},
        'frames_per_s_per_gpu': 4.36e8, 'checksum': 123.456,
        'roofline': roof, 'from_profiles': fat, 'kernels_us_rocprof': fat,
        'end_to_end': {'mfma_frac': 0.5, 'hbm_frac_compulsory': 0.03},
        'cpu_baseline': dict(fat, value=2098.5, unit='utterances/s', cores=16,
                             kind='port', sample='s' * 300, cpu='c' * 40,
                             cgroup_cpu_quota=16.0, physical_cores=128,
                             threads={'1': 185.5}),
        'end_to_end_api': {'float32': dict(fat, utterances_per_s_pipelined=7e4),
                           'pcm16': dict(fat, utterances_per_s_pipelined=1e5)},
        'single_utterance_api': {'default': {'ms_p50': 0.4, 'ms_p99': 0.6},
                                 'conv_tile_auto': {'ms_p50': 0.3},
                                 'oracle_1_core_ms': 5.4},
        'files_api': dict(fat, files_per_s=8e4),
        'configs_1_conv_bf16x3': entry, 'configs_2_transformer': entry,
        'configs_2_transformer_bf16x3': entry,
        'configs_2_transformer_bf16x3_fast': entry,
        'configs_2_transformer_bf16x6': {'error': 'RuntimeError("boom")'},
        'configs_3_corpus': {
            'rank0_of_8_device_only': dict(fat, utterances_per_s=2e5,
                                           frames_per_s=4e8),
            'whole_corpus_device_only': dict(fat, utterances_per_s=2e5)},
        'configs_4_longform': {'device_only': dict(fat, frames_per_s=4e8),
                               'api_pcm16': dict(fat, frames_per_s=1e8)},
        'configs_3_corpus_sharded': dict(fat, utterances_per_s=1e6,
                                         ms_per_step=9.),
        'configs_4_longform_sharded': dict(fat, utterances_per_s=1e4,
                                           ms_per_step=6.),
        'job': dict(fat, utterances=10000, frames=16000000, scores=480000,
                    checksum=1.5, frames_per_rank=[2000000] * 8,
                    lpt_imbalance=1.0001,
                    compute_only_ms_per_rank=[4.777123456789] * 8)}
    assert len(json.dumps(result)) > 50000
    line = bench.compact_line(result)
    assert len(json.dumps(line)) < 7000
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup',
                'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['value'] == result['value']             # (not rounded)
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in line['roofline'], key
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in line['cpu_baseline'], key
    side = line['side']
    assert side['bf16x3_ms_per_step'] == 1.23457
    assert side['single_utterance_ms_p50'] == 0.4
    assert side['configs_3_sharded_utterances_per_s'] == 1e6
    assert side['failed'] == ['configs_2_transformer_bf16x6']
    assert len(line['job']['frames_per_rank']) == 8
    # ... and through emit(): one line, the side file it names
    out = io.StringIO()
    monkeypatch.setattr(bench, 'LINE_OUT', out)
    monkeypatch.setattr(sys, 'argv', [
        'bench.py', '--side-records', str(tmp_path / 'side.json')])
    args = bench.parse_args()
    bench.emit(result, args)
    text = out.getvalue()
    assert text.count('\n') == 1 and len(text) < bench.LINE_LIMIT
    printed = json.loads(text)
    with open(tmp_path / 'side.json', 'rb') as file:
        data = file.read()
    assert printed['side_records']['sha256'] == hashlib.sha256(data).hexdigest()
    assert printed['side_records']['bytes'] == len(data)
    full = json.loads(data)
    assert full['files_api']['field_3'] == 'x' * 200
    assert 'frac' in full['notes']['roofline']
    # the default name of the file follows the command
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--config', 'transformer',
                                      '--precision', 'bf16x3'])
    assert os.path.basename(bench.side_path(bench.parse_args())) == \
        'bench_side_transformer_bf16x3.json'
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    assert bench.side_path(bench.parse_args()) == \
        os.path.join(ROOT, 'bench_side.json')
