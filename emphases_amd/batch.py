"""Ragged batch planning: utterances -> word-boundary chunks -> packed layout.

`chunk_utterance` is the host arithmetic of the reference's `preprocess`
generator (`emphases/core.py:345-418`): greedy accumulation of words up to
`batch_size` frames in float64 with floor division (`convert.py:19-31`), chunk
audio cut at frame-quantised word times out of the 432-zero-padded signal, and
chunk-relative word bounds.  Chunks that the reference silently drops (its
feature extraction raises inside `except RuntimeError: pass`, `core.py:403-415`,
when the chunk is no longer than the 432-sample reflect pad) are dropped here by
an explicit length test.

`Plan` lays a list of chunks ("segments") out for the HIP kernels: every
segment gets its own column range on a packed frame axis and a packed word
axis (offsets aligned to 16 columns so MFMA tiles and 16-byte loads line up),
and all integer metadata is packed into ONE buffer so that a batch costs a
single small host-to-device copy.
"""
import dataclasses

import numpy as np

from . import alignment as alignment_module
from . import config as cfg
from . import convert
from . import runtime

ALIGN = 16     # column alignment of every segment on the packed axes
LEAD = 16      # unused columns before the first segment (halo reads stay inside)
TAIL = 128     # unused columns after the last segment (a 64-wide tile of the
               # last segment may read its halo past the end)


@dataclasses.dataclass
class Segment:
    """One independently processed chunk of one utterance."""
    utterance: int          # index of the utterance in the batch
    start_word: int         # first word (index into the utterance's alignment)
    end_word: int           # one past the last word
    start_sample: int       # chunk start in the 432-zero-padded signal
    length: int             # chunk length in samples
    frames: int
    bounds: np.ndarray      # int64 [2, words], chunk-relative frames


def _word_times(alignment):
    """float64 [W, 2] (start, end) seconds of the words of an alignment that
    follows the pypar protocol (`len()`, `[i]`, `.start()`, `.end()`)."""
    if hasattr(alignment, 'times'):
        return alignment.times()
    return np.array(
        [(word.start(), word.end()) for word in
         (alignment[i] for i in range(len(alignment)))],
        dtype=np.float64).reshape(len(alignment), 2)


def _foreign(alignment):
    """An alignment object that is not ours but carries its own
    `word_bounds` (a real `pypar.Alignment`): the reference takes the chunk's
    bounds from `alignment[start:end].word_bounds(...)` (`core.py:384-392`),
    so whatever that method returns is what the model must see."""
    return type(alignment) is not alignment_module.Alignment and \
        not isinstance(alignment, (list, np.ndarray,
                                   alignment_module.Alignment)) and \
        hasattr(alignment, 'word_bounds')


def chunk_utterance(alignment, num_samples, batch_size=None, utterance=0):
    """Chunk plan of one utterance; `alignment` follows the pypar protocol.

    The reference's per-word Python loops (`core.py:361-400`) as float64 array
    operations with the same IEEE steps: floor division, a sequential running
    sum (`np.cumsum`), truncation."""
    times = alignment if isinstance(alignment, (list, np.ndarray)) else \
        _word_times(alignment)
    foreign = _foreign(alignment)
    padded = num_samples + 2 * cfg.PADDING
    total_frames = int(padded / cfg.HOPSIZE)                     # core.py:359
    limit = total_frames if batch_size is None else batch_size
    segments = []
    count = len(times)
    if not count:
        return segments
    table = np.asarray(times, dtype=np.float64).reshape(count, 2)
    starts, ends = table[:, 0], table[:, 1]
    word_frames = convert.seconds_to_frames(ends - starts)      # core.py:373
    start_frames = (starts * cfg.SAMPLE_RATE / cfg.HOPSIZE).astype(np.int64)
    end_frames = (ends * cfg.SAMPLE_RATE / cfg.HOPSIZE).astype(np.int64)
    start = 0
    while start < count:
        # the chunk grows word by word until the running frame count of the
        # words it already holds exceeds the limit (the alignment's last word
        # is never counted)
        running = np.cumsum(word_frames[start:count - 1])
        over = np.nonzero(running.astype(np.int64) > limit)[0]
        end = start + 1 + int(over[0]) if over.size else count
        origin = int(start_frames[start])
        if foreign:
            # core.py:384-392, verbatim: the caller's own slicing and bounds
            bounds = np.asarray(alignment[start:end].word_bounds(
                cfg.SAMPLE_RATE, cfg.HOPSIZE, silences=True),
                dtype=np.int64).reshape(-1, 2).T
        else:
            bounds = np.stack(
                [start_frames[start:end], end_frames[start:end]]) - origin
        start_sample = int(convert.frames_to_samples(
            int(convert.seconds_to_frames(float(starts[start])))))  # core.py:395
        end_sample = int(convert.frames_to_samples(
            int(convert.seconds_to_frames(float(ends[end - 1])))))  # core.py:398
        start_sample = max(0, min(start_sample, padded))
        end_sample = max(0, min(end_sample, padded))   # slice clamps
        length = max(0, end_sample - start_sample)
        # reflect padding needs more than PADDING samples (mels.py:31-36)
        if length > cfg.PADDING:
            segments.append(Segment(
                utterance, start, end, start_sample, length,
                1 + (length + 2 * cfg.PADDING - cfg.NUM_FFT) // cfg.HOPSIZE,
                bounds))
        start = end
    return segments


def plan_batch(alignments, lengths, batch_size=None, tables=None):
    """`Plan` of a whole batch: utterance u has `lengths[u]` samples at offset
    sum(lengths[:u]) of the packed audio.  One pass of the library over all
    words of all utterances (`emph_plan_batch`) when no utterance needs more
    than one chunk (the default `batch_size=None`); the per-utterance planner
    otherwise.  `tables`: (times float64 [W, 2], counts int64 [U]) of the
    alignments when the caller holds them as one table already
    (`files.FileBatch`) - `alignments` is then only consulted on the slow path
    (and may be a callable that builds the list)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    offsets = np.cumsum(lengths) - lengths
    slow = batch_size is not None
    if not slow and tables is None:
        slow = any(_foreign(a) for a in alignments)
        if not slow and len(alignments):
            rows = [a.times() if type(a) is alignment_module.Alignment else
                    np.asarray(a, dtype=np.float64).reshape(-1, 2)
                    if isinstance(a, (list, np.ndarray)) else _word_times(a)
                    for a in alignments]
            counts = np.array([len(t) for t in rows], dtype=np.int64)
            tables = (np.concatenate(rows) if counts.sum() else
                      np.zeros((0, 2), dtype=np.float64), counts)
    if not slow and tables is not None and len(lengths):
        columns = _plan_columns(tables[0], tables[1], lengths)
        if columns is not None:
            utterance, start_sample, length, frames, words, bounds = columns
            return Plan.from_columns(
                utterance, np.zeros(len(utterance), dtype=np.int64),
                start_sample, length, frames, words, bounds, offsets, lengths)
    if callable(alignments):
        alignments = alignments()
    segments = []
    for index, (alignment, length) in enumerate(zip(alignments, lengths)):
        segments.extend(
            chunk_utterance(alignment, int(length), batch_size, index))
    return Plan(segments, offsets, lengths)


def _plan_columns(times, counts, lengths):
    """`emph_plan_batch`: per-chunk columns (utterance, start_sample, length,
    frames, words, bounds [2, words]) of a batch in which every utterance is
    one chunk, or None when it has to be planned one utterance at a time
    (tests/plan_reference.py holds the numpy restatement)."""
    import ctypes
    lib = runtime.library()
    times = np.ascontiguousarray(times, dtype=np.float64)
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    count, total = len(counts), int(times.shape[0])
    if int(counts.sum()) != total or len(lengths) != count:
        raise ValueError('word tables and utterance counts disagree')
    columns = np.empty((5, max(count, 1)), dtype=np.int64)
    bounds = np.empty((2, max(total, 1)), dtype=np.int64)
    segments, words = ctypes.c_int64(0), ctypes.c_int64(0)
    status = lib.emph_plan_batch(
        times.ctypes.data, counts.ctypes.data, lengths.ctypes.data, count,
        cfg.SAMPLE_RATE, cfg.HOPSIZE, cfg.PADDING, cfg.NUM_FFT,
        *[columns[row].ctypes.data for row in range(5)], bounds.ctypes.data,
        bounds.shape[1], ctypes.byref(segments), ctypes.byref(words))
    if status == 1:
        return None
    runtime.check(status, 'emph_plan_batch')
    kept = columns[:, :segments.value]
    return tuple(kept) + (bounds[:, :words.value],)


def score_counts(alignments, lengths, batch_size=None):
    """Scores each utterance will have: its words minus those of chunks the
    reference drops (`core.py:414-415`) - known from the plan alone, before
    anything is computed (what a sharded run sizes its one score exchange
    from, `dist.exchange_counts`)."""
    plan = plan_batch(alignments, lengths, batch_size)
    if not len(plan):
        return np.zeros(len(lengths), dtype=np.int64)
    return np.bincount(
        plan.utterance, weights=plan.words,
        minlength=len(lengths)).astype(np.int64)


def _round_up(value, multiple):
    return (value + multiple - 1) // multiple * multiple


def _tiles(counts, offsets, block, least=0, most=1 << 62):
    """Tile table int32 [n, 4]: (segment, first position, segment's first
    column, segment's positions) of every `block`-wide tile of the segments
    with `least <= count <= most` (`emph_plan_tiles`, host arithmetic in the
    library; tests/plan_reference.py holds the numpy restatement)."""
    lib = runtime.library()
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    rows = lib.emph_plan_tiles(
        counts.ctypes.data, offsets.ctypes.data, len(counts), block, least,
        most, None)
    table = np.empty((rows, 4), dtype=np.int32)
    lib.emph_plan_tiles(
        counts.ctypes.data, offsets.ctypes.data, len(counts), block, least,
        most, table.ctypes.data)
    return table


class Plan:
    """Packed layout of a batch of segments."""

    def __init__(self, segments, audio_offsets, audio_lengths):
        """`audio_offsets[u]`, `audio_lengths[u]`: where utterance `u` sits in
        the packed audio buffer."""
        count = len(segments)
        column = lambda name: np.array(  # noqa: E731
            [getattr(s, name) for s in segments], dtype=np.int64)
        words = np.array([s.bounds.shape[1] for s in segments], dtype=np.int64)
        bounds = np.concatenate([s.bounds for s in segments], axis=1) \
            if count else np.zeros((2, 0), dtype=np.int64)
        self._build(
            column('utterance'), column('start_word'), column('start_sample'),
            column('length'), column('frames'), words, bounds,
            np.asarray(audio_offsets, dtype=np.int64),
            np.asarray(audio_lengths, dtype=np.int64))
        self._segments = list(segments)

    @classmethod
    def from_columns(cls, utterance, start_word, start_sample, length, frames,
                     words, bounds, audio_offsets, audio_lengths):
        """Plan from per-segment arrays (`bounds`: int64 [2, sum(words)], the
        segments' chunk-relative word bounds back to back)."""
        plan = cls.__new__(cls)
        plan._build(utterance, start_word, start_sample, length, frames, words,
                    bounds, np.asarray(audio_offsets, dtype=np.int64),
                    np.asarray(audio_lengths, dtype=np.int64))
        plan._segments = None
        return plan

    def _build(self, utterance, start_word, start_sample, length, frames,
               words, bounds, audio_offsets, audio_lengths):
        count = len(frames)
        frame_off = LEAD + np.concatenate(
            [[0], np.cumsum(_round_up(frames, ALIGN))[:-1]]) \
            if count else np.zeros(0, dtype=np.int64)
        word_off = LEAD + np.concatenate(
            [[0], np.cumsum(_round_up(words, ALIGN))[:-1]]) \
            if count else np.zeros(0, dtype=np.int64)
        self.utterance = utterance
        self.start_word = start_word
        self.frames = frames
        self.words = words
        self.frame_off = frame_off.astype(np.int64)
        self.word_off = word_off.astype(np.int64)
        self.ld_frames = int(
            LEAD + _round_up(frames, ALIGN).sum() + TAIL) if count else TAIL
        self.ld_words = int(
            LEAD + _round_up(words, ALIGN).sum() + TAIL) if count else TAIL
        self.total_frames = int(frames.sum())
        self.total_words = int(words.sum())
        self.segment_bounds = bounds          # [2, total_words], packed

        table = np.zeros((count, runtime.SEG_FIELDS), dtype=np.int64)
        if count:
            table[:, runtime.SEG_AUDIO_OFF] = audio_offsets[utterance]
            table[:, runtime.SEG_AUDIO_LEN] = audio_lengths[utterance]
            table[:, runtime.SEG_START] = start_sample
            table[:, runtime.SEG_LENGTH] = length
        table[:, runtime.SEG_FRAME_OFF] = self.frame_off
        table[:, runtime.SEG_FRAMES] = frames
        table[:, runtime.SEG_WORD_OFF] = self.word_off
        table[:, runtime.SEG_WORDS] = words
        self.table = table

        # packed word-axis column of every word, in segment order
        first = np.cumsum(words) - words
        self._columns = (
            np.repeat(self.word_off - first, words) +
            np.arange(self.total_words, dtype=np.int64)) if count else \
            np.zeros(0, dtype=np.int64)
        self.bounds = np.zeros((2, self.ld_words), dtype=np.int32)
        self.word_segment = np.full(self.ld_words, -1, dtype=np.int32)
        if self.total_words:
            self.bounds[:, self._columns] = bounds
            self.word_segment[self._columns] = np.repeat(
                np.arange(count, dtype=np.int32), words)
        self._tiles = {}
        self._pieces = {}

    @property
    def segments(self):
        """The segments as `Segment` objects (built on demand: the batch
        planner works on arrays)."""
        if self._segments is None:
            first = np.cumsum(self.words) - self.words
            self._segments = [
                Segment(int(self.utterance[i]), int(self.start_word[i]),
                        int(self.start_word[i] + self.words[i]),
                        int(self.table[i, runtime.SEG_START]),
                        int(self.table[i, runtime.SEG_LENGTH]),
                        int(self.frames[i]),
                        self.segment_bounds[
                            :, first[i]:first[i] + self.words[i]])
                for i in range(len(self.frames))]
        return self._segments

    def __len__(self):
        return len(self.frames)

    def tiles(self, axis, block, least=None, most=None):
        """Tile table of an axis; `least` / `most`: only the tiles of segments
        with that many positions (which kernel takes a segment then depends
        on the segment alone, never on what else is in the batch)."""
        if least is not None:
            key = (axis, block, least, most)
            if key not in self._tiles:
                if axis == runtime.AXIS_FRAMES:
                    self._tiles[key] = _tiles(
                        self.frames, self.frame_off, block, least, most)
                else:
                    self._tiles[key] = _tiles(
                        self.words, self.word_off, block, least, most)
            return self._tiles[key]
        key = (axis, block)
        if key not in self._tiles:
            if axis == runtime.AXIS_FRAMES:
                self._tiles[key] = _tiles(self.frames, self.frame_off, block)
            elif axis == runtime.AXIS_DECODER:
                # `block` = (layers, kernel_size, out_kernel_size) of the decoder
                self._tiles[key] = runtime.word_decoder_tiles(
                    self.words, self.word_off, *block)
            else:
                self._tiles[key] = _tiles(self.words, self.word_off, block)
        return self._tiles[key]

    def conv_spans(self):
        """int32 [n, 8] table of `emph_conv1d_stack` (`emph_conv_stack_spans`):
        every segment cut into the fewest spans one workgroup can own through
        several fused conv layers."""
        cached = self._tiles.get('conv_spans')
        if cached is None:
            lib = runtime.library()
            frames = np.ascontiguousarray(self.frames, dtype=np.int64)
            offsets = np.ascontiguousarray(self.frame_off, dtype=np.int64)
            count = lib.emph_conv_stack_spans(
                frames.ctypes.data, offsets.ctypes.data, len(frames), None)
            cached = np.zeros((count, 8), dtype=np.int32)
            lib.emph_conv_stack_spans(
                frames.ctypes.data, offsets.ctypes.data, len(frames),
                cached.ctypes.data)
            self._tiles['conv_spans'] = cached
        return cached

    def sum_restarts(self, spans=None, tile=64, step=64):
        """Packed frame columns at which the running sum of the last
        frame-rate layer restarts: every `tile` frames from a segment's start
        (`emph_conv1d_winograd4_word_sums`), or - `spans` - at a span's first
        own position and every `step` COMPUTED positions inside it (64:
        `emph_conv1d_stack`; 32: `emph_conv1d_split`, whose waves own 32
        positions each)."""
        if spans is None:
            tiles = -(-self.frames // tile)
            segment = np.repeat(np.arange(len(self.frames)), tiles)
            local = np.arange(int(tiles.sum())) - np.repeat(
                np.cumsum(tiles) - tiles, tiles)
            return self.frame_off[segment] + local * tile
        first, column = spans[:, 1].astype(np.int64), spans[:, 2].astype(np.int64)
        owned, computed = spans[:, 4].astype(np.int64), spans[:, 5].astype(np.int64)
        starts = np.maximum(
            first[:, None],
            computed[:, None] + step * np.arange(256 // step)[None])
        keep = starts < (first + owned)[:, None]
        keep[:, 1:] &= starts[:, 1:] > starts[:, :-1]
        return np.unique((column[:, None] + starts)[keep])

    def word_sum_tables(self, restarts=None):
        """Tables of the fused per-word sum (`emph_conv1d_winograd4_word_sums`
        or `emph_conv1d_stack`, + `emph_word_sums`; `emphases/core.py:438-454`):
        the last frame-rate layer keeps a running sum of its frames that
        restarts at the columns `restarts` (`sum_restarts()`; default: every
        64 frames of a segment) and stores it only at the frames a word needs -

            word [s, e) (clamped to its chunk like a Python slice) is cut at
            the restarts inside it into parts [a, b):
                + running sum at frame b - 1
                - running sum at frame a - 1     (if a is not itself a restart)

        Returns a dict of int32 arrays: `slot_map` [ld_frames] (packed frame
        column -> row of the sums buffer, -1), `terms` (signed rows: r adds,
        ~r subtracts; a word's terms part by part, plus before minus),
        `first` [ld_words + 1] (CSR over packed word columns), `lengths`
        [ld_words] (e - s; -1 on alignment padding columns), and `n_slots`.
        Built by the library (`emph_plan_word_sums`: two passes over the
        words; tests/plan_reference.py holds the numpy restatement)."""
        key = ('word_sums', None if restarts is None else restarts.tobytes())
        cached = self._tiles.get(key)
        if cached is not None:
            return cached
        if restarts is None:
            restarts = self.sum_restarts()
        restarts = np.ascontiguousarray(restarts, dtype=np.int64)
        views = {
            'slot_map': np.empty(self.ld_frames, dtype=np.int32),
            'first': np.empty(self.ld_words + 1, dtype=np.int32),
            'lengths': np.empty(self.ld_words, dtype=np.int32),
            # (two terms per part; a word has 1 + restarts-inside-it parts)
            'terms': np.empty(4 * self.total_words + 2 * len(restarts) + 8,
                              dtype=np.int32)}
        count, slots = self._word_sums_into(restarts, views)
        if count < 0:
            views['terms'] = np.empty(-count, dtype=np.int32)
            count, slots = self._word_sums_into(restarts, views)
        tables = {
            'slot_map': views['slot_map'], 'terms': views['terms'][:count].copy(),
            'first': views['first'], 'lengths': views['lengths'],
            'n_slots': slots}
        self._tiles[key] = tables
        return tables

    def _word_sums_into(self, restarts, views):
        """`emph_plan_word_sums` into the int32 arrays `views` (slot_map
        [ld_frames], first [ld_words + 1], lengths [ld_words], terms);
        returns (terms, slots), terms = -(needed) when `terms` is too small."""
        import ctypes
        lib = runtime.library()
        frames = np.ascontiguousarray(self.frames, dtype=np.int64)
        frame_off = np.ascontiguousarray(self.frame_off, dtype=np.int64)
        words = np.ascontiguousarray(self.words, dtype=np.int64)
        columns = np.ascontiguousarray(self._columns, dtype=np.int64)
        bounds = np.ascontiguousarray(self.segment_bounds, dtype=np.int64)
        slots = ctypes.c_int32(0)
        terms = views['terms']
        count = lib.emph_plan_word_sums(
            frames.ctypes.data, frame_off.ctypes.data, words.ctypes.data,
            len(frames), columns.ctypes.data, bounds.ctypes.data,
            self.total_words, restarts.ctypes.data, len(restarts),
            self.ld_frames, self.ld_words, views['slot_map'].ctypes.data,
            views['first'].ctypes.data, views['lengths'].ctypes.data,
            terms.ctypes.data, terms.size, ctypes.byref(slots))
        return int(count), int(slots.value)

    def pieces(self, method):
        """Layout for DOWNSAMPLE_LOCATION = 'input' (`model/core.py:41-87`,
        `core.py:552-586`): every word of a chunk becomes its own sequence,
        zero-padded to the longest word of that chunk, encoded on its own and
        pooled over the PADDED axis.  Returns a `Pieces`."""
        if method not in self._pieces:
            self._pieces[method] = Pieces(self, method)
        return self._pieces[method]

    def word_columns(self):
        """Packed word-axis column of every word, in segment order."""
        return self._columns

    def stack_restarts(self, step=64):
        """`sum_restarts` of the plan's own span table (kept: the packer and the
        engine both ask)."""
        key = ('stack_restarts', step)
        if key not in self._tiles:
            self._tiles[key] = np.ascontiguousarray(
                self.sum_restarts(self.conv_spans(), step=step), dtype=np.int64)
        return self._tiles[key]

    def pack_metadata(self, tile_requests, word_sums=False, spans=False,
                      sum_step=64, _capacity=None):
        """All integer metadata as one int32 array plus the element offset of
        every piece (the int64 table first, so it stays 8-byte aligned;
        every piece starts on a 16-byte boundary).  `word_sums`: with the
        tables of `word_sum_tables()`; `spans`: with the span table of
        `emph_conv1d_stack`, whose restarts the word sums then follow.
        The word-sum tables - a row per packed frame column among them, the
        largest piece by far - are built by the library straight into the
        array (the terms, whose number is known afterwards, at its end)."""
        pieces = [('table', self.table.view(np.int32).ravel()),
                  ('bounds', self.bounds.ravel()),
                  ('word_segment', self.word_segment)]
        if spans:
            pieces.append(('conv_spans', self.conv_spans().ravel()))
        for request in tile_requests:
            pieces.append(
                (('tiles',) + tuple(request), self.tiles(*request).ravel()))
        restarts = None
        if word_sums:
            restarts = self.stack_restarts(sum_step) if spans else \
                np.ascontiguousarray(self.sum_restarts(), dtype=np.int64)
            # (two terms per part; a word has 1 + restarts-inside-it parts)
            capacity = _capacity or \
                4 * self.total_words + 2 * len(restarts) + 8
            pieces += [(('word_sums', 'slot_map'), self.ld_frames),
                       (('word_sums', 'first'), self.ld_words + 1),
                       (('word_sums', 'lengths'), self.ld_words),
                       (('word_sums', 'terms'), capacity)]
        # one output array, every piece copied (or built) once, 16-byte
        # aligned starts
        offsets = {}
        cursor = 0
        for name, array in pieces:
            size = array if isinstance(array, int) else array.size
            offsets[name] = (cursor, size)
            cursor += _round_up(max(size, 1), 4)
        packed = np.empty(cursor, dtype=np.int32)
        for name, array in pieces:
            start, size = offsets[name]
            if not isinstance(array, int):
                packed[start:start + size] = array
            packed[start + size:start + _round_up(max(size, 1), 4)] = 0
        if word_sums:
            views = {name: packed[offsets[('word_sums', name)][0]:][
                :offsets[('word_sums', name)][1]]
                for name in ('slot_map', 'first', 'lengths', 'terms')}
            count, slots = self._word_sums_into(restarts, views)
            if count < 0:       # (words that overlap: more parts than words)
                return self.pack_metadata(
                    tile_requests, word_sums, spans, sum_step, -count)
            start, _ = offsets[('word_sums', 'terms')]
            offsets[('word_sums', 'terms')] = (start, count)
            packed = packed[:start + _round_up(max(count, 1), 4)]
            packed[start + count:] = 0
            self._tiles[('word_sums', restarts.tobytes())] = {
                'slot_map': views['slot_map'], 'terms': packed[start:start + count],
                'first': views['first'], 'lengths': views['lengths'],
                'n_slots': slots}
        return packed, offsets


class Pieces:
    """Word pieces of a Plan: `plan` is the packed layout of the pieces (one
    segment per word, frames = the chunk's longest word), `gather` the int64
    [n, 4] table of emph_gather_columns, and `bounds` / `word_piece` the
    emph_segment_reduce tables that pool piece p into the ORIGINAL word column
    of its word."""

    def __init__(self, parent, method):
        segments, sources = [], []
        for index, segment in enumerate(parent.segments):
            starts, ends = segment.bounds[0], segment.bounds[1]
            if not starts.size:
                continue
            if (starts < 0).any() or (ends > segment.frames).any() or \
                    (ends < starts).any():
                # the reference's slice assignment raises here too
                raise ValueError(
                    'word bounds outside the chunk are not supported with '
                    "DOWNSAMPLE_LOCATION='input'")
            longest = int((ends - starts).max())
            if longest < 1:
                raise ValueError('no word of the chunk covers a frame')
            for j in range(starts.size):
                length = int(ends[j] - starts[j])
                # center: embedding[:, length // 2]; otherwise the whole
                # padded axis (mean divides by `longest`)
                end = length if method == 'center' else longest
                segments.append(Segment(
                    0, 0, 1, 0, 0, longest,
                    np.array([[0], [end]], dtype=np.int64)))
                sources.append((
                    int(parent.frame_off[index]) + int(starts[j]), length,
                    int(parent.word_off[index]) + j))
        self.plan = Plan(segments, [0], [0])
        count = len(segments)
        self.gather = np.zeros((count, 4), dtype=np.int64)
        self.bounds = np.zeros((2, parent.ld_words), dtype=np.int32)
        self.word_piece = np.full(parent.ld_words, -1, dtype=np.int32)
        for piece, (source, length, column) in enumerate(sources):
            self.gather[piece] = (
                source, length, self.plan.frame_off[piece],
                self.plan.frames[piece])
            self.bounds[:, column] = segments[piece].bounds[:, 0]
            self.word_piece[column] = piece


def chunk_audio(audio, segment):
    """The chunk's samples as the reference slices them out of the
    432-zero-padded signal (`core.py:357-358,395-401`): float32 tensor
    [1, segment.length].  Host-side; only the pitch tracker needs it (the HIP
    front-end does this indexing on the fly)."""
    import torch
    audio = audio.reshape(-1)
    if audio.dtype == torch.int16:                     # 16-bit PCM
        audio = audio.to(torch.float32) / 32768.
    first = segment.start_sample - cfg.PADDING
    last = first + segment.length
    lo, hi = max(first, 0), min(last, int(audio.shape[0]))
    piece = torch.zeros(segment.length, dtype=torch.float32)
    if hi > lo:
        piece[lo - first:hi - first] = audio[lo:hi].to(torch.float32).cpu()
    return piece[None]


def pack_tracks(plan, tracks):
    """Pitch-tracker outputs of every segment on the packed frame axis.

    tracks: per segment, `(pitch [1, Fc], periodicity [1, Fc])` as
    `penn.from_audio` returns them (`data/preprocess/core.py:84-92`).
    Returns float32 numpy [2, ld_frames]; padding columns hold 1 (log2 -> 0)."""
    packed = np.ones((2, plan.ld_frames), dtype=np.float32)
    for segment, off, count, pair in zip(
            plan.segments, plan.frame_off, plan.frames, tracks):
        for row, track in enumerate(pair):
            track = np.asarray(
                track.detach().cpu() if hasattr(track, 'detach') else track,
                dtype=np.float32).reshape(-1)
            if track.size != count:
                # the reference's torch.cat would raise and the chunk would be
                # dropped without a trace (core.py:123, core.py:414-415)
                raise ValueError(
                    f'the pitch tracker returned {track.size} frames for a '
                    f'chunk of {count}')
            packed[row, off:off + count] = track
    return packed
