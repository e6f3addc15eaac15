"""Word alignment container implementing the subset of `pypar.Alignment`
that the reference's hot path touches (`emphases/core.py:366-400`):
`len()`, `[i]`, `[a:b]`, word `.start()/.end()/.duration()`,
`.word_bounds(sample_rate, hopsize, silences=True)`, plus Praat TextGrid and
JSON I/O for the file API (`core.py:49,111`).

`pypar` is a third-party dependency that is not vendored by the reference, so
its behaviour is parity-unpinned; the one semantic the hot path relies on is
stated explicitly here: a *sliced* alignment reports word bounds in frames
relative to the first word of the slice.
"""
import json
import re

SILENCE = '<silent>'


class Word:
    """One aligned word (times in seconds)."""

    def __init__(self, word, start, end, phonemes=None):
        self.word = str(word)
        self._start = float(start)
        self._end = float(end)
        self.phonemes = phonemes

    def __str__(self):
        return self.word

    def __repr__(self):
        return f'Word({self.word!r}, {self._start}, {self._end})'

    def start(self):
        return self._start

    def end(self):
        return self._end

    def duration(self):
        return self._end - self._start


def frame_bounds(words, sample_rate, hopsize, origin=0):
    """`(int(start * sr / hop), int(end * sr / hop))` per word, minus origin."""
    return [
        (int(word.start() * sample_rate / hopsize) - origin,
         int(word.end() * sample_rate / hopsize) - origin)
        for word in words]


class Alignment:
    """Gap-free sequence of words."""

    def __init__(self, alignment, relative=False):
        if isinstance(alignment, Alignment):
            words = list(alignment._words)
        elif isinstance(alignment, (list, tuple)):
            words = list(alignment)
        elif isinstance(alignment, dict):
            words = _words_from_json(alignment)
        else:
            path = str(alignment)
            if path.endswith('.json'):
                with open(path, encoding='utf-8') as file:
                    words = _words_from_json(json.load(file))
            elif path.endswith('.TextGrid'):
                with open(path, encoding='utf-8') as file:
                    words = _words_from_textgrid(file.read())
            else:
                raise ValueError(
                    f'Alignment file format of {path} is not supported')
        self._words = _fill_gaps(words)
        self._relative = relative
        self._times = None
        self.times()          # the batch planner reads the words as one array

    @classmethod
    def from_frames(cls, bounds, names=None, frames_per_second=100.0):
        """Build from integer frame bounds [2, W] (seconds = frames / 100)."""
        starts, ends = bounds
        names = names or [f'w{j}' for j in range(len(starts))]
        return cls([
            Word(name, int(s) / frames_per_second, int(e) / frames_per_second)
            for name, s, e in zip(names, starts, ends)])

    def __len__(self):
        return len(self._words)

    def __iter__(self):
        return iter(self._words)

    def __getitem__(self, index):
        if isinstance(index, slice):
            return Alignment(self._words[index], relative=True)
        return self._words[index]

    def __str__(self):
        return ' '.join(str(word) for word in self._words)

    def start(self):
        return self._words[0].start()

    def end(self):
        return self._words[-1].end()

    def duration(self):
        return self.end() - self.start()

    def words(self):
        return list(self._words)

    def times(self):
        """float64 [W, 2] (start, end) seconds of every word (cached: the
        batch planner reads all words of all utterances as arrays)."""
        cached = self._times
        if cached is None or cached.shape[0] != len(self._words):
            import numpy as np
            cached = np.array(
                [(word._start, word._end) if isinstance(word, Word)
                 else (word.start(), word.end()) for word in self._words],
                dtype=np.float64).reshape(len(self._words), 2)
            self._times = cached
        return cached

    def word_bounds(self, sample_rate, hopsize=1, silences=False):
        words = [
            word for word in self._words
            if silences or str(word) != SILENCE]
        origin = 0
        if self._relative and self._words:
            origin = int(self._words[0].start() * sample_rate / hopsize)
        return frame_bounds(words, sample_rate, hopsize, origin)

    ###########################################################################
    # File I/O
    ###########################################################################

    def json(self):
        return {'words': [
            {'alignedWord': str(word), 'start': word.start(),
             'end': word.end()} for word in self._words]}

    def save(self, file):
        file = str(file)
        if file.endswith('.json'):
            with open(file, 'w', encoding='utf-8') as out:
                json.dump(self.json(), out, indent=4)
        elif file.endswith('.TextGrid'):
            with open(file, 'w', encoding='utf-8') as out:
                out.write(_textgrid(self._words))
        else:
            raise ValueError(
                f'Alignment file format of {file} is not supported')


def _fill_gaps(words):
    """Insert silences so that consecutive words touch."""
    result = []
    for word in words:
        if result and word.start() > result[-1].end():
            result.append(Word(SILENCE, result[-1].end(), word.start()))
        result.append(word)
    return result


def _words_from_json(content):
    return [
        Word(item.get('alignedWord', item.get('word')),
             item['start'], item['end'])
        for item in content['words']]


_INTERVAL = re.compile(
    r'xmin\s*=\s*([-0-9.eE+]+)\s*xmax\s*=\s*([-0-9.eE+]+)\s*'
    r'text\s*=\s*"((?:[^"]|"")*)"')


def _words_from_textgrid(text):
    """Read the word tier of a long-format Praat TextGrid.

    The tier named "words"/"word" is used if present, otherwise the tier with
    the fewest intervals (phoneme tiers are finer than word tiers)."""
    tiers = re.split(r'item\s*\[\d+\]\s*:', text)[1:]
    parsed = []
    for tier in tiers:
        name = re.search(r'name\s*=\s*"([^"]*)"', tier)
        intervals = [
            (float(a), float(b), t.replace('""', '"'))
            for a, b, t in _INTERVAL.findall(tier)]
        parsed.append((name.group(1).lower() if name else '', intervals))
    if not parsed:
        raise ValueError('TextGrid holds no interval tiers')
    named = [t for t in parsed if t[0] in ('words', 'word')]
    _, intervals = named[0] if named else min(
        parsed, key=lambda tier: len(tier[1]))
    return [
        Word(text if text.strip() and text != 'sp' else SILENCE, a, b)
        for a, b, text in intervals]


def _textgrid(words):
    xmax = words[-1].end() if words else 0.
    lines = [
        'File type = "ooTextFile"', 'Object class = "TextGrid"', '',
        'xmin = 0', f'xmax = {xmax!r}', 'tiers? <exists>', 'size = 1',
        'item []:', '    item [1]:', '        class = "IntervalTier"',
        '        name = "words"', '        xmin = 0',
        f'        xmax = {xmax!r}',
        f'        intervals: size = {len(words)}']
    for i, word in enumerate(words):
        text = '' if str(word) == SILENCE else str(word).replace('"', '""')
        lines += [
            f'        intervals [{i + 1}]:',
            f'            xmin = {word.start()!r}',
            f'            xmax = {word.end()!r}',
            f'            text = "{text}"']
    return '\n'.join(lines) + '\n'
