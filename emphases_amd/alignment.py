"""Word alignment container implementing the subset of `pypar.Alignment`
that the reference's hot path touches (`emphases/core.py:366-400`):
`len()`, `[i]`, `[a:b]`, word `.start()/.end()/.duration()`,
`.word_bounds(sample_rate, hopsize, silences=True)`, plus Praat TextGrid and
JSON I/O for the file API (`core.py:49,111`): the reference loads the
alignment with `pypar.Alignment(text_file)` and saves THAT alignment next to
the scores (`core.py:105-112`), phoneme tier included, so reading and writing
here is a round trip - both interval tiers (words and phonemes, original tier
names and order) come back out; Praat's long ("xmin = ...") and short text
formats, UTF-8 (with or without BOM) and UTF-16 files are read.

`pypar` is a third-party dependency that is not vendored by the reference, so
its behaviour is parity-unpinned; the one semantic the hot path relies on is
stated explicitly here: a *sliced* alignment reports word bounds in frames
relative to the first word of the slice.
"""
import json
import re

SILENCE = '<silent>'


class Phoneme:
    """One aligned phoneme (times in seconds): `pypar.Phoneme`'s accessors."""

    def __init__(self, phoneme, start, end):
        self.phoneme = str(phoneme)
        self._start = float(start)
        self._end = float(end)

    def __str__(self):
        return self.phoneme

    def __repr__(self):
        return f'Phoneme({self.phoneme!r}, {self._start}, {self._end})'

    def __eq__(self, other):
        return isinstance(other, Phoneme) and \
            (self.phoneme, self._start, self._end) == \
            (other.phoneme, other._start, other._end)

    def start(self):
        return self._start

    def end(self):
        return self._end

    def duration(self):
        return self._end - self._start


class Word:
    """One aligned word (times in seconds).  `phonemes`: its `Phoneme`s when
    the alignment file had a phoneme tier, else None."""

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

    # (word tier name, phoneme tier name, phoneme tier first?) of the file the
    # alignment came from: what `save` writes back
    tiers = ('words', 'phones', False)

    def __init__(self, alignment, relative=False):
        if isinstance(alignment, Alignment):
            words = list(alignment._words)
            self.tiers = alignment.tiers
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
                with open(path, 'rb') as file:
                    words, self.tiers = _words_from_textgrid(
                        decode(file.read()))
            else:
                raise ValueError(
                    f'Alignment file format of {path} is not supported')
        self._word_list = _fill_gaps(words)
        self._loader = None
        self._relative = relative
        self._times = None
        self.times()          # the batch planner reads the words as one array

    @classmethod
    def lazy(cls, times, loader):
        """Alignment whose word TIMES are known (float64 [W, 2], gaps already
        filled) and whose `Word` objects are built by `loader() -> (words,
        tiers)` only when somebody asks for them: the batch planner reads
        `times()` alone (`files.FileBatch`)."""
        self = cls.__new__(cls)
        self._word_list = None
        self._loader = loader
        self._relative = False
        self._times = times
        return self

    @property
    def _words(self):
        if self._word_list is None:
            self._word_list, self.tiers = self._loader()
            self._loader = None
        return self._word_list

    @classmethod
    def from_frames(cls, bounds, names=None, frames_per_second=100.0):
        """Build from integer frame bounds [2, W] (seconds = frames / 100)."""
        starts, ends = bounds
        names = names or [f'w{j}' for j in range(len(starts))]
        return cls([
            Word(name, int(s) / frames_per_second, int(e) / frames_per_second)
            for name, s, e in zip(names, starts, ends)])

    def __len__(self):
        return len(self._times) if self._word_list is None \
            else len(self._word_list)

    def __iter__(self):
        return iter(self._words)

    def __getitem__(self, index):
        if isinstance(index, slice):
            sliced = Alignment(self._words[index], relative=True)
            sliced.tiers = self.tiers
            return sliced
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

    def phonemes(self):
        """All phonemes in order (`pypar.Alignment.phonemes`); empty when the
        alignment has no phoneme tier."""
        return [phoneme for word in self._words
                for phoneme in (getattr(word, 'phonemes', None) or [])]

    def times(self):
        """float64 [W, 2] (start, end) seconds of every word (cached: the
        batch planner reads all words of all utterances as arrays)."""
        cached = self._times
        if self._word_list is None and cached is not None:
            return cached
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
        items = []
        for word in self._words:
            item = {'alignedWord': str(word), 'start': word.start(),
                    'end': word.end()}
            if getattr(word, 'phonemes', None):
                item['phonemes'] = [
                    {'phoneme': str(p), 'start': p.start(), 'end': p.end()}
                    for p in word.phonemes]
            items.append(item)
        return {'words': items}

    def save(self, file):
        file = str(file)
        if file.endswith('.json'):
            with open(file, 'w', encoding='utf-8') as out:
                json.dump(self.json(), out, indent=4)
        elif file.endswith('.TextGrid'):
            with open(file, 'w', encoding='utf-8') as out:
                out.write(_textgrid(self._words, self.tiers))
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
    words = []
    for item in content['words']:
        phonemes = None
        if item.get('phonemes'):
            phonemes = [
                Phoneme(p.get('phoneme', p.get('text')), p['start'], p['end'])
                for p in item['phonemes']]
        words.append(Word(item.get('alignedWord', item.get('word')),
                          item['start'], item['end'], phonemes))
    return words


def decode(data):
    """Text of a TextGrid file: UTF-16 (either byte order, with BOM - what
    Praat writes for non-ASCII text - or recognisable by the zero bytes of
    its ASCII header) or UTF-8 (with or without BOM)."""
    if data[:2] in (b'\xff\xfe', b'\xfe\xff'):
        return data.decode('utf-16')
    if data[:3] == b'\xef\xbb\xbf':
        return data[3:].decode('utf-8')
    if len(data) >= 4 and data[1] == 0 and data[0] != 0:
        return data.decode('utf-16-le')
    if len(data) >= 4 and data[0] == 0 and data[1] != 0:
        return data.decode('utf-16-be')
    return data.decode('utf-8')


# A TextGrid text file is a sequence of VALUES - numbers, "strings" (a quote
# inside is doubled), <flags> - with free text between them that Praat itself
# ignores: `xmin = `, `intervals [3]:`, `item []:` in the long format, nothing
# in the short one.  Reading the values in order handles both.
_VALUE = re.compile(
    r'"((?:[^"]|"")*)"|<(\w+)>|(?<![\w.\[])([-+]?(?:\d+\.?\d*|\.\d+)'
    r'(?:[eE][-+]?\d+)?)(?![\w\]])')


def _values(text):
    for match in _VALUE.finditer(text):
        string, flag, number = match.groups()
        if string is not None:
            yield string.replace('""', '"')
        elif flag is not None:
            yield ('flag', flag)
        else:
            yield float(number)


def _tiers_from_textgrid(text):
    """[(class, name, [(xmin, xmax, text)])] of a long- or short-format
    TextGrid (point tiers come back with xmin == xmax)."""
    values = _values(text)

    def take(kind):
        value = next(values)
        if not isinstance(value, kind):
            raise ValueError(f'TextGrid: expected {kind.__name__}, found '
                             f'{value!r}')
        return value
    try:
        if take(str) != 'ooTextFile' or take(str) != 'TextGrid':
            raise ValueError('not a TextGrid text file')
        take(float), take(float)                # xmin, xmax
        if take(tuple)[1] != 'exists':
            return []
        tiers = []
        for _ in range(int(take(float))):
            kind, name = take(str), take(str)
            take(float), take(float)
            count = int(take(float))
            if kind == 'IntervalTier':
                items = [(take(float), take(float), take(str))
                         for _ in range(count)]
            else:                               # TextTier: (time, mark)
                items = []
                for _ in range(count):
                    time = take(float)
                    items.append((time, time, take(str)))
            tiers.append((kind, name, items))
        return tiers
    except StopIteration:
        raise ValueError('TextGrid ends in the middle of a tier') from None


def _is_silence(text):
    return not text.strip() or text in ('sp', SILENCE)


def _words_from_textgrid(text):
    """Words (with their phonemes) of a Praat TextGrid -> (words, tiers).

    The word tier is the interval tier named "words" / "word" if there is
    one, otherwise the interval tier with the fewest intervals; the phoneme
    tier is the one named "phones" / "phone" / "phonemes", otherwise the
    finest other interval tier.  A phoneme belongs to the word that contains
    its midpoint (`pypar` walks both tiers in time order the same way)."""
    tiers = [(name, items) for kind, name, items in _tiers_from_textgrid(text)
             if kind == 'IntervalTier']
    if not tiers:
        raise ValueError('TextGrid holds no interval tiers')
    lowered = [name.lower() for name, _ in tiers]
    word_index = next((i for i, name in enumerate(lowered)
                       if name in ('words', 'word')), None)
    if word_index is None:
        word_index = min(range(len(tiers)), key=lambda i: len(tiers[i][1]))
    others = [i for i in range(len(tiers)) if i != word_index]
    phone_index = next((i for i in others if lowered[i] in (
        'phones', 'phone', 'phonemes', 'phoneme')), None)
    if phone_index is None and others:
        finest = max(others, key=lambda i: len(tiers[i][1]))
        if len(tiers[finest][1]) >= len(tiers[word_index][1]):
            phone_index = finest
    words = [
        Word(SILENCE if _is_silence(text) else text, a, b)
        for a, b, text in tiers[word_index][1]]
    names = Alignment.tiers
    if phone_index is not None:
        cursor = 0
        for word in words:
            word.phonemes = []
        for a, b, text in tiers[phone_index][1]:
            middle = 0.5 * (a + b)
            while cursor + 1 < len(words) and middle >= words[cursor].end():
                cursor += 1
            if words:
                words[cursor].phonemes.append(
                    Phoneme(SILENCE if _is_silence(text) else text, a, b))
        names = (tiers[word_index][0], tiers[phone_index][0],
                 phone_index < word_index)
    else:
        names = (tiers[word_index][0], names[1], False)
    return words, names


def _number(value):
    """Shortest text that reads back as the same float64 (0 -> "0")."""
    value = float(value)
    return repr(int(value)) if value == int(value) and abs(value) < 1e15 \
        else repr(value)


def _tier_lines(index, name, items, xmax):
    lines = [
        f'    item [{index}]:', '        class = "IntervalTier"',
        f'        name = "{name}"', '        xmin = 0',
        f'        xmax = {_number(xmax)}',
        f'        intervals: size = {len(items)}']
    for i, item in enumerate(items):
        text = '' if str(item) == SILENCE else str(item).replace('"', '""')
        lines += [
            f'        intervals [{i + 1}]:',
            f'            xmin = {_number(item.start())}',
            f'            xmax = {_number(item.end())}',
            f'            text = "{text}"']
    return lines


def _textgrid(words, tiers=Alignment.tiers):
    """Long-format TextGrid of the alignment: the word tier and, when the
    words carry phonemes, the phoneme tier (`core.py:111`: the reference saves
    the alignment it loaded, both tiers), under the names and in the order of
    the file they came from."""
    word_name, phone_name, phones_first = tiers
    xmax = words[-1].end() if words else 0.
    phonemes = [p for word in words
                for p in (getattr(word, 'phonemes', None) or [])]
    blocks = [(word_name, words)]
    if phonemes:
        blocks.insert(0 if phones_first else 1, (phone_name, phonemes))
    lines = [
        'File type = "ooTextFile"', 'Object class = "TextGrid"', '',
        'xmin = 0', f'xmax = {_number(xmax)}', 'tiers? <exists>',
        f'size = {len(blocks)}', 'item []:']
    for index, (name, items) in enumerate(blocks):
        lines += _tier_lines(index + 1, name, items, xmax)
    return '\n'.join(lines) + '\n'
