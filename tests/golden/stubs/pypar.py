"""Minimal gap-free Word/Alignment with the subset of pypar's interface that
emphases/core.py:366-400 uses.  Slices report bounds relative to the slice."""

SILENCE = '<silent>'


class Word:
    def __init__(self, word, start, end):
        self.word = word
        self._start = float(start)
        self._end = float(end)

    def __str__(self):
        return self.word

    def start(self):
        return self._start

    def end(self):
        return self._end

    def duration(self):
        return self._end - self._start


class Alignment:
    def __init__(self, words, relative=False):
        self._words = list(words)
        self._relative = relative

    def __len__(self):
        return len(self._words)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return Alignment(self._words[idx], relative=True)
        return self._words[idx]

    def start(self):
        return self._words[0].start()

    def end(self):
        return self._words[-1].end()

    def duration(self):
        return self.end() - self.start()

    def words(self):
        return self._words

    def word_bounds(self, sample_rate, hopsize=1, silences=False):
        words = [
            w for w in self._words if silences or str(w) != SILENCE]
        origin = 0
        if self._relative and self._words:
            origin = int(self._words[0].start() * sample_rate / hopsize)
        return [
            (int(w.start() * sample_rate / hopsize) - origin,
             int(w.end() * sample_rate / hopsize) - origin)
            for w in words]

    def save(self, file):
        raise NotImplementedError
