"""Mel filterbank of the log-mel front-end and its packed sparse form.

The reference builds its basis with `librosa.filters.mel(sr=16000, n_fft=1024,
n_mels=80)` on every call (`emphases/data/preprocess/mels.py:94-103`).  librosa
is third-party and absent here, so the filterbank is restated from its
published definition: triangular filters on the Slaney mel scale (linear below
1 kHz, log above), Slaney area normalisation, float64 arithmetic, float32
result.  Each row is a contiguous run of non-zero bins (1001 non-zeros in all
for the default geometry; the Nyquist column is empty), which is what the HIP
front-end consumes: `row_start`, `row_count`, `row_offset` and the run values.
"""
import functools

import numpy as np

from . import config as cfg

_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(hz):
    hz = np.asarray(hz, dtype=np.float64)
    log_part = _MIN_LOG_MEL + \
        np.log(np.maximum(hz, 1e-300) / _MIN_LOG_HZ) / _LOGSTEP
    return np.where(hz >= _MIN_LOG_HZ, log_part, hz / _F_SP)


def mel_to_hz(mel):
    mel = np.asarray(mel, dtype=np.float64)
    log_part = _MIN_LOG_HZ * np.exp(_LOGSTEP * (mel - _MIN_LOG_MEL))
    return np.where(mel >= _MIN_LOG_MEL, log_part, _F_SP * mel)


def filterbank(sample_rate=cfg.SAMPLE_RATE, n_fft=cfg.NUM_FFT,
               n_mels=cfg.NUM_MELS, fmin=0.0, fmax=None):
    """Dense float32 [n_mels, n_fft // 2 + 1] basis."""
    fmax = sample_rate / 2.0 if fmax is None else fmax
    bins = np.fft.rfftfreq(n=n_fft, d=1.0 / sample_rate)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    widths = np.diff(edges)
    # Distance of every bin to every band edge: [n_mels + 2, bins]
    ramps = edges[:, None] - bins[None]
    rising = -ramps[:-2] / widths[:-1, None]
    falling = ramps[2:] / widths[1:, None]
    triangles = np.maximum(0, np.minimum(rising, falling)).astype(np.float32)
    area = 2.0 / (edges[2:] - edges[:-2])
    # float32 triangle times float64 norm, rounded once to float32
    return (triangles.astype(np.float64) * area[:, None]).astype(np.float32)


# Lanes the LDS serves together for a 16-byte read (`ds_read_b128`: four groups of
# sixteen lanes, one cycle each when their 16-byte chunks fall into sixteen
# different bank quads or coincide; MI355X_MICROARCH.md, LDS table).
_READ_GROUPS = (
    tuple(range(0, 4)) + tuple(range(12, 16)) + tuple(range(20, 28)),
    tuple(range(4, 12)) + tuple(range(16, 20)) + tuple(range(28, 32)),
    tuple(range(32, 36)) + tuple(range(44, 48)) + tuple(range(52, 60)),
    tuple(range(36, 44)) + tuple(range(48, 52)) + tuple(range(60, 64)))
_RUN_A, _RUN_B = 24, 12          # csrc/frontend.hip: kRunA, kRunB (floats per lane)


def _read_cycles(chunks):
    """LDS cycles of one 16-byte read per lane, `chunks[lane]` = the 16-byte
    chunk of the magnitude buffer the lane reads."""
    total = 0
    for group in _READ_GROUPS:
        banks = {}
        for lane in group:
            banks.setdefault(chunks[lane] % 16, set()).add(chunks[lane])
        total += max(len(addresses) for addresses in banks.values())
    return total


def _descend(cost, slack, floor, trials=40000):
    """Integer shifts 0..slack[i] minimising `cost`: coordinate descent, then
    seeded random moves over plateaus until `floor` (the conflict-free count) is
    reached or the trials run out.  Deterministic."""
    shifts = [0] * len(slack)
    best = cost(shifts)
    improved = True
    while improved and best > floor:
        improved = False
        for index in range(len(slack)):
            keep = shifts[index]
            for value in range(slack[index] + 1):
                shifts[index] = value
                now = cost(shifts)
                if now < best:
                    best, keep, improved = now, value, True
            shifts[index] = keep
    random = np.random.RandomState(0)
    movable = [index for index, room in enumerate(slack) if room > 0]
    for _ in range(trials):
        if best <= floor or not movable:
            break
        candidate = list(shifts)
        for _ in range(random.randint(1, 4)):
            index = movable[random.randint(len(movable))]
            candidate[index] = random.randint(slack[index] + 1)
        now = cost(candidate)
        if now <= best:
            shifts, best = candidate, now
    return shifts, best


def conflict_free_starts(first, count):
    """Where each row's run of the front-end's mel projection may START.

    A lane of the front-end reads its row's magnitudes as fixed-length runs of
    16-byte LDS reads from a 16-byte aligned start at or before the row's first
    bin (rows 0..63: one lane, 24 floats; rows 64..79: four lanes, 12 floats
    each); the weights in front of the first bin are zero.  Starting a run a few
    chunks EARLY is therefore free, and the starts can be chosen so that the
    sixteen lanes the LDS serves together hit sixteen different bank quads: 36
    LDS cycles per frame for the default basis instead of 69 (`first & ~3`).
    Returns (starts int32 [80], cycles per frame)."""
    first = [int(f) for f in first]
    count = [int(c) for c in count]
    base_a = [first[row] >> 2 for row in range(64)]
    slack_a = [max(0, min(base_a[row], (_RUN_A - (first[row] & 3) - count[row]) // 4))
               for row in range(64)]
    shifts_a, cycles_a = _descend(
        lambda s: _read_cycles([base_a[l] - s[l] for l in range(64)]), slack_a,
        len(_READ_GROUPS))
    base_b = [first[64 + row] >> 2 for row in range(16)]
    slack_b = [max(0, min(base_b[row], (4 * _RUN_B - (first[64 + row] & 3) -
                                        count[64 + row]) // 4))
               for row in range(16)]
    shifts_b, cycles_b = _descend(
        lambda s: _read_cycles([base_b[l >> 2] - s[l >> 2] + (_RUN_B // 4) * (l & 3)
                                for l in range(64)]), slack_b, len(_READ_GROUPS))
    starts = [4 * (base_a[row] - shifts_a[row]) for row in range(64)] + \
        [4 * (base_b[row] - shifts_b[row]) for row in range(16)]
    return np.array(starts, dtype=np.int32), \
        (_RUN_A // 4) * cycles_a + (_RUN_B // 4) * cycles_b


class SparseBasis:
    """Row-run packing of a filterbank whose rows are contiguous runs.

    `row_start` / `row_count` / `row_offset` / `values` are what the HIP
    front-end consumes.  With `aligned` (the default) every run starts at the
    16-byte aligned bin `conflict_free_starts` picked, padded with explicit zero
    weights up to the row's first non-zero bin: the same products in the same
    order (a zero weight adds nothing), without LDS bank conflicts."""

    def __init__(self, basis, aligned=True):
        basis = np.asarray(basis, dtype=np.float32)
        rows, _ = basis.shape
        self.dense = basis
        lows = np.zeros(rows, dtype=np.int64)
        highs = np.zeros(rows, dtype=np.int64)
        for row in range(rows):
            nonzero = np.flatnonzero(basis[row])
            if nonzero.size:
                lows[row], highs[row] = int(nonzero[0]), int(nonzero[-1]) + 1
        # limits of the HIP front-end (csrc/frontend.hip: kRunA, kRunB)
        if rows != 80 or (highs - lows)[:64].max() > 20 or \
                (highs - lows)[64:].max() > 40:
            raise ValueError(
                'the front-end kernel needs 80 filterbank rows whose runs hold '
                'at most 20 bins (rows 0..63) / 40 bins (rows 64..79)')
        self.read_cycles = None
        starts = lows.copy()
        if aligned:
            starts, self.read_cycles = conflict_free_starts(lows, highs - lows)
        self.row_start = np.zeros(rows, dtype=np.int32)
        self.row_count = np.zeros(rows, dtype=np.int32)
        self.row_offset = np.zeros(rows, dtype=np.int32)
        values = []
        offset = 0
        for row in range(rows):
            lo, hi = int(min(starts[row], lows[row])), int(highs[row])
            if hi == lows[row]:
                lo = hi = 0                       # an empty row
            self.row_start[row] = lo
            self.row_count[row] = hi - lo
            self.row_offset[row] = offset
            values.append(basis[row, lo:hi])
            offset += hi - lo
        self.values = np.concatenate(values).astype(np.float32)
        self.max_count = int(self.row_count.max())

    @property
    def nnz(self):
        return int(np.count_nonzero(self.dense))


@functools.lru_cache(maxsize=None)
def default():
    """Cached default basis (the reference's cache is broken by an attribute
    typo, `mels.py:96` vs `mels.py:103`; same values, built once here)."""
    return SparseBasis(filterbank())
