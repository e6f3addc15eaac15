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


class SparseBasis:
    """Row-run packing of a filterbank whose rows are contiguous runs."""

    def __init__(self, basis):
        basis = np.asarray(basis, dtype=np.float32)
        rows, _ = basis.shape
        self.dense = basis
        self.row_start = np.zeros(rows, dtype=np.int32)
        self.row_count = np.zeros(rows, dtype=np.int32)
        self.row_offset = np.zeros(rows, dtype=np.int32)
        values = []
        offset = 0
        for row in range(rows):
            nonzero = np.flatnonzero(basis[row])
            if nonzero.size:
                lo, hi = int(nonzero[0]), int(nonzero[-1]) + 1
            else:
                lo = hi = 0
            self.row_start[row] = lo
            self.row_count[row] = hi - lo
            self.row_offset[row] = offset
            values.append(basis[row, lo:hi])
            offset += hi - lo
        self.values = np.concatenate(values).astype(np.float32)
        self.max_count = int(self.row_count.max())
        # limits of the HIP front-end (csrc/frontend.hip: kRunA, kRunB)
        if rows != 80 or self.row_count[:64].max() > 20 or \
                self.row_count[64:].max() > 40:
            raise ValueError(
                'the front-end kernel needs 80 filterbank rows whose runs hold '
                'at most 20 bins (rows 0..63) / 40 bins (rows 64..79)')

    @property
    def nnz(self):
        return int(np.count_nonzero(self.dense))


@functools.lru_cache(maxsize=None)
def default():
    """Cached default basis (the reference's cache is broken by an attribute
    typo, `mels.py:96` vs `mels.py:103`; same values, built once here)."""
    return SparseBasis(filterbank())
