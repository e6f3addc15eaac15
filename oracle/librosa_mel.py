"""ORACLE (test infrastructure, never shipped or measured as the product).

Restatement of `librosa.filters.mel` as called at
`emphases/data/preprocess/mels.py:97-100` (sr=16000, n_fft=1024, n_mels=80;
defaults fmin=0, fmax=sr/2, htk=False, norm='slaney', float32).  librosa is a
third-party dependency absent from /root/reference and from this image:
PARITY UNPINNED at this boundary; the only external anchor is the value in
librosa's documentation, mel(sr=22050, n_fft=2048)[0, 1] ~= 0.016 (this gives
0.016182853).
"""
import numpy as np


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(
        f >= min_log_hz,
        min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep,
        f / f_sp)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(
        m >= min_log_mel,
        min_log_hz * np.exp(logstep * (m - min_log_mel)),
        f_sp * m)


def mel(*, sr, n_fft, n_mels=128, fmin=0.0, fmax=None, dtype=np.float32):
    """Slaney-scale, Slaney-normalised triangular mel filterbank, built row by
    row the way librosa does (float32 rows, then an in-place scale)."""
    fmax = float(sr) / 2 if fmax is None else fmax
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=dtype)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = _mel_to_hz(
        np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights
