"""Restatement of the few librosa functions the reference's hot path calls.

Written from librosa's documented definitions (Slaney mel scale, Slaney area
normalisation, amplitude_to_db, A_weighting); librosa itself is not installed,
so these are parity-unpinned at the library boundary.
"""
import numpy as np

from . import filters


def fft_frequencies(sr=22050, n_fft=2048):
    return np.fft.rfftfreq(n=n_fft, d=1.0 / sr)


def stft(y, n_fft=2048, hop_length=None, win_length=None, center=True,
         pad_mode='constant'):
    win_length = n_fft if win_length is None else win_length
    hop_length = win_length // 4 if hop_length is None else hop_length
    assert win_length == n_fft
    y = np.asarray(y, dtype=np.float32)
    if center:
        y = np.pad(y, n_fft // 2, mode=pad_mode)
    n = np.arange(n_fft)
    # scipy.signal.get_window('hann', fftbins=True): periodic Hann
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)
    frames = 1 + (len(y) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(frames)[None]
    return np.fft.rfft(window[:, None] * y[idx], axis=0).astype(np.complex64)


def amplitude_to_db(S, ref=1.0, amin=1e-5, top_db=80.0):
    magnitude = np.abs(np.asarray(S))
    power = np.square(magnitude, out=magnitude)
    log_spec = 10.0 * np.log10(np.maximum(amin ** 2, power))
    log_spec -= 10.0 * np.log10(np.maximum(amin ** 2, ref ** 2))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def A_weighting(frequencies, min_db=-80.0):
    f_sq = np.asanyarray(frequencies) ** 2.0
    const = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2.0
    weights = 2.0 + 20.0 * (
        np.log10(const[0]) + 2 * np.log10(f_sq)
        - np.log10(f_sq + const[0]) - np.log10(f_sq + const[1])
        - 0.5 * np.log10(f_sq + const[2]) - 0.5 * np.log10(f_sq + const[3]))
    return weights if min_db is None else np.maximum(min_db, weights)
