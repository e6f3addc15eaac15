"""Deterministic synthetic workloads (audio, alignments, random weights).

Everything here is derived from a counter-mode SplitMix64 stream evaluated in
exact integer arithmetic, so the same bits come out on every machine and numpy
version — the GPU box regenerates the benchmark and parity inputs without any
file from the build container.  Shapes follow SURVEY.md §8(d):

* audio *i*: 5 harmonics of `f0 = 100 + 7 (i mod 20)` Hz under a 4 Hz
  raised-cosine envelope, plus uniform noise; every 10th utterance has a 0.5 s
  all-zero segment (exercises the 1e-6 / 1e-5 floors of `mels.py:51,109`);
  quantised to int16 so the float32 samples are exactly representable.
* alignment *i*: word lengths uniform in 8..60 frames, cumulative to exactly the
  utterance's frame count, every 8th token `<silent>`; times are
  `frames / 100` seconds.
"""
import numpy as np

from . import config as cfg

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, count):
    """`count` uint64 values of the SplitMix64 stream started at `seed`."""
    with np.errstate(over='ignore'):
        index = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + index * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(seed, count):
    """float64 uniform in [0, 1) with 53 random bits."""
    return (splitmix64(seed, count) >> np.uint64(11)).astype(np.float64) * \
        (1.0 / 9007199254740992.0)


def integers(seed, count, low, high):
    """Integers uniform in [low, high] (inclusive)."""
    span = np.uint64(high - low + 1)
    return (splitmix64(seed, count) % span).astype(np.int64) + low


def audio(index, frames):
    """Synthetic utterance `index` with `frames * 160` samples, float32 [1, S]
    holding exact multiples of 2**-15."""
    samples = frames * cfg.HOPSIZE
    n = np.arange(samples, dtype=np.float64)
    f0 = 100.0 + 7.0 * (index % 20)
    envelope = 0.55 - 0.45 * np.cos(2.0 * np.pi * 4.0 * n / cfg.SAMPLE_RATE)
    x = np.zeros(samples, dtype=np.float64)
    for h in range(1, 6):
        x += np.sin(2.0 * np.pi * h * f0 * n / cfg.SAMPLE_RATE) / h
    x = 0.08 * envelope * x
    x += 0.04 * (uniform(1000 + index, samples) - 0.5) * 1.7320508
    if index % 10 == 9 and samples >= 3 * 8000:
        start = samples // 3
        x[start:start + 8000] = 0.0
    pcm = np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16)
    return pcm_to_float(pcm)


def pcm_to_float(pcm):
    return (pcm.astype(np.float32) / np.float32(32768.0))[None]


def pitch_tracks(audio, *args, **kwargs):
    """Deterministic stand-in for `penn.from_audio` (a third-party neural
    pitch tracker the reference calls at `data/preprocess/core.py:84-92`):
    per-frame pitch in [80, 380) Hz and periodicity in [0, 1) derived from the
    frame energy of `audio` [1, S].  Same signature tail as penn so that it can
    be patched over it when the goldens are captured; returns torch tensors
    [1, S // 160] like penn with `pad=True`."""
    import torch
    x = np.asarray(
        audio.detach().cpu() if hasattr(audio, 'detach') else audio,
        dtype=np.float32).reshape(-1)
    frames = x.size // cfg.HOPSIZE
    x = x[:frames * cfg.HOPSIZE].reshape(frames, cfg.HOPSIZE).astype(np.float64)
    rms = np.sqrt((x * x).mean(axis=1))
    pitch = (80.0 + 300.0 * rms / (rms + 0.05)).astype(np.float32)
    periodicity = (rms / (rms + 0.02)).astype(np.float32)
    return torch.from_numpy(pitch)[None], torch.from_numpy(periodicity)[None]


def word_frames(index, frames, low=8, high=60):
    """Word boundaries in integer frames: int64 [2, W], gap-free, first start
    0, last end `frames`."""
    lengths = integers(2000 + index, frames // low + 1, low, high)
    ends = np.cumsum(lengths)
    count = int(np.searchsorted(ends, frames, side='left')) + 1
    ends = ends[:count].copy()
    ends[-1] = frames
    # Avoid a degenerate final word shorter than 2 frames
    if count > 1 and ends[-1] - ends[-2] < 2:
        ends = np.delete(ends, -2)
    starts = np.concatenate([[0], ends[:-1]])
    return np.stack([starts, ends]).astype(np.int64)


def word_names(count):
    return ['<silent>' if j % 8 == 7 else f'w{j}' for j in range(count)]


def corpus_frames(count, low=200, high=3000, seed=3000):
    """Frame counts of the mixed-length corpus (config C4: 2..30 s)."""
    return integers(seed, count, low, high)


def weights(seed, shape, bound):
    """float32 uniform(-bound, bound) tensor, exact in float64."""
    count = int(np.prod(shape))
    values = (2.0 * uniform(seed, count) - 1.0) * bound
    return values.astype(np.float32).reshape(shape)
