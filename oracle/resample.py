"""ORACLE — CPU restatement of the reference's sample-rate conversion
(`emphases/core.py:613-619` -> `torchaudio.transforms.Resample(orig, 16000)`
with its defaults).  TEST INFRASTRUCTURE ONLY: nothing in the product imports
this file.

PARITY UNPINNED for the third-party half: torchaudio (the reference's
`requirements`: unpinned `torchaudio`) is absent from /root/reference and from
the image, so its published algorithm (`torchaudio.functional.resample`:
`sinc_interp_hann`, `lowpass_filter_width=6`, `rolloff=0.99`) is restated here
and anchored on closed-form known answers (tests/test_oracle.py: DC gain, the
amplitude and phase of a sinusoid below the cut-off, output lengths).

Deliberately NOT the product's formulation: `emphases_amd/load.py` and
`csrc/features.hip` build a polyphase table `[new][2 width + orig]` with torch
ops and run it as a strided correlation.  Here every output sample is
evaluated on its own, in float64 numpy, from the continuous-time definition:

    rates reduced by their gcd: orig, new
    cut-off  base = min(orig, new) * rolloff            (cycles per `orig` unit)
    output sample m sits at time m / new; input sample n at n / orig
    y[m] = (base / orig) * sum_n x[n] * sinc(base * d) * hann(base * d),
           d = n / orig - m / new,
           sinc(u) = sin(pi u) / (pi u),  hann(u) = (1 + cos(pi u / 6)) / 2
           for |u| < 6, else 0,
    over the taps torchaudio's kernel holds for that output: with i = m // new,
    n from i * orig - width to i * orig + width + orig - 1,
    width = ceil(6 * orig / base); samples outside the signal are zero;
    len(y) = ceil(new * len(x) / orig).

torchaudio forms the weights in float64 and rounds them to float32 before a
float32 correlation; `weights_float32=True` (the default) applies the same
rounding, and the sum is accumulated in float64, so what is left between this
and a float32 implementation is accumulation order only (~1e-7).
"""
import math

import numpy as np

LOWPASS_FILTER_WIDTH = 6
ROLLOFF = 0.99


def reduced(sample_rate, target_rate):
    divisor = math.gcd(int(sample_rate), int(target_rate))
    return int(sample_rate) // divisor, int(target_rate) // divisor


def output_length(length, sample_rate, target_rate):
    orig, new = reduced(sample_rate, target_rate)
    return -((-new * int(length)) // orig)          # ceil, exact in integers


def tap_weight(distance, base, orig):
    """Weight of an input sample `distance` (= n / orig - m / new, float64
    array) away from an output sample."""
    u = np.clip(distance * base, -LOWPASS_FILTER_WIDTH, LOWPASS_FILTER_WIDTH)
    window = 0.5 * (1. + np.cos(np.pi * u / LOWPASS_FILTER_WIDTH))
    with np.errstate(invalid='ignore', divide='ignore'):
        sinc = np.where(u == 0., 1., np.sin(np.pi * u) / (np.pi * u))
    return sinc * window * (base / orig)


def resample(audio, sample_rate, target_rate=16000, weights_float32=True):
    """1-D float array at `sample_rate` -> float64 array at `target_rate`."""
    audio = np.asarray(audio, dtype=np.float64).reshape(-1)
    if int(sample_rate) == int(target_rate):
        return audio.copy()
    orig, new = reduced(sample_rate, target_rate)
    base = min(orig, new) * ROLLOFF
    width = int(math.ceil(LOWPASS_FILTER_WIDTH * orig / base))
    count = output_length(len(audio), sample_rate, target_rate)
    m = np.arange(count, dtype=np.int64)
    block = m // new
    taps = np.arange(-width, width + orig, dtype=np.int64)
    result = np.zeros(count, dtype=np.float64)
    # (rows of outputs at a time: [rows, taps] float64 temporaries)
    rows = max(1, (1 << 22) // len(taps))
    for lo in range(0, count, rows):
        hi = min(lo + rows, count)
        n = block[lo:hi, None] * orig + taps[None]
        # d = n / orig - m / new with the whole blocks cancelled exactly
        phase = m[lo:hi, None] - block[lo:hi, None] * new
        distance = taps[None] / float(orig) - phase / float(new)
        weight = tap_weight(distance, base, orig)
        if weights_float32:
            weight = weight.astype(np.float32).astype(np.float64)
        inside = (n >= 0) & (n < len(audio))
        samples = np.where(inside, audio[np.clip(n, 0, len(audio) - 1)], 0.) \
            if len(audio) else np.zeros_like(weight)
        result[lo:hi] = (weight * samples).sum(axis=1)
    return result
