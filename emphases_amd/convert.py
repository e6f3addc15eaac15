"""Time-unit conversions, bit-faithful to `emphases/convert.py:9-36`.

The reference does these in Python float64 with *floor division*, and the
chunker depends on the exact results (e.g. 8.03 s -> frame 802, not 803), so
the same operations are applied in the same order here.
"""
from . import config as cfg


def frames_to_samples(frames):
    """convert.py:9-11"""
    return frames * cfg.HOPSIZE


def frames_to_seconds(frames):
    """convert.py:14-16"""
    return frames * cfg.HOPSIZE_SECONDS


def seconds_to_samples(seconds):
    """convert.py:24-26"""
    return seconds * cfg.SAMPLE_RATE


def samples_to_frames(samples):
    """convert.py:29-31 (floor division; float in -> float out)"""
    return samples // cfg.HOPSIZE


def seconds_to_frames(seconds):
    """convert.py:19-21"""
    return samples_to_frames(seconds_to_samples(seconds))


def samples_to_seconds(samples):
    """convert.py:34-36"""
    return samples / cfg.SAMPLE_RATE
