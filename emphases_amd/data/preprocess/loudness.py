"""`emphases/data/preprocess/loudness.py:84-120` on the HIP front-end: the
A-weighted per-frame loudness row (`emph_frontend_peak` for `amplitude_to_db`'s
global `top_db` floor, then the loudness row of `emph_logmel`)."""
import dataclasses

from ... import config as cfg
from ... import core
from . import core as preprocess


def from_audio(audio, sample_rate=cfg.SAMPLE_RATE):
    """A-weighted loudness of audio [1, S]: float32 [1, F], on the device the
    audio is on (host in, host out).  The reference evaluates the weights on
    penn's 8 kHz grid whatever `sample_rate` says (`loudness.py:99-104`,
    SURVEY App. B.2); the audio itself must be at 16 kHz here, as in every
    caller of the reference."""
    if int(sample_rate) != cfg.SAMPLE_RATE:
        audio = core.resample(audio, sample_rate)
    active = core.active_config()
    config = dataclasses.replace(
        active, mel_feature=False, pitch_feature=False,
        periodicity_feature=False, loudness_feature=True)
    result, _ = preprocess.features(audio, None, config)
    return result if audio.is_cuda else result.cpu()
