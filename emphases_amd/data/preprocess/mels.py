"""`emphases/data/preprocess/mels.py:16-59` on the HIP front-end
(`emph_logmel`: reflect padding, STFT(1024, hop 160, periodic Hann),
magnitude, `librosa.filters.mel` basis, log - one kernel)."""
import dataclasses

from ... import core
from . import core as preprocess


def from_audio(audio):
    """Compute the log-mel spectrogram of audio [1, S] (16 kHz): float32
    [80, F], on the device the audio is on (a host tensor is featurised on the
    current HIP device and comes back on the host).  `NORMALIZE` of the active
    configuration applies (`mels.py:56-58`)."""
    active = core.active_config()
    config = dataclasses.replace(
        active, mel_feature=True, pitch_feature=False,
        periodicity_feature=False, loudness_feature=False)
    result, _ = preprocess.features(audio, None, config)
    return result if audio.is_cuda else result.cpu()
