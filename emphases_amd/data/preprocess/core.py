"""`emphases/data/preprocess/core.py:71-125` on the device."""
import numpy as np
import torch

from ... import batch
from ... import config as cfg
from ... import core
from ... import runtime


def whole_audio_plan(samples):
    """A `batch.Plan` whose one segment is ALL of an audio of `samples`
    samples - what `mels.from_audio(audio)` of the reference sees - i.e. the
    slice [432, 432 + samples) of the zero-padded signal (`core.py:357-358`)."""
    if samples <= cfg.PADDING:
        # torch's reflect padding: "Padding size should be less than the
        # corresponding input dimension" (mels.py:31-36)
        raise RuntimeError(
            f'reflect padding of {cfg.PADDING} needs more than {cfg.PADDING} '
            f'samples, the audio has {samples}')
    frames = 1 + (samples + 2 * cfg.PADDING - cfg.NUM_FFT) // cfg.HOPSIZE
    segment = batch.Segment(0, 0, 0, cfg.PADDING, samples, frames,
                            np.zeros((2, 0), dtype=np.int64))
    return batch.Plan([segment], [0], [samples])


def _tracks(config, engine, plan, audio, pitch_tracker, gpu):
    """penn's pitch and periodicity of the whole audio on the packed frame
    axis (`data/preprocess/core.py:84-92,107-116`), or None."""
    if not (config.pitch_feature or config.periodicity_feature):
        return None
    tracker = pitch_tracker or core.penn_tracker(gpu)
    pitch, periodicity = tracker(
        batch.chunk_audio(audio.cpu(), plan.segments[0]))
    # penn's floating-point hopsize can yield one frame more than the integer
    # hopsize of the mels: the reference drops it (core.py:107-116)
    frames = int(plan.frames[0])
    if pitch.shape[-1] == frames + 1:
        pitch, periodicity = pitch[..., :-1], periodicity[..., :-1]
    pairs = [(pitch, periodicity)]
    packed = torch.from_numpy(batch.pack_tracks(plan, pairs))
    return packed.pin_memory().to(engine.device, non_blocking=True)


def features(audio, gpu=None, config=None, pitch_tracker=None):
    """(float32 [NUM_FEATURES, F] on the device - a fresh tensor, device)."""
    if audio.dim() == 2:
        audio = audio[0]                        # channel 0 only (mels.py:48)
    if audio.dim() != 1:
        raise ValueError('audio must be [1, samples] or [samples]')
    if gpu is None and audio.is_cuda:
        gpu = audio.device.index
    config = config or core.active_config()
    # (the front-end's constants live in an engine; which model it holds does
    # not matter here, and the default one always loads)
    engine = core.get_engine(None, gpu, cfg.DEFAULT)
    plan = whole_audio_plan(int(audio.shape[0]))
    if audio.dtype != torch.int16:              # (int16 = 16-bit PCM, as elsewhere)
        audio = audio.to(torch.float32)
    with torch.cuda.device(engine.device), engine.lock:
        resident = audio.to(engine.device).contiguous()
        meta = engine.upload(plan)
        tracks = _tracks(config, engine, plan, audio, pitch_tracker, gpu)
        out = engine.features(resident, plan, meta, tracks=tracks,
                              config=config)
        first = int(plan.frame_off[0])
        return out[:, first:first + int(plan.frames[0])].clone(), engine.device


def from_audio(audio, gpu=None, pitch_tracker=None):
    """Preprocess one audio file (`data/preprocess/core.py:71-125`): audio
    [1, S] at 16 kHz -> the active configuration's feature stack
    [1, NUM_FEATURES, F], F = S // 160 for S a multiple of 160 - mels, then
    log2 pitch and periodicity (from `pitch_tracker`, the stand-in for
    `penn.from_audio`, see `core.from_alignments_and_audios`), then loudness.
    Computed on GPU `gpu` (None: the current HIP device - there is no CPU
    path - and the result comes back on the host, where the reference would
    have computed it)."""
    result, _ = features(audio, gpu, core.active_config(), pitch_tracker)
    result = result[None]
    return result.cpu() if gpu is None and not audio.is_cuda else result
