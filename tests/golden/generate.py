"""Capture golden vectors by running the UNMODIFIED reference in this container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate.py

Imports `emphases` from /root/reference with the third-party stand-ins of
`tests/golden/stubs/` (see its README) and, for every case below, records the
inputs and what the reference's own modules produce in float32 with autocast
disabled ("O-fp32", the parity oracle of SURVEY.md §8c) plus, for information,
what the shipped API returns under its bf16 autocast ("O-shipped").

Outputs (committed; data only, no reference source):
  tests/golden/cases.npz       default conv model, bundled checkpoint
  tests/golden/chunks.npz      chunk plans of `emphases.preprocess`
  tests/golden/variants.npz    config-variant matrix with seeded weights
  tests/golden/metrics.npz     the reference's own evaluate.metrics on seeded
                               ragged batches (`generate.py metrics` = only this)
  tests/golden/seams.npz       `emphases.segment` and `data.preprocess.from_audio`
                               / `mels.from_audio` / `loudness.from_audio` on whole
                               audios (`generate.py seams` = only this)
  emphases_amd/assets/checkpoint.npz   the reference's trained weights
The GPU box never runs this script; it only reads the .npz files.
"""
import hashlib
import math
import os
import sys

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = '/root/reference'
sys.path[:0] = [os.path.join(HERE, 'stubs'), REFERENCE, ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import emphases  # noqa: E402  (the reference)
import pypar  # noqa: E402  (stand-in)

from emphases_amd import config as acfg  # noqa: E402
from emphases_amd import synth, weights  # noqa: E402
from oracle import prominence as oracle  # noqa: E402

torch.set_num_threads(1)
CHECKPOINT = os.path.join(
    REFERENCE, 'emphases', 'assets', 'checkpoints', 'checkpoint.pt')


###############################################################################
# Reference drivers
###############################################################################


def make_alignment(bounds_seconds, names=None):
    names = names or synth.word_names(len(bounds_seconds))
    return pypar.Alignment([
        pypar.Word(name, start, end)
        for name, (start, end) in zip(names, bounds_seconds)])


def reference_model(state=None):
    model = emphases.Model()
    if state is None:
        state = torch.load(
            CHECKPOINT, map_location='cpu', weights_only=False)['model']
        model.load_state_dict(state)
    else:
        result = model.load_state_dict(
            {k: torch.from_numpy(v) for k, v in state.items()}, strict=False)
        assert not result.unexpected_keys, result
        assert all('position.encoding' in k for k in result.missing_keys), \
            result
    return model.eval()


def run_fp32(model, alignment, audio, batch_size=None):
    """O-fp32: the reference's `preprocess` and `Model`, autocast disabled."""
    stages = []
    captured = {}
    hooks = [
        model.input_layer.register_forward_hook(
            lambda m, i, o: captured.__setitem__('input_layer', o)),
        model.frame_encoder.register_forward_hook(
            lambda m, i, o: captured.__setitem__('encoder', o))]
    original = emphases.downsample

    def downsample(*args):
        result = original(*args)
        captured['downsampled'] = result
        return result
    emphases.downsample = downsample
    try:
        with torch.no_grad():
            for features, bounds in emphases.preprocess(
                    alignment, audio, emphases.SAMPLE_RATE, batch_size, None):
                captured.clear()
                frame_lengths = torch.tensor([features.shape[-1]])
                word_lengths = torch.tensor([bounds.shape[-1]])
                logits = model(features, frame_lengths, bounds, word_lengths)
                stages.append(dict(
                    features=features[0].numpy(),
                    bounds=bounds[0].numpy(),
                    logits=logits[0, 0].numpy(),
                    scores=emphases.postprocess(logits)[0, 0].numpy(),
                    **{k: v[0].numpy() for k, v in captured.items()}))
    finally:
        emphases.downsample = original
        for hook in hooks:
            hook.remove()
    return stages


def run_shipped(alignment, audio, batch_size=None):
    """O-shipped: the public API verbatim (bf16 autocast on CPU)."""
    for name in ('model', 'checkpoint', 'device_type'):
        if hasattr(emphases.infer, name):
            delattr(emphases.infer, name)
    return emphases.from_alignment_and_audio(
        alignment, audio, emphases.SAMPLE_RATE, CHECKPOINT, batch_size,
        None).float().numpy()


###############################################################################
# Cases
###############################################################################


def seconds(bounds_frames):
    return [(int(s) / 100.0, int(e) / 100.0) for s, e in bounds_frames.T]


def two_tone():
    """SURVEY.md App. E known-answer input."""
    n = np.arange(16000, dtype=np.float64)
    x = 0.1 * np.sin(2 * np.pi * 220 * n / 16000) + \
        0.05 * np.sin(2 * np.pi * 1000 * n / 16000)
    bounds = np.array([[0, 25, 50, 75], [25, 50, 75, 100]])
    return x.astype(np.float32)[None], bounds


def default_cases():
    cases = {}
    audio, bounds = two_tone()
    cases['two_tone_1s'] = (audio, bounds, None)
    cases['utt_2p5s'] = (
        synth.audio(3, 250), synth.word_frames(3, 250), None)
    cases['utt_10s'] = (
        synth.audio(0, 1000), synth.word_frames(0, 1000), None)
    cases['utt_silence_6s'] = (
        synth.audio(9, 600), synth.word_frames(9, 600), None)
    # words of 1 and 2 frames, and a leading stretch of digital silence
    audio = synth.audio(5, 300)
    audio[:, :16000] = 0.
    bounds = np.array([
        [0, 100, 101, 103, 150, 151, 220],
        [100, 101, 103, 150, 151, 220, 300]])
    cases['short_words_3s'] = (audio, bounds, None)
    # the float-floor quirk: 8.03 s -> frame 802 (convert.py:29-31)
    ends = [57, 203, 411, 803, 811, 819, 1000, 1206, 1606, 1613, 1615, 1622,
            1631, 1700]
    bounds = np.array([[0] + ends[:-1], ends])
    cases['float_floor_17s'] = (synth.audio(7, 1700), bounds, None)
    frames = 4118
    cases['chunked_41s_b500'] = (
        synth.audio(11, frames), synth.word_frames(11, frames, 8, 60), 500)
    cases['chunked_41s_b1000'] = (
        synth.audio(11, frames), synth.word_frames(11, frames, 8, 60), 1000)
    return cases


def capture_default(out):
    model = reference_model()
    state = {k: v.numpy() for k, v in model.state_dict().items()}
    worst = 0.
    for name, (audio, bounds, batch_size) in default_cases().items():
        audio_t = torch.from_numpy(audio)
        words = seconds(bounds)
        alignment = make_alignment(words)
        stages = run_fp32(model, alignment, audio_t, batch_size)
        shipped = run_shipped(alignment, audio_t, batch_size)
        scores = np.concatenate([s['scores'] for s in stages])
        pcm = np.rint(audio[0] * 32768.0)
        if np.array_equal(pcm / 32768.0, audio[0].astype(np.float64)):
            out[f'{name}/pcm'] = pcm.astype(np.int16)
        else:
            out[f'{name}/audio'] = audio[0]
        out[f'{name}/bounds_frames'] = bounds.astype(np.int32)
        out[f'{name}/batch_size'] = np.int64(
            -1 if batch_size is None else batch_size)
        out[f'{name}/scores'] = scores
        out[f'{name}/scores_shipped_bf16'] = shipped[0]
        out[f'{name}/chunk_frames'] = np.array(
            [s['features'].shape[-1] for s in stages], dtype=np.int32)
        out[f'{name}/chunk_words'] = np.array(
            [s['bounds'].shape[-1] for s in stages], dtype=np.int32)
        out[f'{name}/chunk_bounds'] = np.concatenate(
            [s['bounds'] for s in stages], axis=1).astype(np.int32)
        out[f'{name}/logits'] = np.concatenate(
            [s['logits'] for s in stages])
        out[f'{name}/downsampled'] = np.concatenate(
            [s['downsampled'] for s in stages], axis=1)
        mel = np.concatenate([s['features'] for s in stages], axis=1)
        encoder = np.concatenate([s['encoder'] for s in stages], axis=1)
        first = np.concatenate([s['input_layer'] for s in stages], axis=1)
        if mel.shape[1] <= 1000:
            out[f'{name}/mel'] = mel
            out[f'{name}/encoder'] = encoder
            out[f'{name}/input_layer'] = first
        else:
            # long cases: every 7th frame plus the chunk edges
            out[f'{name}/mel_stride7'] = mel[:, ::7]
            out[f'{name}/encoder_stride7'] = encoder[:, ::7]
        # the oracle must agree with the reference it restates
        mine = oracle.from_alignment_and_audio(
            words, audio_t, {k: torch.from_numpy(v) for k, v in state.items()},
            {}, batch_size)[0].numpy()
        delta = float(np.abs(mine - scores).max())
        worst = max(worst, delta)
        print(f'{name:22s} F={mel.shape[1]:5d} W={scores.size:4d} '
              f'chunks={len(stages):2d} |oracle-ref|={delta:.2e} '
              f'|shipped-ref|={np.abs(shipped[0] - scores).max():.2e}')
    assert worst < 2e-6, worst
    return state


###############################################################################
# Chunk plans (host arithmetic of core.py:345-418)
###############################################################################


def capture_chunks(out):
    plans = {
        'gapfree_b500': (synth.word_frames(21, 2600), 500),
        'gapfree_b1000': (synth.word_frames(21, 2600), 1000),
        'gapfree_none': (synth.word_frames(21, 2600), None),
        'gapfree_b64': (synth.word_frames(22, 700, 2, 40), 64),
        'floor_b300': (np.array([
            [0, 57, 203, 411, 803, 811, 819, 1000, 1206, 1606, 1613, 1615,
             1622, 1631],
            [57, 203, 411, 803, 811, 819, 1000, 1206, 1606, 1613, 1615, 1622,
             1631, 1700]]), 300),
        # a 2-frame chunk: shorter than the reflect pad -> dropped
        'short_chunk_b10': (np.array(
            [[0, 40, 42, 90], [40, 42, 90, 130]]), 10),
        'dropped_chunk_b0': (np.array(
            [[0, 40, 42, 90], [40, 42, 90, 130]]), 0),
    }
    for name, (bounds, batch_size) in plans.items():
        frames = int(bounds[1, -1])
        audio = torch.from_numpy(synth.audio(1, frames))
        alignment = make_alignment(seconds(bounds))
        produced = [
            (features.shape[-1], word_bounds[0].numpy())
            for features, word_bounds in emphases.preprocess(
                alignment, audio, 16000, batch_size, None)]
        out[f'{name}/bounds_frames'] = bounds.astype(np.int32)
        out[f'{name}/batch_size'] = np.int64(
            -1 if batch_size is None else batch_size)
        out[f'{name}/chunk_frames'] = np.array(
            [p[0] for p in produced], dtype=np.int32)
        out[f'{name}/chunk_words'] = np.array(
            [p[1].shape[-1] for p in produced], dtype=np.int32)
        out[f'{name}/chunk_bounds'] = np.concatenate(
            [p[1] for p in produced], axis=1).astype(np.int32)
        print(f'chunks {name:18s}', [
            (p[0], p[1].shape[-1]) for p in produced])


###############################################################################
# Variant matrix with seeded weights
###############################################################################


ACTIVATIONS = {
    'relu': torch.nn.ReLU, 'gelu': torch.nn.GELU, 'silu': torch.nn.SiLU,
    'leaky_relu': torch.nn.LeakyReLU}


def variant_list():
    variants = []
    for location in acfg.DOWNSAMPLE_LOCATIONS:
        for method in acfg.DOWNSAMPLE_METHODS:
            variants.append(dict(
                downsample_location=location, downsample_method=method))
    variants += [dict(activation=a) for a in ('gelu', 'silu', 'leaky_relu')]
    variants += [dict(encoder_kernel_size=k) for k in (5, 7)]
    variants += [dict(decoder_kernel_size=k) for k in (1, 5)]
    variants += [dict(channels=c) for c in (64, 128)]
    variants += [dict(layers=n) for n in (5, 7)]
    variants += [dict(loss='mse'), dict(normalize=True),
                 dict(loudness_feature=True),
                 dict(loudness_feature=True, normalize=True)]
    # pitch / periodicity rows: `penn.from_audio` is patched with
    # synth.pitch_tracks (the tracker itself is third-party and absent)
    variants += [dict(pitch_feature=True, periodicity_feature=True),
                 dict(pitch_feature=True, periodicity_feature=True,
                      normalize=True),
                 dict(pitch_feature=True, periodicity_feature=True,
                      loudness_feature=True),
                 dict(periodicity_feature=True)]
    variants += [dict(architecture='transformer'),
                 dict(architecture='transformer',
                      downsample_location='inference',
                      downsample_method='average'),
                 # key-padding mask over zero-padded word pieces
                 # (transformer.py:26-29 with model/core.py:41-87)
                 dict(architecture='transformer',
                      downsample_location='input'),
                 dict(architecture='transformer',
                      downsample_location='input',
                      downsample_method='center')]
    return variants


def variant_name(overrides):
    return ','.join(f'{k}={v}' for k, v in sorted(overrides.items())) or \
        'default'


def configure_reference(config):
    emphases.ARCHITECTURE = config.architecture
    emphases.ACTIVATION_FUNCTION = ACTIVATIONS[config.activation]
    emphases.CHANNELS = config.channels
    emphases.LAYERS = config.layers
    emphases.ENCODER_KERNEL_SIZE = config.encoder_kernel_size
    emphases.DECODER_KERNEL_SIZE = config.decoder_kernel_size
    emphases.DOWNSAMPLE_LOCATION = config.downsample_location
    emphases.DOWNSAMPLE_METHOD = config.downsample_method
    emphases.LOSS = config.loss
    emphases.NORMALIZE = config.normalize
    emphases.LOUDNESS_FEATURE = config.loudness_feature
    emphases.PITCH_FEATURE = config.pitch_feature
    emphases.PERIODICITY_FEATURE = config.periodicity_feature
    emphases.NUM_FEATURES = config.num_features


def capture_variants(out):
    import penn
    penn.from_audio = synth.pitch_tracks
    frames = 300
    audio = synth.audio(4, frames)
    bounds = synth.word_frames(4, frames, 3, 40)
    audio_t = torch.from_numpy(audio)
    alignment = make_alignment(seconds(bounds))
    out['audio_pcm'] = np.rint(audio[0] * 32768.0).astype(np.int16)
    out['bounds_frames'] = bounds.astype(np.int32)
    names = []
    for overrides in variant_list():
        config = acfg.Config(**overrides)
        name = variant_name(overrides)
        # Seeded weights put |logit| in the hundreds for several variants, where
        # the sigmoid saturates and a score comparison shows nothing: the
        # reference runs once to find the scale, then the output layer is
        # scaled by a power of two (exact in float32) that brings the largest
        # |logit| into (2, 4], and THAT state is the variant's
        configure_reference(config)
        probe = run_fp32(reference_model(weights.random_state(config, seed=7)),
                         alignment, audio_t)
        largest = float(np.abs(probe[0]['logits']).max())
        gain = float(2.0 ** math.floor(math.log2(4.0 / largest)))
        out[f'{name}/output_gain'] = np.float32(gain)
        state = weights.random_state(config, seed=7, output_gain=gain)
        model = reference_model(state)
        expected = {
            k: tuple(v.shape) for k, v in model.state_dict().items()
            if 'position.encoding' not in k}
        assert expected == {
            k: tuple(v) for k, v in
            weights.parameter_shapes(config).items()}, name
        stages = run_fp32(model, alignment, audio_t)
        assert len(stages) == 1
        out[f'{name}/logits'] = stages[0]['logits']
        out[f'{name}/scores'] = stages[0]['scores']
        if config.loudness_feature or config.pitch_feature or \
                config.periodicity_feature:
            out[f'{name}/features'] = stages[0]['features']
        cfg_dict = {k: getattr(config, k) for k in overrides}
        if True:
            mine = oracle.forward(
                torch.from_numpy(stages[0]['features']), stages[0]['bounds'],
                {k: torch.from_numpy(v) for k, v in state.items()},
                dict(cfg_dict)).numpy()
            delta = float(np.abs(mine - stages[0]['logits']).max())
            scale = float(np.abs(stages[0]['logits']).max())
            assert delta < 5e-6 * max(1., scale), (name, delta, scale)
        names.append(name)
        print(f'variant {name:60s} gain 2^{int(math.log2(gain)):+d} |oracle-ref|logit={delta:.2e} '
              f'range=[{stages[0]["logits"].min():.3f}, '
              f'{stages[0]["logits"].max():.3f}]')
    configure_reference(acfg.DEFAULT)
    out['names'] = np.array(names)


###############################################################################
# Evaluation metrics (emphases/evaluate/metrics.py:12-110)
###############################################################################


def metric_batches(loss, seed):
    """Seeded ragged batches [B, 1, W]: logits, targets, word_lengths.  Padding
    columns hold large finite garbage (the mask must drop them)."""
    generator = np.random.default_rng(seed)
    batches = []
    for lengths in ([37, 1, 12, 30, 5], [1], [64, 64], [3, 17, 2, 9, 40, 8, 1]):
        lengths = np.array(lengths, dtype=np.int64)
        shape = (len(lengths), 1, int(lengths.max()))
        if loss == 'bce':
            logits = generator.normal(0., 3., shape)
        else:   # raw scores around [0, 1], some outside (the clamp matters)
            logits = generator.uniform(-0.4, 1.4, shape)
            logits.flat[::7] = 0.            # log(0 + 1e-6) branch
            logits.flat[3::11] = 1.
        targets = generator.uniform(0., 1., shape)
        targets.flat[::13] = 0.
        targets.flat[5::17] = 1.
        for row, length in enumerate(lengths):
            logits[row, :, length:] = 1e4
            targets[row, :, length:] = -1e4
        batches.append((logits.astype(np.float32),
                        targets.astype(np.float32), lengths))
    return batches


def capture_metrics(out):
    """The reference's `Statistics` then `Metrics` exactly as
    `evaluate/core.py:28-52` chains them, per `LOSS`; the per-word values its
    first-party `update`s hand to the (stand-in) running averages are recorded
    too, so the first-party arithmetic is pinned element by element."""
    import torchutil
    metrics = emphases.evaluate.metrics
    for loss in ('bce', 'mse'):
        emphases.LOSS = loss
        batches = metric_batches(loss, 100 + len(loss))
        predicted_stats, target_stats = metrics.Statistics(), metrics.Statistics()
        for logits, targets, lengths in batches:
            scores = emphases.postprocess(torch.from_numpy(logits))
            target_stats.update(torch.from_numpy(targets), torch.from_numpy(lengths))
            predicted_stats.update(scores, torch.from_numpy(lengths))
        out[f'{loss}/predicted_stats'] = np.array(predicted_stats(), np.float64)
        out[f'{loss}/target_stats'] = np.array(target_stats(), np.float64)
        recorded = []
        original = torchutil.metrics.Average.update

        def update(self, values, count, original=original, recorded=recorded):
            recorded.append(values.detach().numpy().copy())
            original(self, values, count)
        torchutil.metrics.Average.update = update
        try:
            total = metrics.Metrics(predicted_stats, target_stats)
            for index, (logits, targets, lengths) in enumerate(batches):
                single = metrics.Metrics(predicted_stats, target_stats)
                for metric in (total, single):
                    metric.update(
                        torch.from_numpy(logits), torch.from_numpy(targets),
                        torch.from_numpy(lengths))
                out[f'{loss}/{index}/logits'] = logits
                out[f'{loss}/{index}/targets'] = targets
                out[f'{loss}/{index}/word_lengths'] = lengths
                # (total's bce, mse, then single's bce, mse)
                out[f'{loss}/{index}/bce_values'] = recorded[-4]
                out[f'{loss}/{index}/squared_errors'] = recorded[-3]
                result = single()
                out[f'{loss}/{index}/result'] = np.array(
                    [result['pearson_correlation'], result['bce'],
                     result['mse']], np.float64)
            result = total()
            out[f'{loss}/result'] = np.array(
                [result['pearson_correlation'], result['bce'], result['mse']],
                np.float64)
            out[f'{loss}/batches'] = np.int64(len(batches))
            print(f'metrics {loss}: stats p={predicted_stats()} '
                  f't={target_stats()} -> {result}')
        finally:
            torchutil.metrics.Average.update = original
    emphases.LOSS = acfg.DEFAULT.loss


###############################################################################
# The reference-named seams: segment, data.preprocess.from_audio, mels.from_audio
###############################################################################


def capture_seams(out):
    """`emphases.segment` (core.py:552-586) on seeded padded batches (columns
    beyond an item's word count repeat its last word, one item's words are
    one frame long), and `emphases.data.preprocess.from_audio` /
    `mels.from_audio` / `loudness.from_audio` (data/preprocess/core.py:71-125,
    mels.py:16-59, loudness.py:84-120) on WHOLE audios - no zero-pad-and-slice
    in front, unlike `emphases.preprocess` - under the default feature
    switches, with NORMALIZE, and with the loudness row."""
    generator = np.random.default_rng(2024)
    shapes = [(1, 3, 40, [5]), (3, 8, 120, [7, 2, 4]), (2, 80, 300, [12, 1])]
    for index, (items, channels, frames, words) in enumerate(shapes):
        width = max(words)
        xs = generator.normal(0., 1., (items, channels, frames)).astype(np.float32)
        bounds = np.zeros((items, 2, width), dtype=np.int64)
        for item, count in enumerate(words):
            if item == 1:       # one-frame words
                starts = np.sort(generator.choice(frames - 1, count, replace=False))
                ends = starts + 1
            else:
                edges = np.sort(generator.choice(
                    np.arange(1, frames), count, replace=False))
                starts = np.concatenate([[0], edges[:-1]])
                ends = edges
            bounds[item, 0, :count], bounds[item, 1, :count] = starts, ends
        result, result_bounds, result_lengths = emphases.segment(
            torch.from_numpy(xs), torch.from_numpy(bounds),
            torch.tensor(words))
        out[f'segment/{index}/xs'] = xs
        out[f'segment/{index}/word_bounds'] = bounds
        out[f'segment/{index}/word_lengths'] = np.array(words, dtype=np.int64)
        out[f'segment/{index}/result'] = result.numpy()
        out[f'segment/{index}/result_bounds'] = result_bounds.numpy()
        out[f'segment/{index}/result_lengths'] = result_lengths.numpy()
    out['segment/count'] = np.int64(len(shapes))
    audios = {
        'noise_1p3s': synth.audio(31, 130)[:, :20731],      # not a multiple of 160
        'utt_4s': synth.audio(32, 400),
        'short_433': synth.audio(33, 10)[:, :433]}          # the least reflect padding takes
    switches = {
        'default': dict(MEL_FEATURE=True, LOUDNESS_FEATURE=False, NORMALIZE=False),
        'normalized': dict(MEL_FEATURE=True, LOUDNESS_FEATURE=False, NORMALIZE=True),
        'mel_loudness': dict(MEL_FEATURE=True, LOUDNESS_FEATURE=True, NORMALIZE=False),
        'loudness_normalized': dict(
            MEL_FEATURE=False, LOUDNESS_FEATURE=True, NORMALIZE=True)}
    saved = {name: getattr(emphases, name)
             for name in ('MEL_FEATURE', 'LOUDNESS_FEATURE', 'NORMALIZE')}
    try:
        for name, audio in audios.items():
            out[f'audio/{name}'] = audio
            tensor = torch.from_numpy(audio)
            for tag, values in switches.items():
                for key, value in values.items():
                    setattr(emphases, key, value)
                out[f'from_audio/{name}/{tag}'] = \
                    emphases.data.preprocess.from_audio(tensor).numpy()
                if tag in ('default', 'normalized'):
                    out[f'mels/{name}/{tag}'] = \
                        emphases.data.preprocess.mels.from_audio(tensor).numpy()
                if tag == 'mel_loudness':
                    out[f'loudness/{name}'] = \
                        emphases.data.preprocess.loudness.from_audio(
                            tensor, emphases.SAMPLE_RATE).numpy()
    finally:
        for key, value in saved.items():
            setattr(emphases, key, value)
    out['audio/names'] = np.array(sorted(audios))
    out['from_audio/tags'] = np.array(sorted(switches))
    print('seams:', {k: v.shape for k, v in out.items() if k.startswith('from_audio/')})


###############################################################################
# Entry point
###############################################################################


def main():
    seams = {}
    capture_seams(seams)
    np.savez_compressed(os.path.join(HERE, 'seams.npz'), **seams)
    if sys.argv[1:] == ['seams']:
        return
    measures = {}
    capture_metrics(measures)
    np.savez_compressed(os.path.join(HERE, 'metrics.npz'), **measures)
    if sys.argv[1:] == ['metrics']:
        return
    default, chunk_plans, variants = {}, {}, {}
    state = capture_default(default)
    capture_chunks(chunk_plans)
    capture_variants(variants)

    flat = np.concatenate([v.ravel() for v in state.values()])
    digest = hashlib.sha256(flat.astype('<f4').tobytes()).hexdigest()
    print('weights sha256', digest)
    assert digest == \
        'be2fb6555ba5fab57a4abb3c4e55a6cefc9dbd03c3ccbdcc2c735784aa277e69'
    os.makedirs(os.path.join(ROOT, 'emphases_amd', 'assets'), exist_ok=True)
    np.savez(
        os.path.join(ROOT, 'emphases_amd', 'assets', 'checkpoint.npz'),
        **state)
    np.savez_compressed(os.path.join(HERE, 'cases.npz'), **default)
    np.savez_compressed(os.path.join(HERE, 'chunks.npz'), **chunk_plans)
    np.savez_compressed(os.path.join(HERE, 'variants.npz'), **variants)
    leaked = [
        root for root, dirs, _ in os.walk(REFERENCE) if '__pycache__' in dirs]
    assert not leaked, leaked


if __name__ == '__main__':
    main()
