"""CPU: the oracle against the golden vectors captured from the reference
(tests/golden/generate.py), and the known answers of SURVEY.md App. E."""
import math

import numpy as np
import pytest
import torch

from conftest import case_inputs, case_names, seconds, variant_config, \
    variant_state
from emphases_amd import synth, weights
from oracle import librosa_mel
from oracle import prominence as oracle

CASES = ['two_tone_1s', 'utt_2p5s', 'utt_10s', 'utt_silence_6s',
         'short_words_3s', 'float_floor_17s', 'chunked_41s_b500',
         'chunked_41s_b1000']


@pytest.fixture(scope='module')
def state():
    return {k: torch.from_numpy(v) for k, v in weights.load().items()}


def test_case_list_is_complete(cases):
    assert case_names(cases) == sorted(CASES)


@pytest.mark.parametrize('name', CASES)
def test_scores_match_reference(cases, state, name):
    audio, bounds, batch_size = case_inputs(cases, name)
    scores = oracle.from_alignment_and_audio(
        seconds(bounds), torch.from_numpy(audio), state, {}, batch_size)
    assert scores.dtype == torch.float32
    np.testing.assert_allclose(
        scores[0].numpy(), cases[f'{name}/scores'], rtol=0, atol=2e-6)
    # the shipped bf16 autocast path is only informational: ~2e-3 away
    shipped = cases[f'{name}/scores_shipped_bf16']
    assert np.abs(shipped - cases[f'{name}/scores']).max() < 5e-3


@pytest.mark.parametrize('name', ['two_tone_1s', 'utt_2p5s', 'short_words_3s'])
def test_stages_match_reference(cases, state, name):
    audio, bounds, _ = case_inputs(cases, name)
    padded = torch.nn.functional.pad(torch.from_numpy(audio), (432, 432))
    mel = oracle.logmel(padded[:, :audio.shape[1]])
    np.testing.assert_allclose(
        mel.numpy(), cases[f'{name}/mel'], rtol=0, atol=1e-5)
    stages = {}
    oracle.forward(mel, bounds, state, {}, stages)
    for key in ('input_layer', 'encoder', 'downsampled', 'logits'):
        np.testing.assert_allclose(
            stages[key].numpy(), cases[f'{name}/{key}'], rtol=0, atol=2e-5,
            err_msg=key)


def test_known_answers_of_the_survey(cases):
    """1 s two-tone signal, bundled checkpoint, fp32."""
    mel = cases['two_tone_1s/mel']
    np.testing.assert_allclose(
        mel[:4, 0], [-4.0047779, -3.8994355, -3.6789193, -3.4502010],
        atol=2e-6)
    np.testing.assert_allclose(
        mel[:4, 50], [-9.0457182, -8.5435104, -7.6494780, -6.2575841],
        atol=2e-5)
    np.testing.assert_allclose(
        cases['two_tone_1s/logits'],
        [-1.2813212, -1.3857896, -1.1205515, -2.7076979], atol=2e-6)
    np.testing.assert_allclose(
        cases['two_tone_1s/scores'],
        [0.2173254, 0.2000808, 0.2459090, 0.0625206], atol=1e-6)
    np.testing.assert_allclose(
        cases['two_tone_1s/scores_shipped_bf16'],
        [0.2158203, 0.2001953, 0.2460938, 0.0629883], atol=1e-6)


def test_mel_basis_known_answers():
    basis = librosa_mel.mel(sr=16000, n_fft=1024, n_mels=80)
    assert basis.shape == (80, 513) and basis.dtype == np.float32
    assert np.count_nonzero(basis) == 1001
    assert np.all(basis[:, 512] == 0)
    counts = (basis != 0).sum(1)
    assert counts.min() == 4 and counts.max() == 37
    assert 0.0622 < basis.sum(1).min() and basis.sum(1).max() < 0.0666
    # the one external anchor: the value printed in librosa's documentation
    doc = librosa_mel.mel(sr=22050, n_fft=2048, n_mels=128)
    assert abs(float(doc[0, 1]) - 0.016182853) < 1e-9


def test_loudness_known_answers():
    """SURVEY.md App. E loudness row of the two-tone signal."""
    n = np.arange(16000, dtype=np.float64)
    x = 0.1 * np.sin(2 * np.pi * 220 * n / 16000) + \
        0.05 * np.sin(2 * np.pi * 1000 * n / 16000)
    audio = torch.from_numpy(x.astype(np.float32))[None]
    chunk = torch.nn.functional.pad(audio, (432, 432))[:, :16000]
    loud = oracle.loudness(chunk)[0].numpy()
    np.testing.assert_allclose(
        loud[0:3], [-64.263748, -55.540615, -52.383636], atol=2e-4)
    np.testing.assert_allclose(
        loud[50:53], [-71.353615, -71.353981, -71.355431], atol=2e-4)
    assert abs(loud.mean() + 70.118088) < 2e-4
    silent = oracle.loudness(torch.zeros(1, 16000))[0].numpy()
    assert np.all(silent == -100.0)
    weights_ = oracle.a_weights()[:, 0]
    np.testing.assert_allclose(
        weights_[[0, 1, 2, 128, 512]],
        [-100.0, -98.32126992, -77.08839866, -19.99965554, -19.03635464],
        atol=1e-6)


def test_variant_matrix_matches_reference(variants):
    """Every config variant of SURVEY.md App. A.6 with seeded weights."""
    audio = torch.from_numpy(synth.pcm_to_float(variants['audio_pcm']))
    bounds = variants['bounds_frames'].astype(np.int64)
    padded = torch.nn.functional.pad(audio, (432, 432))
    checked = 0
    for name in variants['names']:
        config, overrides = variant_config(name)
        state = {k: torch.from_numpy(v) for k, v in
                 variant_state(variants, name, config).items()}
        feats = oracle.features(
            padded[:, :audio.shape[1]], overrides, synth.pitch_tracks)[0]
        logits = oracle.forward(feats, bounds, state, overrides).numpy()
        want = variants[f'{name}/logits']
        scale = float(np.abs(want).max())
        assert 2. < scale <= 4., (name, scale)        # (the goldens' output gain)
        assert np.abs(logits - want).max() < 5e-6 * scale, name
        scores = oracle.postprocess(torch.from_numpy(logits), config.loss)
        assert np.abs(scores.numpy() -
                      variants[f'{name}/scores']).max() < 2e-6, name
        if config.loudness_feature or config.pitch_feature or \
                config.periodicity_feature:
            np.testing.assert_allclose(
                feats.numpy(), variants[f'{name}/features'][0]
                if variants[f'{name}/features'].ndim == 3
                else variants[f'{name}/features'], atol=2e-4)
        checked += 1
    assert checked == len(variants["names"]) == 39


def test_metrics_restatement_known_answers():
    """oracle/metrics.py (emphases/evaluate/metrics.py:12-110) on values that
    can be checked by hand."""
    from oracle import metrics
    logits = torch.tensor([[[0., 2., -1.]], [[1., 9., 9.]]])
    targets = torch.tensor([[[0.5, 1., 0.]], [[0., 7., 7.]]])
    lengths = torch.tensor([3, 1])
    m = metrics.Metrics((0.5, 0.25), (0.25, 0.5), 'bce')
    m.update(logits, targets, lengths)
    p = torch.sigmoid(torch.tensor([0., 2., -1., 1.]))
    t = torch.tensor([0.5, 1., 0., 0.])
    want_bce = float(-(t * torch.log(p) + (1 - t) * torch.log(1 - p)).mean())
    got = m()
    assert abs(got['bce'] - want_bce) < 1e-6
    assert abs(got['mse'] - float(((p - t) ** 2).mean())) < 1e-7
    want_r = float(((p - 0.5) * (t - 0.25)).sum()) / 4 / (0.25 * 0.5)
    assert abs(got['pearson_correlation'] - want_r) < 1e-6
    mean, std = metrics.mean_std([1., 2., 3., 4.])
    assert mean == 2.5 and abs(std - 1.2909944487358056) < 1e-12
    clamped = metrics.Metrics((0., 1.), (0., 1.), 'mse')
    clamped.update(torch.tensor([[[2., -3.]]]), torch.tensor([[[1., 0.]]]),
                   torch.tensor([2]))
    assert abs(clamped()['mse']) < 1e-12          # clamp(2)=1, clamp(-3)=0


###############################################################################
# resampling (torchaudio is third-party and absent: the
# restatement is anchored on closed-form answers)
###############################################################################


def windowed_sinc_gain(nu):
    """Continuous-time frequency response of sinc(u) hann(u), |u| < 6, at nu =
    f / (cut-off): the integral of sin(pi u)/(pi u) (1 + cos(pi u / 6))/2
    cos(pi nu u) du over [-6, 6], written with sine integrals."""
    from scipy.special import sici
    a = 6 * np.pi
    si = lambda x: sici(x)[0]        # noqa: E731
    return (si(a * (1 + nu)) + si(a * (1 - nu))) / (2 * np.pi) + (
        si(a * (1 + nu + 1 / 6)) + si(a * (1 + nu - 1 / 6)) +
        si(a * (1 - nu + 1 / 6)) + si(a * (1 - nu - 1 / 6))) / (4 * np.pi)


RATES = [8000, 22050, 44100, 48000]


@pytest.mark.parametrize('rate', RATES)
def test_resample_known_answers(rate):
    """oracle/resample.py: output lengths, DC gain and the amplitude / phase
    of a 1 kHz sinusoid against the closed-form response of the Hann-windowed
    sinc (zero phase: the filter is symmetric).  When upsampling, the image of
    DC at the input rate sits in the transition band and leaves a 4e-4 ripple
    (a property of the filter, not of the restatement)."""
    from oracle import resample as oracle_resample
    for length in (0, 1, 5, rate // 3 + 17, rate):
        got = oracle_resample.resample(np.zeros(length), rate)
        assert len(got) == -(-16000 * length // rate) == \
            oracle_resample.output_length(length, rate, 16000)
    count = rate // 2
    interior = slice(2000, 6000)             # of the 8000 output samples
    dc = oracle_resample.resample(np.ones(count), rate)[interior]
    ripple = {8000: 5e-4, 22050: 1e-4}.get(rate, 5e-6)
    assert np.abs(dc - windowed_sinc_gain(0.)).max() < ripple
    tone = np.sin(2 * np.pi * 1000. * np.arange(count) / rate)
    got = oracle_resample.resample(tone, rate)
    cutoff = min(rate, 16000) * 0.99 / 2
    want = windowed_sinc_gain(1000. / cutoff) * np.sin(
        2 * np.pi * 1000. * np.arange(len(got)) / 16000.)
    assert np.abs(got[interior] - want[interior]).max() < 1e-4
    assert abs(windowed_sinc_gain(1000. / cutoff) - 1) < 5e-4
    # an impulse comes back as the filter itself (peak base / orig at its time)
    impulse = np.zeros(64 * rate // math.gcd(rate, 16000))
    impulse[len(impulse) // 2] = 1.
    response = oracle_resample.resample(impulse, rate)
    orig, new = oracle_resample.reduced(rate, 16000)
    assert abs(response.max() - np.float32(min(orig, new) * 0.99 / orig)) < 1e-7
    assert response.argmax() == (len(impulse) // 2) * new // orig


@pytest.mark.parametrize('rate', RATES)
def test_resample_table_matches_oracle(rate):
    """The product's polyphase TABLE (`emphases_amd.load.resample_kernel`,
    built on the host like every weight pack; `emph_resample` applies it on
    the device) against the independent restatement: the table is applied
    here, in the test, as the strided correlation it describes."""
    from emphases_amd import load
    from oracle import resample as oracle_resample
    kernel, orig, new, width = load.resample_kernel(rate)
    taps = kernel.reshape(new, -1).numpy().astype(np.float64)
    for index, length in enumerate((rate // 3 + 17, 5, 1)):
        audio = synth.weights(200 + index, (length,), 0.9)
        padded = np.concatenate(
            [np.zeros(width), audio.astype(np.float64),
             np.zeros(width + orig)])
        steps = (len(padded) - taps.shape[1]) // orig + 1
        windows = np.lib.stride_tricks.sliding_window_view(
            padded, taps.shape[1])[::orig][:steps]
        got = (windows @ taps.T).reshape(-1)[
            :load.resampled_length(length, orig, new)]
        want = oracle_resample.resample(audio, rate)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-6


###############################################################################
# evaluation metrics against the reference's own run (tests/golden/metrics.npz)
###############################################################################


@pytest.mark.parametrize('loss', ['bce', 'mse'])
def test_metrics_restatement_matches_reference(loss):
    """oracle/metrics.py against what the reference's own `Statistics` /
    `Metrics.update` (evaluate/metrics.py:12-110) produced on seeded ragged
    batches: per-word BCE and squared-error values, per-batch and running
    results."""
    import os
    from oracle import metrics
    golden = np.load(os.path.join(
        os.path.dirname(__file__), 'golden', 'metrics.npz'))
    count = int(golden[f'{loss}/batches'])
    batches = [tuple(torch.from_numpy(golden[f'{loss}/{i}/{key}'])
                     for key in ('logits', 'targets', 'word_lengths'))
               for i in range(count)]
    predicted, target = [], []
    for logits, targets, lengths in batches:
        mask = metrics.mask_from_lengths(lengths)
        predicted += metrics.postprocess(logits, loss)[mask].tolist()
        target += targets[mask].tolist()
    stats_p, stats_t = metrics.mean_std(predicted), metrics.mean_std(target)
    np.testing.assert_allclose(stats_p, golden[f'{loss}/predicted_stats'], rtol=1e-12)
    np.testing.assert_allclose(stats_t, golden[f'{loss}/target_stats'], rtol=1e-12)
    total = metrics.Metrics(stats_p, stats_t, loss)
    for index, (logits, targets, lengths) in enumerate(batches):
        single = metrics.Metrics(stats_p, stats_t, loss)
        single.update(logits, targets, lengths)
        total.update(logits, targets, lengths)
        got = single()
        want = golden[f'{loss}/{index}/result']
        np.testing.assert_allclose(
            [got['pearson_correlation'], got['bce'], got['mse']], want,
            rtol=2e-6, atol=1e-7)
        values = metrics.word_values(logits, targets, lengths, loss)
        assert np.array_equal(values[0].numpy(), golden[f'{loss}/{index}/bce_values'])
        assert np.array_equal(
            values[1].numpy(), golden[f'{loss}/{index}/squared_errors'])
    got = total()
    np.testing.assert_allclose(
        [got['pearson_correlation'], got['bce'], got['mse']],
        golden[f'{loss}/result'], rtol=2e-6, atol=1e-7)


def test_whole_audio_features_match_reference(seams):
    """`data.preprocess.from_audio` / `mels.from_audio` / `loudness.from_audio`
    of the reference on WHOLE audios (tests/golden/generate.py seams; no
    zero-pad-and-slice in front) against the oracle's restatement."""
    switches = {
        'default': {}, 'normalized': {'normalize': True},
        'mel_loudness': {'loudness_feature': True},
        'loudness_normalized': {'mel_feature': False, 'loudness_feature': True,
                                'normalize': True}}
    for name in seams['audio/names']:
        audio = torch.from_numpy(seams[f'audio/{name}'])
        for tag, overrides in switches.items():
            want = seams[f'from_audio/{name}/{tag}']
            got = oracle.features(audio, overrides).numpy()
            assert got.shape == want.shape
            mels = 80 if overrides.get('mel_feature', True) else 0
            np.testing.assert_allclose(
                got[:, :mels], want[:, :mels], rtol=0, atol=1e-5)
            if overrides.get('loudness_feature'):
                tolerance = 2e-5 if overrides.get('normalize') else 1e-3
                np.testing.assert_allclose(
                    got[:, -1], want[:, -1], rtol=0, atol=tolerance)
        np.testing.assert_array_equal(
            seams[f'mels/{name}/default'], seams[f'from_audio/{name}/default'][0])
        np.testing.assert_array_equal(
            seams[f'loudness/{name}'],
            seams[f'from_audio/{name}/mel_loudness'][0, 80:])
