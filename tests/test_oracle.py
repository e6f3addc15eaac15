"""CPU: the oracle against the golden vectors captured from the reference
(tests/golden/generate.py), and the known answers of SURVEY.md App. E."""
import numpy as np
import pytest
import torch

from conftest import case_inputs, case_names, seconds, variant_config
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
                 weights.random_state(config, seed=7).items()}
        feats = oracle.features(
            padded[:, :audio.shape[1]], overrides, synth.pitch_tracks)[0]
        logits = oracle.forward(feats, bounds, state, overrides).numpy()
        want = variants[f'{name}/logits']
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(logits - want).max() < 2e-5 * scale, name
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
