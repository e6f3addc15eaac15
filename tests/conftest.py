import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs an MI355X (run with `-m gpu` on a GPU box)')


@pytest.fixture(scope='session')
def cases():
    return np.load(os.path.join(GOLDEN, 'cases.npz'))


@pytest.fixture(scope='session')
def chunk_goldens():
    return np.load(os.path.join(GOLDEN, 'chunks.npz'))


@pytest.fixture(scope='session')
def variants():
    return np.load(os.path.join(GOLDEN, 'variants.npz'))


@pytest.fixture(scope='session')
def seams():
    return np.load(os.path.join(GOLDEN, 'seams.npz'))


def case_names(archive):
    return sorted({key.split('/')[0] for key in archive.files if '/' in key})


def case_inputs(archive, name):
    """(audio float32 [1, S], bounds int64 [2, W], batch_size or None)."""
    from emphases_amd import synth
    if f'{name}/pcm' in archive.files:
        audio = synth.pcm_to_float(archive[f'{name}/pcm'])
    else:
        audio = archive[f'{name}/audio'][None]
    bounds = archive[f'{name}/bounds_frames'].astype(np.int64)
    batch_size = int(archive[f'{name}/batch_size'])
    return audio, bounds, None if batch_size < 0 else batch_size


def seconds(bounds):
    return [(int(s) / 100.0, int(e) / 100.0) for s, e in bounds.T]


def variant_config(name):
    """Config + overrides dict from a `key=value,...` variant name."""
    from emphases_amd import config as cfg
    overrides = {}
    for item in str(name).split(','):
        key, value = item.split('=')
        default = getattr(cfg.DEFAULT, key)
        if isinstance(default, bool):
            overrides[key] = value == 'True'
        elif isinstance(default, int):
            overrides[key] = int(value)
        else:
            overrides[key] = value
    return cfg.Config(**overrides), overrides


def variant_state(archive, name, config):
    """The seeded weights a variant's goldens were captured with: seed 7 and
    the output-layer gain `tests/golden/generate.py` chose for it (a power of
    two that keeps |logit| in (2, 4], so that the scores are not saturated)."""
    from emphases_amd import weights
    return weights.random_state(
        config, seed=7, output_gain=float(archive[f'{name}/output_gain']))
