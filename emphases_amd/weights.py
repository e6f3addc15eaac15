"""Model parameters: state-dict layout, checkpoint reader, seeded random init.

Names and shapes follow the reference's `Model.state_dict()`
(`emphases/model/core.py:13-37`, `model/layers/convolution.py:21-33`,
`model/layers/transformer.py:15-23`; listed in SURVEY.md App. A.5).  The
reference loads `{'model': state_dict, ...}` pickles through
`torchutil.checkpoint.load` (`emphases/core.py:313`) and downloads a default
from HuggingFace when `checkpoint is None` (`core.py:307-310`); here the
default is the bundled copy of the reference's trained weights
(`assets/checkpoint.npz`, extracted from
`emphases/assets/checkpoints/checkpoint.pt`), so inference works offline.
"""
import collections
import math
import os

import numpy as np

from . import config as cfg
from . import synth

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets')
DEFAULT_CHECKPOINT = os.path.join(ASSETS, 'checkpoint.npz')


def _stack_shapes(shapes, prefix, config, kernel_size):
    channels = config.channels
    if config.architecture == 'convolution':
        for i in range(config.layers):
            shapes[f'{prefix}.{2 * i}.weight'] = \
                (channels, channels, kernel_size)
            shapes[f'{prefix}.{2 * i}.bias'] = (channels,)
    else:
        for i in range(config.layers):
            p = f'{prefix}.model.layers.{i}.'
            shapes[p + 'self_attn.in_proj_weight'] = (3 * channels, channels)
            shapes[p + 'self_attn.in_proj_bias'] = (3 * channels,)
            shapes[p + 'self_attn.out_proj.weight'] = (channels, channels)
            shapes[p + 'self_attn.out_proj.bias'] = (channels,)
            shapes[p + 'linear1.weight'] = (channels, channels)
            shapes[p + 'linear1.bias'] = (channels,)
            shapes[p + 'linear2.weight'] = (channels, channels)
            shapes[p + 'linear2.bias'] = (channels,)
            shapes[p + 'norm1.weight'] = (channels,)
            shapes[p + 'norm1.bias'] = (channels,)
            shapes[p + 'norm2.weight'] = (channels,)
            shapes[p + 'norm2.bias'] = (channels,)


def parameter_shapes(config=cfg.DEFAULT):
    """Ordered name -> shape of the learnable tensors (the transformer's
    `position.encoding` buffers are recomputed, not stored)."""
    shapes = collections.OrderedDict()
    shapes['input_layer.weight'] = (
        config.channels, config.num_features, config.encoder_kernel_size)
    shapes['input_layer.bias'] = (config.channels,)
    _stack_shapes(
        shapes, 'frame_encoder', config, config.encoder_kernel_size)
    if config.has_decoder:
        _stack_shapes(
            shapes, 'word_decoder', config, config.decoder_kernel_size)
    shapes['output_layer.weight'] = (
        1, config.channels, config.decoder_kernel_size)
    shapes['output_layer.bias'] = (1,)
    return shapes


def random_state(config=cfg.DEFAULT, seed=0, output_gain=1.0):
    """Seeded parameters (uniform, variance-preserving fan-in scale) for configurations that have
    no trained checkpoint — the reference ships none for the transformer or
    any hparam-search variant.  `output_gain` scales the
    output layer (a power of two scales the logits exactly): the variant
    goldens use it to keep |logit| within a few units, where a sigmoid still
    shows a difference (`tests/golden/generate.py`)."""
    state = collections.OrderedDict()
    for index, (name, shape) in enumerate(parameter_shapes(config).items()):
        stream = seed * 1000 + index
        if name.endswith('norm1.weight') or name.endswith('norm2.weight'):
            state[name] = 1.0 + synth.weights(stream, shape, 0.1)
        elif len(shape) == 1:
            state[name] = synth.weights(stream, shape, 0.1)
        else:
            fan_in = int(np.prod(shape[1:]))
            state[name] = synth.weights(stream, shape, math.sqrt(6.0 / fan_in))
    if output_gain != 1.0:
        for name in ('output_layer.weight', 'output_layer.bias'):
            state[name] = (state[name] * np.float32(output_gain)).astype(np.float32)
    return state


def load(checkpoint=None, config=cfg.DEFAULT):
    """Read a checkpoint into an ordered dict of float32 numpy arrays and
    check it against the configuration's layout."""
    if checkpoint is None:
        checkpoint = DEFAULT_CHECKPOINT
    if isinstance(checkpoint, dict):
        raw = checkpoint
    elif str(checkpoint).endswith('.npz'):
        with np.load(checkpoint) as file:
            raw = {name: file[name] for name in file.files}
    else:
        import torch
        raw = torch.load(checkpoint, map_location='cpu', weights_only=False)
        raw = raw['model'] if 'model' in raw else raw
    raw = _renumber(raw, config)
    state = collections.OrderedDict()
    for name, shape in parameter_shapes(config).items():
        if name not in raw:
            raise KeyError(f'checkpoint is missing parameter {name}')
        value = raw[name]
        value = value.detach().cpu().numpy() if hasattr(value, 'detach') \
            else np.asarray(value)
        if tuple(value.shape) != tuple(shape):
            raise ValueError(
                f'parameter {name} has shape {tuple(value.shape)}, '
                f'expected {tuple(shape)}')
        state[name] = np.ascontiguousarray(value, dtype=np.float32)
    return state


def _renumber(raw, config):
    """Map the Sequential indices of a conv stack onto layer numbers.  The
    reference's `Convolution` is `Sequential(conv, activation[, Dropout])` per
    layer (`convolution.py:25-33`): checkpoints of the dropout configs
    (`config/hparam-search/dropout-{05,10}.py`) keep layer i at index 3 i, not
    2 i.  Dropout is the identity at inference, so only the names differ."""
    if config.architecture != 'convolution':
        return raw
    import re
    renamed = dict(raw)
    for prefix in ('frame_encoder', 'word_decoder'):
        pattern = re.compile(rf'^{prefix}\.(\d+)\.weight$')
        indices = sorted(
            int(m.group(1)) for m in map(pattern.match, raw) if m)
        if indices == [2 * i for i in range(len(indices))]:
            continue
        for layer, index in enumerate(indices):
            for kind in ('weight', 'bias'):
                old = f'{prefix}.{index}.{kind}'
                renamed.pop(old, None)
            for kind in ('weight', 'bias'):
                old = f'{prefix}.{index}.{kind}'
                if old in raw:
                    renamed[f'{prefix}.{2 * layer}.{kind}'] = raw[old]
    return renamed


def positional_encoding(length, channels):
    """Sinusoidal table of `transformer.py:43-48`, float32 [length, channels],
    evaluated with the same float32 operations as the reference (arange *
    exp(...) products in float32, then sin/cos)."""
    import torch
    index = torch.arange(length).unsqueeze(1)
    frequency = torch.exp(
        torch.arange(0, channels, 2) * (-math.log(10000.0) / channels))
    table = torch.zeros(length, channels)
    table[:, 0::2] = torch.sin(index * frequency)
    table[:, 1::2] = torch.cos(index * frequency)
    return table.numpy()
