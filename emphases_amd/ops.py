"""The operator seams of SURVEY §8b as `torch.library` custom ops over the C ABI:
`at::Tensor` in, `at::Tensor` out, ragged ("varlen") axes described by `cu_*`
prefix sums, so that a caller of the reference's ATen ops
(`/root/reference/emphases/model/core.py:93-138`) needs no `ctypes`:

    torch.ops.emphases_amd.logmel(audio_packed, cu_samples)            mels.py:16-109
    torch.ops.emphases_amd.conv1d_same_act(x, w, b, cu_T, act)         convolution.py:25-37
    torch.ops.emphases_amd.segment_reduce(x, bounds, cu_frames, cu_words, mode)
                                                                       core.py:426-469
    torch.ops.emphases_amd.encoder_layer(x, in_w, in_b, out_w, out_b, norm1_w, norm1_b,
                                         ff1_w, ff1_b, ff2_w, ff2_b, norm2_w, norm2_b,
                                         cu_T, heads)                  transformer.py:18-30
    torch.ops.emphases_amd.prominence_forward(audio_packed, cu_samples, bounds, cu_words)
                                                                       core.py:295-342

Conventions: every segment keeps its OWN zero halo (the
reference's B = 1 semantics); inputs are borrowed, contiguous, on the op's HIP
device; outputs are freshly allocated on the current stream; `cu_*` are int32 /
int64 tensors of N + 1 prefix sums (on the host, or on the device at the price
of a synchronising copy - the table of a batch is built on the host).  The
ops run the library's kernels on the library's packed layout (segments start
16-aligned) and gather the result back into the caller's back-to-back layout;
callers that own the layout use `engine.Engine` directly and skip both copies.
No CPU implementation is registered: on a host tensor the dispatcher raises.
"""
import functools

import numpy as np
import torch

from . import batch
from . import config as cfg
from . import core
from . import engine as engine_module
from . import runtime
from . import weights as weights_module

__all__ = ['logmel', 'conv1d_same_act', 'segment_reduce', 'encoder_layer',
           'prominence_forward']


def _counts(cu):
    """Per-segment counts (python ints) of a prefix-sum tensor."""
    values = cu.detach().cpu().to(torch.int64).numpy()
    if values.ndim != 1 or values.size < 1 or values[0] != 0 or \
            np.any(np.diff(values) < 0):
        raise ValueError('cu_* must be non-decreasing prefix sums that start at 0')
    return np.diff(values).astype(np.int64), values


def _frame_plan(frames, words=None, bounds=None):
    """`batch.Plan` of already-featurised segments of `frames[i]` positions."""
    segments = []
    word_first = 0
    for index, count in enumerate(frames):
        if words is None:
            own = np.zeros((2, 0), dtype=np.int64)
        else:
            own = bounds[:, word_first:word_first + int(words[index])]
            word_first += int(words[index])
        segments.append(batch.Segment(
            index, 0, own.shape[1], 0, 0, int(count), own))
    return batch.Plan(segments, [0] * len(segments), [0] * len(segments))


def _columns(offsets, counts, device):
    """The library's packed columns of N back-to-back segments, as one index."""
    if not len(counts):
        return torch.zeros(0, dtype=torch.int64, device=device)
    index = np.concatenate([np.arange(off, off + count, dtype=np.int64)
                            for off, count in zip(offsets, counts)])
    return torch.from_numpy(index).to(device)


def _scatter(x, plan, offsets, counts, ld):
    """Back-to-back columns -> the library's packed axis (one indexed copy, not
    one per segment: 64 segments cost 0.5 ms of launches, tools/ops_cost.py)."""
    packed = torch.zeros((x.shape[0], ld), dtype=torch.float32, device=x.device)
    index = _columns(offsets, counts, x.device)
    packed.index_copy_(1, index, x[:, :index.numel()].to(torch.float32))
    return packed


def _gather(packed, offsets, counts):
    """The library's packed axis -> back-to-back columns (a fresh tensor)."""
    return packed.index_select(1, _columns(offsets, counts, packed.device))


def _device_index(tensor):
    if not tensor.is_cuda:
        raise runtime.LibraryError(
            'emphases_amd ops run on an MI355X / HIP device only (no CPU '
            'fallback): the tensor is on ' + str(tensor.device))
    return tensor.device.index


@torch.library.custom_op('emphases_amd::logmel', mutates_args=())
def logmel(audio_packed: torch.Tensor, cu_samples: torch.Tensor) -> torch.Tensor:
    """`mels.from_audio` (`mels.py:16-109`) of N chunks back to back: float32
    (or int16 PCM) `[sum S_i]` -> float32 `[80, sum F_i]`, F_i = 1 + (S_i -
    160) // 160 as in the reference (reflect padding of 432 per chunk)."""
    index = _device_index(audio_packed)
    samples, edges = _counts(cu_samples)
    engine = core.get_engine(None, index, cfg.DEFAULT)
    frames = [1 + (int(n) + 2 * cfg.PADDING - cfg.NUM_FFT) // cfg.HOPSIZE
              if n > cfg.PADDING else 0 for n in samples]
    if any(f <= 0 for f in frames):
        raise ValueError('a chunk needs more than 432 samples (reflect padding, mels.py:31-36)')
    # a chunk = an utterance whose one segment starts behind the zero padding
    segments = [batch.Segment(i, 0, 0, cfg.PADDING, int(n), f,
                              np.zeros((2, 0), dtype=np.int64))
                for i, (n, f) in enumerate(zip(samples, frames))]
    plan = batch.Plan(segments, edges[:-1], samples)
    with torch.cuda.device(index), engine.lock:
        meta = engine.upload(plan)
        packed = engine.features(audio_packed.contiguous(), plan, meta)
        return _gather(packed, plan.frame_off, plan.frames)


class _Borrowed:
    """A tensor as an lru_cache argument that neither hashes nor compares."""

    def __init__(self, tensor):
        self.tensor = tensor

    def __hash__(self):
        return 0

    def __eq__(self, other):
        return True


def _cached_conv(weight, bias, index):
    return _conv_layer_cached(
        weight.data_ptr(), weight._version, tuple(weight.shape),
        None if bias is None else (bias.data_ptr(), bias._version), index,
        _Borrowed(weight), _Borrowed(bias))


@functools.lru_cache(maxsize=32)
def _conv_layer_cached(pointer, version, shape, bias_key, index, weight, bias):
    bias = bias.tensor
    return engine_module._Conv(
        weight.tensor.detach().cpu().numpy(),
        None if bias is None else bias.detach().cpu().numpy(),
        torch.device('cuda', index), winograd=True)


@torch.library.custom_op('emphases_amd::conv1d_same_act', mutates_args=())
def conv1d_same_act(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
                    cu_T: torch.Tensor, activation: str) -> torch.Tensor:
    """`Conv1d(C_in, C_out, k, padding='same')` + activation
    (`convolution.py:25-37`) over N segments back to back, each with its own
    zero halo: x `[C_in, sum T_i]`, weight `[C_out, C_in, k]`, bias `[C_out]`
    -> `[C_out, sum T_i]`.  activation: 'none' | 'relu' | 'gelu' | 'silu' |
    'leaky_relu'.  The packed weights are cached per (weight storage,
    version)."""
    index = _device_index(x)
    counts, _ = _counts(cu_T)
    if int(counts.sum()) != x.shape[1]:
        raise ValueError('cu_T does not cover the columns of x')
    activation = None if activation in ('none', '') else activation
    if activation is not None and activation not in cfg.ACTIVATIONS:
        raise ValueError(f'Activation {activation} is not defined')
    engine = core.get_engine(None, index, cfg.DEFAULT)
    layer = _cached_conv(weight, bias, index)
    plan = _frame_plan(counts)
    with torch.cuda.device(index), engine.lock:
        tile = 64 if (layer.winograd4 is not None and engine.quad and
                      activation in (None, 'relu')) else \
            32 if layer.winograd is not None else 16
        host, offsets = plan.pack_metadata([(runtime.AXIS_FRAMES, tile)])
        device_meta = torch.from_numpy(host).to(x.device)
        meta = {name: (device_meta[start:start + size], size)
                for name, (start, size) in offsets.items()}
        meta['positions'] = (plan.total_frames, plan.total_words)
        packed = _scatter(x.to(torch.float32), plan, plan.frame_off, plan.frames,
                          plan.ld_frames)
        out = torch.zeros((layer.c_out, plan.ld_frames), dtype=torch.float32,
                          device=x.device)
        engine._conv(layer, packed, plan.ld_frames, out, plan.ld_frames, meta,
                     runtime.AXIS_FRAMES, tile, activation)
        return _gather(out, plan.frame_off, plan.frames)


@torch.library.custom_op('emphases_amd::segment_reduce', mutates_args=())
def segment_reduce(x: torch.Tensor, bounds: torch.Tensor, cu_frames: torch.Tensor,
                   cu_words: torch.Tensor, mode: str) -> torch.Tensor:
    """`emphases.downsample` (`core.py:426-469`): x `[C, sum F_i]`, bounds int
    `[2, sum W_i]` (frames relative to the word's own segment), mode 'sum' |
    'average' | 'max' | 'center' -> `[C, sum W_i]`.  An empty word: 0 (sum),
    NaN (average), as in the reference; 'max' of an empty word raises."""
    index = _device_index(x)
    if mode not in cfg.DOWNSAMPLE_METHODS:
        raise ValueError(f'Interpolation method {mode} is not defined')
    frames, _ = _counts(cu_frames)
    words, _ = _counts(cu_words)
    if len(frames) != len(words):
        raise ValueError('cu_frames and cu_words describe different numbers of segments')
    host_bounds = bounds.detach().cpu().to(torch.int64).numpy().reshape(2, -1)
    plan = _frame_plan(frames, words, host_bounds)
    engine_module.check_bounds(plan, mode)
    lib = runtime.library()
    with torch.cuda.device(index):
        host, offsets = plan.pack_metadata([])
        meta = torch.from_numpy(host).to(x.device)
        view = lambda name: meta[  # noqa: E731
            offsets[name][0]:offsets[name][0] + offsets[name][1]]
        packed = _scatter(x.to(torch.float32), plan, plan.frame_off, plan.frames,
                          plan.ld_frames)
        out = torch.zeros((x.shape[0], plan.ld_words), dtype=torch.float32,
                          device=x.device)
        runtime.check(lib.emph_segment_reduce(
            packed.data_ptr(), plan.ld_frames, view('bounds').data_ptr(),
            out.data_ptr(), plan.ld_words, x.shape[0], view('table').data_ptr(),
            view('word_segment').data_ptr(), plan.ld_words,
            runtime.REDUCTIONS[mode], runtime.stream()), 'emph_segment_reduce')
        return _gather(out, plan.word_off, plan.words)


_LAYER_NAMES = ('self_attn.in_proj_weight', 'self_attn.in_proj_bias',
                'self_attn.out_proj.weight', 'self_attn.out_proj.bias',
                'norm1.weight', 'norm1.bias', 'linear1.weight', 'linear1.bias',
                'linear2.weight', 'linear2.bias', 'norm2.weight', 'norm2.bias')


@functools.lru_cache(maxsize=8)
def _layer_engine(key, index, channels, heads, tensors):
    config = cfg.Config(architecture='transformer', layers=1, channels=channels,
                        heads=heads)
    state = weights_module.random_state(config, seed=0)
    for name, tensor in zip(_LAYER_NAMES, tensors.tensor):
        state['frame_encoder.model.layers.0.' + name] = np.ascontiguousarray(
            tensor.detach().cpu().numpy(), dtype=np.float32)
    return engine_module.Engine(config, state, index)


@torch.library.custom_op('emphases_amd::encoder_layer', mutates_args=())
def encoder_layer(x: torch.Tensor, in_proj_weight: torch.Tensor,
                  in_proj_bias: torch.Tensor, out_proj_weight: torch.Tensor,
                  out_proj_bias: torch.Tensor, norm1_weight: torch.Tensor,
                  norm1_bias: torch.Tensor, linear1_weight: torch.Tensor,
                  linear1_bias: torch.Tensor, linear2_weight: torch.Tensor,
                  linear2_bias: torch.Tensor, norm2_weight: torch.Tensor,
                  norm2_bias: torch.Tensor, cu_T: torch.Tensor,
                  heads: int) -> torch.Tensor:
    """One `nn.TransformerEncoderLayer` (post-LN, ReLU, eps 1e-5; dropout is the
    identity at inference) as `transformer.py:18-30` stacks them, over N
    segments back to back (attention within a segment only): x `[C, sum T_i]`
    -> `[C, sum T_i]`; no positional encoding is added."""
    index = _device_index(x)
    counts, _ = _counts(cu_T)
    tensors = (in_proj_weight, in_proj_bias, out_proj_weight, out_proj_bias,
               norm1_weight, norm1_bias, linear1_weight, linear1_bias,
               linear2_weight, linear2_bias, norm2_weight, norm2_bias)
    key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)
    engine = _layer_engine(key, index, int(x.shape[0]), int(heads),
                           _Borrowed(tensors))
    plan = _frame_plan(counts)
    with torch.cuda.device(index), engine.lock:
        meta = engine.upload(plan)
        packed = _scatter(x.to(torch.float32), plan, plan.frame_off, plan.frames,
                          plan.ld_frames)
        other = torch.zeros_like(packed)
        encoded = engine._stack_forward(
            engine.frame_encoder, packed, other, plan.ld_frames, plan, meta,
            runtime.AXIS_FRAMES, meta['tile'], 'op', positioned=True)
        return _gather(encoded, plan.frame_off, plan.frames)


@torch.library.custom_op('emphases_amd::prominence_forward', mutates_args=())
def prominence_forward(audio_packed: torch.Tensor, cu_samples: torch.Tensor,
                       bounds: torch.Tensor, cu_words: torch.Tensor) -> torch.Tensor:
    """`infer` + `postprocess` (`core.py:295-342`) with the active
    configuration and the bundled checkpoint over N CHUNKS back to back (what
    `preprocess` hands to `infer`, `core.py:345-418`: the audio slice at the
    frame-quantised word times, bounds relative to the chunk): audio `[sum
    S_i]`, bounds int `[2, sum W_i]` -> scores float32 `[sum W_i]`."""
    index = _device_index(audio_packed)
    samples, edges = _counts(cu_samples)
    words, _ = _counts(cu_words)
    if len(samples) != len(words):
        raise ValueError('cu_samples and cu_words describe different numbers of chunks')
    host_bounds = bounds.detach().cpu().to(torch.int64).numpy().reshape(2, -1)
    engine = core.get_engine(None, index)
    segments, first = [], 0
    for i, (n, w) in enumerate(zip(samples, words)):
        if n <= cfg.PADDING:
            raise ValueError('a chunk needs more than 432 samples (mels.py:31-36)')
        frames = 1 + (int(n) + 2 * cfg.PADDING - cfg.NUM_FFT) // cfg.HOPSIZE
        segments.append(batch.Segment(
            i, 0, int(w), cfg.PADDING, int(n), frames,
            host_bounds[:, first:first + int(w)]))
        first += int(w)
    plan = batch.Plan(segments, edges[:-1], samples)
    with torch.cuda.device(index), engine.lock:
        scores, _ = engine.forward(audio_packed.contiguous(), plan)
        columns = torch.from_numpy(plan.word_columns()).to(audio_packed.device)
        return scores[columns].clone()
