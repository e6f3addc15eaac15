"""ctypes binding of libemphases_hip.so (the C ABI of `include/emphases_hip.h`).

PyTorch is used here only for device memory and streams: tensors are handed to
the library as raw device pointers plus the current HIP stream.  There is NO
CPU fallback — if the library is missing, or no GPU is visible when a kernel
is requested, the call fails loudly.
"""
import ctypes
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(HERE, 'libemphases_hip.so')
ABI_VERSION = 30

# include/emphases_hip.h
SEG_FIELDS = 8
TILE_FIELDS = 4
(SEG_AUDIO_OFF, SEG_AUDIO_LEN, SEG_START, SEG_LENGTH, SEG_FRAME_OFF,
 SEG_FRAMES, SEG_WORD_OFF, SEG_WORDS) = range(8)
AXIS_FRAMES, AXIS_WORDS = 0, 1
# pseudo-axis of tile requests: the word axis cut the way emph_word_decoder wants
# it (emph_word_decoder_tiles); the 'block' of such a request is (layers,
# kernel_size, out_kernel_size)
AXIS_DECODER = 2
ACTIVATIONS = {None: 0, 'none': 0, 'relu': 1, 'gelu': 2, 'silu': 3,
               'leaky_relu': 4}
REDUCTIONS = {'sum': 0, 'average': 1, 'max': 2, 'center': 3}
POSTPROCESS = {None: 0, 'bce': 1, 'mse': 2}
AUDIO_F32, AUDIO_PCM16 = 0, 1
(METRIC_COUNT, METRIC_BCE, METRIC_SQUARED_ERROR, METRIC_COVARIANCE,
 METRIC_SUM_PREDICTED, METRIC_SUMSQ_PREDICTED, METRIC_SUM_TARGET,
 METRIC_SUMSQ_TARGET, METRIC_FIELDS) = range(9)

_c = ctypes
_ptr, _i32, _i64, _f32 = _c.c_void_p, _c.c_int32, _c.c_int64, _c.c_float

# name -> (restype, argtypes); every symbol include/emphases_hip.h declares
SIGNATURES = {
    'emph_abi_version': (_c.c_int, []),
    'emph_last_error': (_c.c_char_p, []),
    'emph_launch_probe': (_c.c_int, [_ptr]),
    'emph_launch_timer_begin': (_c.c_int, [_i32]),
    'emph_launch_timer_count': (_i32, []),
    'emph_launch_timer_end': (_c.c_int, [_ptr, _i32, _ptr]),
    'emph_frontend_table_size': (_i64, []),
    'emph_frontend_table_fill': (_c.c_int, [_ptr]),
    'emph_frontend_block': (_i32, []),
    'emph_logmel': (_c.c_int, [
        _ptr, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr,
        _i64, _i32, _i32, _ptr, _ptr, _i32, _ptr]),
    'emph_frontend_peak': (_c.c_int, [
        _ptr, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr]),
    'emph_host_gather': (_c.c_int, [_ptr, _ptr, _ptr, _i32, _ptr, _i32]),
    'emph_resample': (_c.c_int, [
        _ptr, _i32, _ptr, _i32, _i64, _ptr, _i32, _i32, _i32, _ptr, _ptr]),
    'emph_pitch_rows': (_c.c_int, [
        _ptr, _ptr, _ptr, _i64, _i32, _i32, _i32, _f32, _f32, _ptr]),
    'emph_conv_pack_size': (_i64, [_i32, _i32, _i32]),
    'emph_conv_pack': (_c.c_int, [_ptr, _i32, _i32, _i32, _ptr]),
    'emph_conv1d': (_c.c_int, [
        _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _i32, _i32, _i32, _ptr, _i32,
        _i32, _i32, _ptr]),
    'emph_conv_winograd_pack_size': (_i64, [_i32, _i32]),
    'emph_conv_winograd_lds_bytes': (_i64, [_i32, _i32]),
    'emph_conv_winograd_pack': (_c.c_int, [_ptr, _i32, _i32, _ptr]),
    'emph_conv1d_winograd': (_c.c_int, [
        _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _i32, _i32, _ptr, _i32, _i32,
        _ptr]),
    'emph_conv_winograd4_pack_size': (_i64, [_i32, _i32]),
    'emph_conv_winograd4_lds_bytes': (_i64, [_i32, _i32]),
    'emph_conv_winograd4_pack': (_c.c_int, [_ptr, _i32, _i32, _ptr]),
    'emph_conv1d_winograd4': (_c.c_int, [
        _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _i32, _i32, _ptr, _i32,
        _ptr]),
    'emph_conv1d_winograd4_position': (_c.c_int, [
        _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _i32, _i32, _ptr, _i32,
        _ptr, _i32, _ptr]),
    'emph_conv1d_winograd4_word_sums': (_c.c_int, [
        _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _i32, _i32, _ptr, _i32,
        _ptr, _ptr]),
    'emph_conv_stack_max_layers': (_i32, []),
    'emph_conv_stack_spans': (_i32, [_ptr, _ptr, _i32, _ptr]),
    'emph_conv1d_stack': (_c.c_int, [
        _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _i32, _ptr, _i32, _ptr,
        _ptr]),
    'emph_conv_split_pack_size': (_i64, []),
    'emph_conv_split_pack': (_c.c_int, [_ptr, _ptr]),
    'emph_conv1d_split': (_c.c_int, [
        _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _i32, _ptr, _i32, _ptr,
        _ptr]),
    'emph_word_sums': (_c.c_int, [
        _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i32, _i64, _i32, _ptr]),
    'emph_conv_winograd4_split_pack': (_c.c_int, [_ptr, _i32, _i32, _ptr]),
    'emph_conv1d_winograd4_half': (_c.c_int, [
        _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _i32, _i32, _ptr, _i32,
        _ptr, _i32, _i32, _ptr]),
    'emph_segment_reduce': (_c.c_int, [
        _ptr, _i64, _ptr, _ptr, _i64, _i32, _ptr, _ptr, _i64, _i32, _ptr]),
    'emph_output_layer': (_c.c_int, [
        _ptr, _i64, _ptr, _ptr, _i32, _i32, _ptr, _ptr, _i64, _i32, _i32,
        _ptr, _ptr, _ptr]),
    'emph_linear_chain_pack_size': (_i64, [_i32]),
    'emph_linear_chain_pack': (_c.c_int, [_ptr, _i32, _i32, _ptr]),
    'emph_transformer_block': (_c.c_int, [
        _ptr, _ptr, _i64, _i32, _ptr, _ptr, _c.c_float, _i32, _ptr, _i32,
        _i32, _ptr]),
    'emph_transformer_block_qkv': (_c.c_int, [
        _ptr, _ptr, _i64, _i32, _ptr, _ptr, _c.c_float, _i32, _ptr, _i32,
        _i32, _ptr, _ptr, _ptr]),
    'emph_qkv_projection': (_c.c_int, [
        _ptr, _i64, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    'emph_linear_split_pack_size': (_i64, [_i32]),
    'emph_linear_split_pack': (_c.c_int, [_ptr, _i32, _ptr]),
    'emph_transformer_block_split': (_c.c_int, [
        _ptr, _ptr, _i64, _i32, _ptr, _i32, _ptr, _c.c_float, _i32, _ptr, _i32,
        _i32, _ptr]),
    'emph_linear_split_pack16_size': (_i64, [_i32]),
    'emph_linear_split_pack16': (_c.c_int, [_ptr, _i32, _ptr]),
    'emph_position_wise_split': (_c.c_int, [
        _ptr, _ptr, _i64, _i32, _i32, _ptr, _ptr, _ptr, _ptr, _i32, _i32,
        _c.c_float, _i32, _ptr, _i32, _i32, _ptr, _ptr, _ptr, _ptr]),
    'emph_transformer_block_qkv_split': (_c.c_int, [
        _ptr, _ptr, _i64, _i32, _i32, _ptr, _ptr, _i32, _i32, _ptr, _ptr,
        _c.c_float, _i32, _ptr, _i32, _i32, _ptr, _ptr, _ptr, _ptr]),
    'emph_qkv_projection_split': (_c.c_int, [
        _ptr, _i64, _ptr, _ptr, _i32, _ptr, _i32, _ptr, _ptr, _i32, _i32,
        _ptr]),
    'emph_qkv_projection_split_images': (_c.c_int, [
        _ptr, _i64, _ptr, _ptr, _i32, _i32, _ptr, _i32, _i32, _ptr, _ptr,
        _i32, _i32, _ptr]),
    'emph_prominence_workspace_floats': (
        _i64, [_i32, _i32, _i64, _i64, _i32]),
    'emph_prominence_forward': (_c.c_int, [
        _ptr, _ptr, _i32, _ptr, _ptr, _i32, _ptr, _i32, _i32, _ptr, _i32, _ptr,
        _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr]),
    'emph_files_affinity': (_c.c_int, [_ptr, _i32]),
    'emph_files_open': (_c.c_int, [_ptr, _ptr, _i32, _i32, _ptr]),
    'emph_files_close': (None, [_ptr]),
    'emph_files_error': (_c.c_char_p, [_ptr, _i32]),
    'emph_files_sizes': (_c.c_int, [_ptr, _ptr]),
    'emph_files_alignments': (_c.c_int, [
        _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'emph_files_tier_name_bytes': (_i64, [_ptr]),
    'emph_files_read_audio': (_c.c_int, [
        _ptr, _ptr, _ptr, _ptr, _i32, _ptr, _i32]),
    'emph_files_write': (_c.c_int, [
        _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32]),
    'emph_plan_tiles': (_i64, [_ptr, _ptr, _i32, _i32, _i64, _i64, _ptr]),
    'emph_plan_batch': (_c.c_int, [
        _ptr, _ptr, _ptr, _i32, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr,
        _ptr, _ptr, _i64, _ptr, _ptr]),
    'emph_plan_word_sums': (_i64, [
        _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i64,
        _ptr, _ptr, _ptr, _ptr, _i64, _ptr]),
    'emph_gather_columns': (_c.c_int, [
        _ptr, _i64, _ptr, _i64, _i32, _ptr, _i32, _ptr]),
    'emph_word_decoder_block': (_i32, [_i32, _i32, _i32]),
    'emph_word_decoder_tiles': (_i32, [_ptr, _ptr, _i32, _i32, _i32, _i32, _ptr]),
    'emph_word_decoder_pack_size': (_i64, [_i32, _i32]),
    'emph_word_decoder_pack': (_c.c_int, [_ptr, _i32, _i32, _ptr]),
    'emph_word_decoder': (_c.c_int, [
        _ptr, _i64, _ptr, _i32, _i32, _ptr, _ptr, _i32, _i32, _i32, _ptr, _ptr,
        _i32, _i32, _ptr, _ptr, _ptr]),
    'emph_add_position': (_c.c_int, [
        _ptr, _i64, _ptr, _i32, _i32, _ptr, _i32, _i32, _ptr]),
    'emph_attention': (_c.c_int, [
        _ptr, _ptr, _ptr, _i64, _i32, _i32, _ptr, _i32, _i32, _ptr, _ptr]),
    'emph_split_kv_bytes': (_i64, [_i64, _i32, _i32, _i32, _i32]),
    'emph_split_kv': (_c.c_int, [
        _ptr, _ptr, _i64, _i32, _i32, _ptr, _i32, _i32, _i32, _ptr, _ptr]),
    'emph_attention_split': (_c.c_int, [
        _ptr, _ptr, _ptr, _i64, _i32, _i32, _ptr, _i32, _i32, _ptr, _i32,
        _ptr]),
    'emph_word_transformer_pack_size': (_i64, [_i32, _i32]),
    'emph_word_transformer_pack': (_c.c_int, [_ptr] * 12 + [_i32, _i32, _ptr]),
    'emph_word_transformer': (_c.c_int, [
        _ptr, _i64, _ptr, _i32, _i32, _i32, _ptr, _i32, _f32, _ptr, _i32, _ptr]),
    'emph_word_metrics': (_c.c_int, [
        _ptr, _ptr, _ptr, _i64, _i32, _f32, _f32, _ptr, _ptr]),
    'emph_add_layernorm': (_c.c_int, [
        _ptr, _ptr, _ptr, _i64, _i32, _ptr, _ptr, _f32, _i64, _i64, _ptr]),
}

_library = None


class LibraryError(RuntimeError):
    """The HIP library is missing, stale, or reported an error."""


def library():
    """Load (once) and return the shared library; raise if it is absent."""
    global _library
    if _library is None:
        if not os.path.exists(LIBRARY_PATH):
            raise LibraryError(
                f'{LIBRARY_PATH} not found: build it with '
                '`make -C emphases_amd/csrc` (or `__graft_entry__.build()`). '
                'There is no CPU fallback for the HIP path.')
        lib = ctypes.CDLL(LIBRARY_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                function = getattr(lib, name)
            except AttributeError as error:
                raise LibraryError(
                    f'{LIBRARY_PATH} does not export {name}') from error
            function.restype = restype
            function.argtypes = argtypes
        if lib.emph_abi_version() != ABI_VERSION:
            raise LibraryError(
                f'ABI version {lib.emph_abi_version()} != {ABI_VERSION}; '
                'rebuild the library')
        _library = lib
    return _library


# Host code of this package enters no torch CPU parallel region: what it does
# with tensors on the host is either small (below ATen's grain size) or goes
# through numpy / the library (`session.host_*`).  ANY parallel region wakes
# torch's whole OpenMP pool - as many threads as the machine shows cores - and
# the pool spins after the region; in a container whose cgroup allows fewer
# CPUs than it shows (128 cores, quota 16 on the test box) a few such regions
# exhaust the quota and the kernel FREEZES the process for the rest of the
# 100 ms period (`from_files_to_files`: 5 k files/s with four freezes in
# 0.4 s, 24 k without; tools/files_probe.py).  Round 4 capped torch's pool
# around every API call instead (`torch.set_num_threads(1)`, restored after):
# a process-global switch that throttled a caller's own CPU work in other
# threads.  It is gone: nothing in this package calls torch.set_num_threads.


def require_gpu(device=None):
    """Resolve the torch device of the HIP path or fail loudly."""
    if not torch.cuda.is_available():
        raise LibraryError(
            'no MI355X/HIP device is visible; the prominence hot path has no '
            'CPU fallback (the CPU restatement lives in oracle/ and is test '
            'infrastructure only)')
    if device is None:
        return torch.device('cuda', torch.cuda.current_device())
    if isinstance(device, int):
        return torch.device('cuda', device)
    return torch.device(device)


def check(status, name):
    if status != 0:
        message = library().emph_last_error().decode('utf-8', 'replace')
        raise LibraryError(f'{name} failed with status {status}: {message}')


def pointer(tensor):
    """Device (or host) address of a contiguous tensor / numpy array."""
    if tensor is None:
        return None
    if isinstance(tensor, np.ndarray):
        assert tensor.flags['C_CONTIGUOUS']
        return tensor.ctypes.data
    assert tensor.is_contiguous()
    return tensor.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


###############################################################################
# Host helpers (no GPU needed)
###############################################################################


class LaunchTimer:
    """Kernel-exact durations of the launches the library makes from this
    thread inside the `with` block (`emph_launch_timer_*`: events bound to each
    kernel's own dispatch packet - the kernel's begin -> end as rocprofv3 sees
    it, without the dispatch a pair of recorded events brackets too).
    `microseconds` (float32 array, launch order) and `launches` are valid after
    the block; `count()` inside it.  Measurement only."""

    def __init__(self, capacity=4096):
        self.capacity = int(capacity)
        self.microseconds = np.zeros(0, dtype=np.float32)
        self.launches = 0

    def __enter__(self):
        check(library().emph_launch_timer_begin(self.capacity),
              'emph_launch_timer_begin')
        return self

    def count(self):
        return int(library().emph_launch_timer_count())

    def __exit__(self, kind, value, trace):
        durations = np.zeros(self.capacity, dtype=np.float32)
        seen = _i32(0)
        status = library().emph_launch_timer_end(
            durations.ctypes.data, self.capacity, _c.byref(seen))
        if kind is None:
            check(status, 'emph_launch_timer_end')
        self.launches = int(seen.value)
        self.microseconds = durations[:min(self.launches, self.capacity)]
        return False


def frontend_table():
    """float32 table of Hann window + FFT twiddles (host, numpy)."""
    lib = library()
    table = np.zeros(lib.emph_frontend_table_size(), dtype=np.float32)
    check(lib.emph_frontend_table_fill(table.ctypes.data),
          'emph_frontend_table_fill')
    return table


def linear_chain_pack(weight, natural):
    """emph_transformer_block's layout of a square Linear weight [C, C]."""
    lib = library()
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    channels = weight.shape[0]
    assert weight.shape == (channels, channels)
    pack = np.zeros(lib.emph_linear_chain_pack_size(channels),
                    dtype=np.float32)
    check(lib.emph_linear_chain_pack(
        weight.ctypes.data, channels, int(natural), pack.ctypes.data),
        'emph_linear_chain_pack')
    return pack


def word_decoder_tiles(counts, offsets, layers, kernel_size, out_kernel_size):
    """Tile table int32 [n, 4] of emph_word_decoder for segments of `counts`
    words whose first columns are `offsets`."""
    lib = library()
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    arguments = (counts.ctypes.data, offsets.ctypes.data, len(counts),
                 int(layers), int(kernel_size), int(out_kernel_size))
    total = lib.emph_word_decoder_tiles(*arguments, None)
    if total < 0:
        raise ValueError('emph_word_decoder_tiles: bad decoder shape')
    tiles = np.zeros((total, TILE_FIELDS), dtype=np.int32)
    if total:
        lib.emph_word_decoder_tiles(*arguments, tiles.ctypes.data)
    return tiles


def word_transformer_pack(state, prefix, channels, heads):
    """emph_word_transformer's image of one encoder layer (`state`: numpy
    float32 arrays under `prefix`, the names of nn.TransformerEncoderLayer)."""
    lib = library()
    names = ('self_attn.in_proj_weight', 'self_attn.in_proj_bias',
             'self_attn.out_proj.weight', 'self_attn.out_proj.bias',
             'linear1.weight', 'linear1.bias', 'linear2.weight', 'linear2.bias',
             'norm1.weight', 'norm1.bias', 'norm2.weight', 'norm2.bias')
    arrays = [np.ascontiguousarray(state[prefix + name], dtype=np.float32)
              for name in names]
    pack = np.zeros(lib.emph_word_transformer_pack_size(channels, heads),
                    dtype=np.float32)
    check(lib.emph_word_transformer_pack(
        *[array.ctypes.data for array in arrays], channels, heads,
        pack.ctypes.data), 'emph_word_transformer_pack')
    return pack


class ConvModel(_c.Structure):
    """`emph_conv_model` of include/emphases_hip.h."""
    _fields_ = [(name, _i32) for name in (
        'channels', 'features', 'encoder_layers', 'decoder_layers',
        'decoder_kernel_size', 'activation', 'reduction', 'post',
        'normalize', 'mel_nnz', 'conv_variant')] + [(name, _ptr) for name in (
            'table', 'mel_start', 'mel_count', 'mel_offset', 'mel_values',
            'input_pack', 'input_bias', 'encoder_packs', 'encoder_biases',
            'decoder_packs', 'decoder_biases', 'out_weight', 'out_bias')]


class WordSumTables(_c.Structure):
    """`emph_word_sum_tables` of include/emphases_hip.h."""
    _fields_ = [(name, _ptr) for name in (
        'slot_map', 'terms', 'first', 'lengths')] + [('n_slots', _i32)]


def word_decoder_pack(weight):
    """emph_word_decoder's layout of one [channels, channels, k] weight."""
    lib = library()
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    channels, c_in, kernel_size = weight.shape
    assert channels == c_in
    pack = np.zeros(
        lib.emph_word_decoder_pack_size(channels, kernel_size),
        dtype=np.float32)
    check(lib.emph_word_decoder_pack(
        weight.ctypes.data, channels, kernel_size, pack.ctypes.data),
        'emph_word_decoder_pack')
    return pack


def conv_split_pack(weight):
    """bf16x3 pack (two bf16 pieces per weight, `emph_conv_split_pack`) of a
    [80, 80, 3] weight (host, numpy uint8)."""
    lib = library()
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    assert weight.shape == (80, 80, 3)
    pack = np.zeros(lib.emph_conv_split_pack_size(), dtype=np.uint8)
    check(lib.emph_conv_split_pack(weight.ctypes.data, pack.ctypes.data),
          'emph_conv_split_pack')
    return pack


def linear_split_pack(weight, pieces=2, tile=32):
    """Pack of `pieces` bf16 pieces per weight of a [80, 80] Linear weight
    (host, numpy uint8): `emph_linear_split_pack` for the kernels that own 32
    positions per wave, `emph_linear_split_pack16` for those that own 16."""
    lib = library()
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    assert weight.shape == (80, 80) and tile in (16, 32)
    size, fill = (lib.emph_linear_split_pack_size, lib.emph_linear_split_pack) \
        if tile == 32 else \
        (lib.emph_linear_split_pack16_size, lib.emph_linear_split_pack16)
    pack = np.zeros(size(pieces), dtype=np.uint8)
    check(fill(weight.ctypes.data, pieces, pack.ctypes.data),
          'emph_linear_split_pack')
    return pack


def conv_winograd4_pack(weight, split=False):
    """Winograd F(4,3) pack of a [c_out, c_in, 3] weight (host, numpy);
    `split`: cut into the two halves of `emph_conv1d_winograd4_half`."""
    lib = library()
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    c_out, c_in, kernel_size = weight.shape
    assert kernel_size == 3
    pack = np.zeros(
        lib.emph_conv_winograd4_pack_size(c_out, c_in), dtype=np.float32)
    function = lib.emph_conv_winograd4_split_pack if split else \
        lib.emph_conv_winograd4_pack
    check(function(weight.ctypes.data, c_out, c_in, pack.ctypes.data),
          'emph_conv_winograd4_pack')
    return pack


def conv_winograd4_lds_bytes(c_out, c_in):
    return int(library().emph_conv_winograd4_lds_bytes(c_out, c_in))


def conv_winograd_lds_bytes(c_out, c_in):
    return int(library().emph_conv_winograd_lds_bytes(c_out, c_in))


def conv_winograd_pack(weight):
    """Winograd F(2,3) pack of a [c_out, c_in, 3] weight (host, numpy)."""
    lib = library()
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    c_out, c_in, kernel_size = weight.shape
    assert kernel_size == 3
    pack = np.zeros(
        lib.emph_conv_winograd_pack_size(c_out, c_in), dtype=np.float32)
    check(lib.emph_conv_winograd_pack(
        weight.ctypes.data, c_out, c_in, pack.ctypes.data),
        'emph_conv_winograd_pack')
    return pack


def conv_pack(weight):
    """Reorder a [c_out, c_in, k] (or [c_out, c_in]) weight into the MFMA
    fragment order of emph_conv1d (host, numpy)."""
    lib = library()
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    if weight.ndim == 2:
        weight = weight[:, :, None]
    c_out, c_in, kernel_size = weight.shape
    pack = np.zeros(
        lib.emph_conv_pack_size(c_out, c_in, kernel_size), dtype=np.float32)
    check(lib.emph_conv_pack(
        weight.ctypes.data, c_out, c_in, kernel_size, pack.ctypes.data),
        'emph_conv_pack')
    return pack
