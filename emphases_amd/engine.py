"""Device-side execution of the prominence path on one MI355X.

`Engine` owns the immutable per-(checkpoint, device) state — MFMA-ordered
weight packs, biases, the front-end tables, the sparse mel basis — and runs a
`batch.Plan` through the HIP kernels of libemphases_hip.so:

    audio -> log-mel (+loudness) -> input conv -> frame encoder (conv stack or
    transformer) -> word-boundary reduce -> word decoder -> output conv ->
    sigmoid/clamp

which is `Model.forward` + `postprocess` of the reference
(`emphases/model/core.py:39-138`, `emphases/core.py:335-342`) for every chunk
of the batch at once, with per-chunk (B=1) edge semantics.  The reference
keeps its model on function attributes of `infer` (`core.py:298-315`); here the
cache is an explicit object.
"""
import contextlib
import functools
import os
import ctypes
import threading

import numpy as np
import torch

from . import config as cfg
from . import melbasis
from . import runtime
from . import weights as weights_module

FRONTEND_BLOCK = None   # frames per front-end tile: emph_frontend_block()
# `precision` of an engine -> (bf16 pieces per operand of the frame-rate convs,
# `pieces` code of emph_attention_split, pieces per operand of the Transformer's
# projections and position-wise GEMMs: csrc/block_split.hip); 0: the fp32 kernels.
#   'bf16x3'       convs: two pieces, three products per term.  Attention: THREE pieces
#                  for queries and keys (six products: the scores' error sits in front
#                  of an exponential and would grow with their range), two behind the
#                  softmax.  Projections and block: three pieces (24 GEMMs and 12
#                  LayerNorms in a row: with two, 2e-5 on the scores of configs[2];
#                  with three, 2e-6 - the f32 kernels' own noise)
#   'bf16x3_fast'  ... two pieces there too: every GEMM but the scores at three
#                  products per term
#   'bf16x6'       attention, projections and block: three pieces everywhere (fp32
#                  grade); the convs stay fp32 (six products of the direct form would
#                  not beat fp32 F(4,3))
PRECISIONS = {'f32': (0, 0, 0), 'bf16x3': (2, 32, 3), 'bf16x3_fast': (2, 32, 2),
              'bf16x6': (0, 3, 3)}
ATTENTION_BLOCK = 64    # queries per attention wave (csrc/transformer.hip)
ATTENTION_GROUP = 256   # queries per attention workgroup (LDS-staged keys/values)
# Which kernel takes a segment depends on the segment alone (scores must not
# depend on the rest of the batch): frame-axis segments of at least
# GROUPED_FROM positions take the workgroup attention kernel, shorter ones the
# one-wave kernel; word-axis segments of at most FUSED_WORDS words take the
# one-launch decoder (csrc/word_transformer.hip), longer ones the per-layer
# kernels.
GROUPED_FROM = ATTENTION_GROUP // 2
FUSED_WORDS = 64
EVERYTHING = 1 << 30
SHORT, LONG = (0, GROUPED_FROM - 1), (GROUPED_FROM, EVERYTHING)
FEW_WORDS, MANY_WORDS = (0, FUSED_WORDS), (FUSED_WORDS + 1, EVERYTHING)
WORD_TILE = 16
WINOGRAD_LDS_BUDGET = 160 * 1024


def a_weighting():
    """A-weighting minus REF_DB on the 8 kHz / 1024-bin grid the reference
    evaluates it on (`loudness.py:110-120`; librosa.A_weighting restated from
    its published formula — third-party, parity-unpinned)."""
    frequencies = np.fft.rfftfreq(n=1024, d=1.0 / 8000)
    f_sq = frequencies ** 2.0
    const = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2.0
    with np.errstate(divide='ignore'):
        weights = 2.0 + 20.0 * (
            np.log10(const[0]) + 2 * np.log10(f_sq)
            - np.log10(f_sq + const[0]) - np.log10(f_sq + const[1])
            - 0.5 * np.log10(f_sq + const[2])
            - 0.5 * np.log10(f_sq + const[3]))
    return (np.maximum(-80.0, weights) - cfg.REF_DB).astype(np.float32)


class _Conv:
    """Device copy of one Conv1d / Linear in MFMA fragment order."""

    def __init__(self, weight, bias, device, winograd=False):
        weight = np.asarray(weight, dtype=np.float32)
        if weight.ndim == 2:
            weight = weight[:, :, None]
        self.c_out, self.c_in, self.kernel_size = weight.shape
        self.weight = weight
        self.pack = torch.from_numpy(runtime.conv_pack(weight)).to(device)
        # Winograd F(2,3) form for the frame-rate k=3 layers whose transformed
        # weights fit in one CU's LDS next to the output patches
        self.winograd = None
        if winograd and self.kernel_size == 3 and \
                runtime.conv_winograd_lds_bytes(self.c_out, self.c_in) <= \
                WINOGRAD_LDS_BUDGET:
            self.winograd = torch.from_numpy(
                runtime.conv_winograd_pack(weight)).to(device)
        # Winograd F(4,3): half the MFMA work, identity / ReLU, 64-position tiles
        self.winograd4 = None
        if winograd and self.kernel_size == 3 and self.c_out <= 96 and \
                self.c_in % 4 == 0 and \
                runtime.conv_winograd4_lds_bytes(self.c_out, self.c_in) <= \
                WINOGRAD_LDS_BUDGET:
            self.winograd4 = torch.from_numpy(
                runtime.conv_winograd4_pack(weight)).to(device)
        self.bias = None if bias is None else torch.from_numpy(
            np.ascontiguousarray(bias, dtype=np.float32)).to(device)


class Engine:
    """Weights + constants on one device, and the kernel sequence."""

    def __init__(self, config=cfg.DEFAULT, state=None, device=None,
                 conv_tile=None, winograd=True, precision='f32'):
        """`precision`: 'f32' (default: every product on the fp32 matrix
        instruction) or an opt-in, 'bf16x3' / 'bf16x3_fast' / 'bf16x6': fp32
        operands split into two / three bf16 pieces, three / six products
        per term on the bf16 matrix pipe, fp32 accumulation (csrc/
        conv_split.hip, attention_split.hip, block_split.hip; the reference
        itself runs these matmuls under bf16 / fp16 autocast,
        core.py:594-607).  See `PRECISIONS` for what each name selects per
        kernel."""
        if precision not in PRECISIONS:
            raise ValueError(
                f'precision {precision!r} is not one of {sorted(PRECISIONS)}')
        self.config = config
        self.precision = precision
        self.split_pieces, self.attention_pieces, self.linear_pieces = \
            PRECISIONS[precision]
        self.winograd = winograd
        self.device = runtime.require_gpu(device)
        self.lib = runtime.library()
        global FRONTEND_BLOCK
        FRONTEND_BLOCK = int(self.lib.emph_frontend_block())
        self.conv_tile = conv_tile
        # when a list, every kernel launch is bracketed by HIP events on the
        # launch stream: (name, algorithmic flops, start, end)
        self.timers = None
        self._workspace = {}
        self._staging = {}
        # forward() returns workspace buffers: callers that may run on several
        # host threads hold this lock from pack_audio() until they have cloned
        # the scores (core.from_alignments_and_audios does)
        self.lock = threading.RLock()
        # positions per wave of the split position-wise kernels: 16
        # (csrc/block_split16.hip: two waves a SIMD, block + next projections in
        # one launch) or 32 (csrc/block_split.hip, one wave a SIMD)
        self.split_tile = int(os.environ.get('EMPHASES_SPLIT_TILE', 16))
        if self.split_tile not in (16, 32):
            raise ValueError('EMPHASES_SPLIT_TILE must be 16 or 32')
        state = weights_module.load(state, config)
        self.state = state
        dev = self.device
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731

        # Front-end constants
        self.table = to(runtime.frontend_table())
        basis = melbasis.default()
        self.mel_start = to(basis.row_start)
        self.mel_count = to(basis.row_count)
        self.mel_offset = to(basis.row_offset)
        self.mel_values = to(basis.values)
        self.mel_nnz = int(basis.values.size)
        self.a_weights = to(a_weighting())

        # Model
        self.input_layer = _Conv(
            state['input_layer.weight'], state['input_layer.bias'], dev,
            winograd)
        self.frame_encoder = self._stack('frame_encoder')
        self.word_decoder = self._stack('word_decoder') \
            if config.has_decoder else None
        self.output_weight = to(state['output_layer.weight'])
        self.output_bias = to(state['output_layer.bias'])
        # fused word stage (csrc/decoder.hip): conv decoder, or no decoder
        self.fused_words = config.architecture == 'convolution' or \
            not config.has_decoder
        self.word_block = WORD_TILE
        if self.fused_words:
            layers = self.word_decoder if config.has_decoder else []
            self.decoder_layers = len(layers)
            self.decoder_packs = torch.from_numpy(np.concatenate([
                runtime.word_decoder_pack(l.weight) for l in layers])).to(dev) \
                if layers else None
            self.decoder_biases = torch.cat([l.bias for l in layers]) \
                if layers else None
            self.word_block = int(self.lib.emph_word_decoder_block(
                self.decoder_layers, config.decoder_kernel_size,
                config.decoder_kernel_size))
            # the tile request of the fused word stage (emph_word_decoder_tiles)
            self.decoder_tiles = (runtime.AXIS_DECODER, (
                self.decoder_layers, config.decoder_kernel_size,
                config.decoder_kernel_size))
        if config.architecture == 'transformer':
            self.position = to(weights_module.positional_encoding(
                cfg.MAX_POSITIONS, config.channels))
            # channel-major copy: what the input layer's epilogue adds when the
            # encoding is fused into it (emph_conv1d_winograd4_position)
            self.position_rows = self.position.t().contiguous()
        # the whole word-rate Transformer decoder as one launch (segments of
        # at most 64 words): csrc/word_transformer.hip
        self.word_transformer = None
        if config.architecture == 'transformer' and config.has_decoder and \
                config.channels in (64, 80) and config.heads == 2 and all(
                    tuple(state[f'word_decoder.model.layers.{i}.{name}.weight']
                          .shape) == (config.channels, config.channels)
                    for i in range(config.layers)
                    for name in ('linear1', 'linear2')):
            self.word_transformer = to(np.concatenate([
                runtime.word_transformer_pack(
                    state, f'word_decoder.model.layers.{i}.', config.channels,
                    config.heads) for i in range(config.layers)]))
        # F(4,3) when every frame-rate k=3 layer has a pack and the activation
        # is one its register epilogue handles
        frame_layers = [self.input_layer] + (
            self.frame_encoder if config.architecture == 'convolution' else [])
        # EMPHASES_FUSE_QKV=0: the block and the next layer's projections as two launches
        self.fuse_qkv = os.environ.get('EMPHASES_FUSE_QKV', '1') != '0'
        self.quad = winograd and config.activation in (None, 'relu') and \
            all(layer.winograd4 is not None for layer in frame_layers)
        # the per-word sum folded into the last frame-rate layer's epilogue
        # (emph_conv1d_winograd4_word_sums + emph_word_sums): by configuration,
        # never by batch
        self.fold = bool(
            self.quad and config.architecture == 'convolution' and
            config.downsample_location != 'input' and
            config.downsample_method in ('sum', 'average') and
            len(self.frame_encoder) > 0 and config.channels % 4 == 0 and
            os.environ.get('EMPHASES_FOLD_WORD_SUMS', '1') != '0')
        # the frame-rate layers as a few launches of several layers each
        # (emph_conv1d_stack): the 80 -> 80, k = 3, identity / ReLU family
        frame_stack = [self.input_layer] + list(self.frame_encoder) \
            if config.architecture == 'convolution' else []
        self.stack = bool(
            self.quad and frame_stack and
            config.downsample_location != 'input' and
            all(layer.c_in == 80 and layer.c_out == 80 and
                layer.kernel_size == 3 and layer.winograd4 is not None
                for layer in frame_stack) and
            os.environ.get('EMPHASES_CONV_STACK', '1') != '0')
        if self.stack:
            # one allocation, the input layer first: the layout
            # emph_prominence_forward checks for
            self._stack_packs = torch.cat(
                [layer.winograd4 for layer in frame_stack])
            self._stack_biases = torch.cat(
                [layer.bias for layer in frame_stack])
        # precision='bf16x3': the same layers on the bf16 matrix pipe, direct
        # form, operands split into two bf16 pieces (csrc/conv_split.hip); up
        # to five layers per launch.  ('bf16x6' keeps the fp32 kernel here:
        # six products of the direct form would not beat F(4,3) in fp32)
        self.split_conv = bool(self.stack and self.split_pieces == 2)
        # computed positions between two restarts of the folded running sum
        self.sum_step = 32 if self.split_conv else 64
        if self.split_conv:
            self._split_packs = to(np.concatenate(
                [runtime.conv_split_pack(layer.weight) for layer in frame_stack]))
        self.model = self._conv_model()

    def lane(self):
        """A second handle on the same weights with its own workspace (and
        lock): what a batch in flight on another stream runs through."""
        import copy
        other = copy.copy(self)
        other._workspace = {}
        other._staging = {}
        other.timers = None
        other.lock = threading.RLock()
        return other

    def _conv_model(self):
        """`emph_conv_model` for emph_prominence_forward (the whole path in one
        C call), or None when this configuration needs the step-by-step path:
        Transformer, word pieces, extra feature rows, an encoder kernel other
        than 3, or direct-form convs requested."""
        config = self.config
        layers = [self.input_layer] + (
            self.frame_encoder if config.architecture == 'convolution' else [])
        if config.architecture != 'convolution' or not self.winograd or \
                config.downsample_location == 'input' or \
                config.num_features != cfg.NUM_MELS or \
                any(layer.winograd is None for layer in layers):
            return None
        encoder = self.frame_encoder
        pick = (lambda l: l.winograd4) if self.quad else (lambda l: l.winograd)
        input_pack, input_bias = pick(self.input_layer), self.input_layer.bias
        if self.stack:
            size = input_pack.numel()
            input_pack = self._stack_packs[:size]
            self._encoder_packs = self._stack_packs[size:]
            input_bias = self._stack_biases[:config.channels]
            self._encoder_biases = self._stack_biases[config.channels:]
        else:
            self._encoder_packs = torch.cat([pick(l) for l in encoder]) \
                if encoder else torch.zeros(1, device=self.device)
            self._encoder_biases = torch.cat([l.bias for l in encoder]) \
                if encoder else torch.zeros(1, device=self.device)
        pointer = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        return runtime.ConvModel(
            channels=config.channels, features=config.num_features,
            encoder_layers=len(encoder), decoder_layers=self.decoder_layers,
            decoder_kernel_size=config.decoder_kernel_size,
            activation=runtime.ACTIVATIONS[config.activation],
            reduction=runtime.REDUCTIONS[config.downsample_method],
            post=runtime.POSTPROCESS[config.loss],
            normalize=int(config.normalize), mel_nnz=self.mel_nnz,
            conv_variant=int(self.quad),
            table=pointer(self.table), mel_start=pointer(self.mel_start),
            mel_count=pointer(self.mel_count),
            mel_offset=pointer(self.mel_offset),
            mel_values=pointer(self.mel_values),
            input_pack=pointer(input_pack),
            input_bias=pointer(input_bias),
            encoder_packs=pointer(self._encoder_packs),
            encoder_biases=pointer(self._encoder_biases),
            decoder_packs=pointer(self.decoder_packs),
            decoder_biases=pointer(self.decoder_biases),
            out_weight=pointer(self.output_weight),
            out_bias=pointer(self.output_bias))

    def _stack(self, prefix):
        config, state, dev = self.config, self.state, self.device
        layers = []
        if config.architecture == 'convolution':
            for i in range(config.layers):
                layers.append(_Conv(
                    state[f'{prefix}.{2 * i}.weight'],
                    state[f'{prefix}.{2 * i}.bias'], dev,
                    self.winograd and prefix == 'frame_encoder'))
            return layers
        channels = config.channels
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        for i in range(config.layers):
            p = f'{prefix}.model.layers.{i}.'
            in_w = state[p + 'self_attn.in_proj_weight']
            in_b = state[p + 'self_attn.in_proj_bias']
            block = None
            square = (channels, channels)
            if channels in (64, 80) and all(
                    tuple(state[p + name].shape) == square for name in (
                        'self_attn.out_proj.weight', 'linear1.weight',
                        'linear2.weight')):
                # the position-wise half of the layer as one launch
                block = (
                    to(np.concatenate([
                        runtime.linear_chain_pack(
                            state[p + 'self_attn.out_proj.weight'], True),
                        runtime.linear_chain_pack(
                            state[p + 'linear1.weight'], False),
                        runtime.linear_chain_pack(
                            state[p + 'linear2.weight'], False)])),
                    to(np.concatenate([
                        state[p + name].astype(np.float32) for name in (
                            'self_attn.out_proj.bias', 'norm1.weight',
                            'norm1.bias', 'linear1.bias', 'linear2.bias',
                            'norm2.weight', 'norm2.bias')])))
            qkv = None
            if channels in (64, 80):
                qkv = (to(np.concatenate([
                    runtime.linear_chain_pack(
                        in_w[part * channels:(part + 1) * channels], True)
                    for part in range(3)])), to(in_b.astype(np.float32)))
            # precision='bf16x3': the same two launches on the bf16 matrix pipe
            block_split = qkv_split = None
            if self.linear_pieces and channels == 80 and block is not None:
                pieces = self.linear_pieces
                pack = functools.partial(
                    runtime.linear_split_pack, pieces=pieces,
                    tile=self.split_tile)
                block_split = (
                    to(np.concatenate([
                        pack(state[p + name])
                        for name in ('self_attn.out_proj.weight',
                                     'linear1.weight', 'linear2.weight')])),
                    block[1])
                qkv_split = (to(np.concatenate([
                    pack(in_w[part * channels:(part + 1) * channels])
                    for part in range(3)])), qkv[1])
            layers.append(dict(
                block=block, qkv=qkv, block_split=block_split,
                qkv_split=qkv_split,
                qk=_Conv(in_w[:2 * channels], in_b[:2 * channels], dev),
                v=_Conv(in_w[2 * channels:], in_b[2 * channels:], dev),
                out=_Conv(state[p + 'self_attn.out_proj.weight'],
                          state[p + 'self_attn.out_proj.bias'], dev),
                linear1=_Conv(state[p + 'linear1.weight'],
                              state[p + 'linear1.bias'], dev),
                linear2=_Conv(state[p + 'linear2.weight'],
                              state[p + 'linear2.bias'], dev),
                norm1=(to(state[p + 'norm1.weight']),
                       to(state[p + 'norm1.bias'])),
                norm2=(to(state[p + 'norm2.weight']),
                       to(state[p + 'norm2.bias']))))
        # layer i's block with layer i + 1's Q / K / V projections behind it in
        # one launch (six packs and ten vectors must fit in LDS: 80 channels do)
        for i in range(config.layers - 1):
            p = f'{prefix}.model.layers.{i + 1}.'
            block = layers[i]['block']
            layers[i]['block_qkv'] = None
            if block is None or layers[i + 1]['qkv'] is None or \
                    (6 * channels * channels + 10 * channels) * 4 > 160 * 1024:
                continue
            in_w = state[p + 'self_attn.in_proj_weight']
            in_b = state[p + 'self_attn.in_proj_bias']
            layers[i]['block_qkv'] = (
                torch.cat([block[0]] + [to(runtime.linear_chain_pack(
                    in_w[part * channels:(part + 1) * channels], False))
                    for part in range(3)]),
                torch.cat([block[1], to(in_b.astype(np.float32))]))
        if layers:
            layers[-1]['block_qkv'] = None
        return layers

    ###########################################################################
    # Plan upload
    ###########################################################################

    def frame_tile(self, plan):
        """Positions per conv wave on the frame axis.

        Default (`conv_tile=None`): the kernel FAMILY of a frame-rate layer is
        fixed by the configuration, never by the batch - F(4,3) (64-position
        tiles) where it is legal, else F(2,3), else the direct form - so an
        utterance's scores are bitwise the same alone, inside any batch and
        on any shard of any world size.  Within a family the tile size only
        partitions positions (same arithmetic per output), so it is picked by
        cost.  `conv_tile='auto'`: the round-3 policy, lowest latency - a
        handful of utterances takes the direct form's 16-position tiles,
        which agree with the Winograd kernels to 1e-6, not bitwise (one 10 s
        utterance: 4.6 us per layer less).

        Cost model: the chip runs 256 workgroups at a time (one per CU: the
        weights fill most of the LDS) of 4 waves (64-position tiles) or 8
        waves (16 / 32), so a launch costs (trips over the 256 CUs) x (time of
        one trip).  Microseconds per trip measured on the 80x80 k=3 layer
        (tools/micro/conv_bench.hip): Winograd 32: 24.0, Winograd 64: 26.7,
        direct 64: 33.5, direct 32: 31.3, direct 16: 16.1; Winograd F(4,3)
        (64-position tiles, identity / ReLU layers): 20.7."""
        if self.conv_tile is not None and self.conv_tile != 'auto':
            return self.conv_tile
        if self.conv_tile is None and self.quad:
            return 64
        frames = [segment.frames for segment in plan.segments]
        if self.winograd:
            cost = {64: 20.7 if self.quad else 26.7, 32: 23.2, 16: 16.1}
        else:
            cost = {64: 33.5, 32: 31.3, 16: 16.1}
        candidates = (64, 32, 16) if self.quad else (32, 64, 16)
        if self.conv_tile is None and self.winograd:
            candidates = (32, 64)       # stay inside the F(2,3) family
        best = None
        for tile in candidates:
            tiles = sum(-(-count // tile) for count in frames)
            waves = 4 if tile == 64 else 8
            if tile == 64 and self.quad:
                waves = 4          # four 64-position tiles per 8-wave workgroup
            groups = -(-tiles // waves)
            trips = -(-groups // 256)
            if best is None or trips * cost[tile] < best[0]:
                best = (trips * cost[tile], tile)
        return best[1]

    def _pinned(self, name, array):
        """`array` (int32 numpy) in a reusable pinned staging tensor: a fresh
        `pin_memory()` per batch costs up to milliseconds (hipHostMalloc).  The
        previous copy out of the buffer must have been consumed - the API
        paths synchronise on a batch before they reuse its engine - so the
        buffers rotate over a few slots to stay clear of copies in flight."""
        slots = self._staging.setdefault(name, [[], 0])
        ring, cursor = slots
        if len(ring) < 4:
            ring.append(None)
        index = cursor % len(ring)
        slots[1] = cursor + 1
        tensor = ring[index]
        if tensor is None or tensor.numel() < array.size:
            tensor = torch.empty(
                max(array.size, 1) * 3 // 2, dtype=torch.int32).pin_memory()
            ring[index] = tensor
        view = tensor[:array.size]
        np.copyto(view.numpy(), array)
        return view

    def prepare(self, plan):
        """The host half of `upload` ahead of time: tile tables, span table,
        word-sum tables and the packed metadata array, kept on the plan - what
        a prefetch thread does for batch i + 1 while batch i is submitted
        (`core.files_to_scores`).  Returns the plan."""
        if self.config.downsample_location != 'input':
            tile = self.frame_tile(plan)
            plan.prepared = (tile,) + self._pack(plan, tile, False)
        return plan

    def _pack(self, plan, tile, nested):
        requests = [
            (runtime.AXIS_FRAMES, FRONTEND_BLOCK),
            (runtime.AXIS_FRAMES, tile),
            self.decoder_tiles if self.fused_words
            else (runtime.AXIS_WORDS, self.word_block)]
        if self.config.architecture == 'transformer':
            # (the fused projection / block kernels own 32 positions per wave,
            # whatever tile the input layer's conv kernel takes)
            requests += [(runtime.AXIS_FRAMES, ATTENTION_BLOCK),
                         (runtime.AXIS_FRAMES, ATTENTION_BLOCK) + SHORT,
                         (runtime.AXIS_FRAMES, ATTENTION_GROUP) + LONG,
                         (runtime.AXIS_FRAMES, 32)]
            if self.linear_pieces:
                requests += [(runtime.AXIS_FRAMES, self.split_tile)]
            if not nested:
                requests += [(runtime.AXIS_WORDS, ATTENTION_BLOCK)]
                if self.word_transformer is not None:
                    requests += [
                        (runtime.AXIS_WORDS, ATTENTION_BLOCK) + FEW_WORDS,
                        (runtime.AXIS_WORDS, ATTENTION_BLOCK) + MANY_WORDS,
                        (runtime.AXIS_WORDS, self.word_block) + MANY_WORDS]
        requests = list(dict.fromkeys(requests))
        fold = self.fold and not nested and tile == 64
        stack = self.stack and not nested and tile == 64
        host, offsets = plan.pack_metadata(
            requests, word_sums=fold, spans=stack, sum_step=self.sum_step)
        return host, offsets, fold, stack

    def upload(self, plan, tile=None, nested=False):
        """One H2D copy of all integer metadata; returns device views.
        (`nested`: the layout of the word pieces of another plan.)"""
        tile = tile or self.frame_tile(plan)
        prepared = getattr(plan, 'prepared', None)
        if prepared is not None and prepared[0] == tile and not nested:
            host, offsets, fold, stack = prepared[1:]
        else:
            host, offsets, fold, stack = self._pack(plan, tile, nested)
        pinned = self._pinned(('meta', nested), host)
        device_buffer = pinned.to(self.device, non_blocking=True)
        views = {'_buffer': device_buffer, '_pinned': pinned, 'tile': tile,
                 'positions': (plan.total_frames, plan.total_words)}
        for name, (start, size) in offsets.items():
            views[name] = (device_buffer[start:start + size], size)
        if fold:
            views['n_slots'] = plan.word_sum_tables(
                plan.stack_restarts(self.sum_step)
                if stack else None)['n_slots']
            views['word_sum_tables'] = runtime.WordSumTables(
                *[views[('word_sums', name)][0].data_ptr() for name in (
                    'slot_map', 'terms', 'first', 'lengths')],
                views['n_slots'])
        if self.config.downsample_location == 'input' and not nested and \
                len(plan):
            # every word is its own padded sequence: a second packed layout
            pieces = plan.pieces(self.config.downsample_method)
            piece_meta = self.upload(
                pieces.plan, self.frame_tile(pieces.plan), nested=True)
            extra = self._pinned('pieces', np.concatenate([
                pieces.gather.view(np.int32).ravel(),
                pieces.bounds.ravel(), pieces.word_piece,
                pieces.gather[:, 1].astype(np.int32)]))
            extra = extra.to(self.device, non_blocking=True)
            cut = pieces.gather.size * 2
            piece_meta['gather'] = extra[:cut]
            piece_meta['piece_bounds'] = extra[cut:cut + pieces.bounds.size]
            cut += pieces.bounds.size
            piece_meta['word_piece'] = extra[cut:cut + pieces.word_piece.size]
            # frames of each piece that are the word itself (the rest of the
            # piece is zero padding): the Transformer's key-padding mask
            piece_meta['key_counts'] = extra[cut + pieces.word_piece.size:]
            piece_meta['plan'] = pieces.plan
            piece_meta['_extra'] = extra
            views['pieces'] = piece_meta
        return views

    ###########################################################################
    # Kernel wrappers
    ###########################################################################

    @contextlib.contextmanager
    def _timed(self, name, flops=0.):
        if self.timers is None:
            yield
            return
        begin = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        # (with a `runtime.LaunchTimer` armed around the pass: the indices of
        # this region's launches among the timer's kernel-exact durations)
        first = self.lib.emph_launch_timer_count()
        begin.record()
        yield
        end.record()
        self.timers.append((name, flops, begin, end, first,
                            self.lib.emph_launch_timer_count()))

    def _buffer(self, name, *shape):
        """Reusable float32 scratch tensor.  Contents are undefined: every
        kernel masks what lies outside a segment by selection, never by
        arithmetic, so stale or non-finite padding cannot leak into results."""
        key = (name, shape)
        tensor = self._workspace.get(key)
        if tensor is None:
            for stale in [k for k in self._workspace if k[0] == name]:
                del self._workspace[stale]
            tensor = torch.empty(shape, dtype=torch.float32, device=self.device)
            self._workspace[key] = tensor
        return tensor

    def pack_audio(self, audios):
        """All utterances (1-D float tensors) back to back in one device
        buffer: one copy per utterance straight into its slice (a host-side
        torch.cat of 64 x 10 s first costs 80 ms of page faults on 41 MB, and
        a pinned staging buffer is slower still to fill: 0.45 GB/s).  The
        returned tensor is a workspace buffer, valid until the next call."""
        lengths = [int(audio.shape[0]) for audio in audios]
        total = sum(lengths)
        packed = self._buffer('packed_audio', max(total, 1))[:total]
        offset = 0
        for audio, length in zip(audios, lengths):
            packed[offset:offset + length].copy_(audio, non_blocking=True)
            offset += length
        return packed

    def _conv(self, layer, x, ldx, y, ldy, meta, axis, block, activation,
              transpose_out=False, position=False):
        """`position`: the layer feeds a Transformer encoder - add the
        positional encoding in the same launch when the F(4,3) kernel takes
        the layer.  Returns whether it did."""
        tiles, size = meta[('tiles', axis, block)]
        positions = meta['positions'][axis]
        name = (f'conv1d_{"frames" if axis == runtime.AXIS_FRAMES else "words"}'
                f'_{layer.c_in}x{layer.c_out}_k{layer.kernel_size}')
        flops = 2. * layer.c_in * layer.c_out * layer.kernel_size * positions
        bias = None if layer.bias is None else layer.bias.data_ptr()
        if layer.winograd4 is not None and axis == runtime.AXIS_FRAMES and \
                block == 64 and self.quad and not transpose_out and \
                activation in (None, 'relu'):
            with self._timed(name.replace('conv1d', 'conv1d_winograd4'), flops):
                if position:
                    runtime.check(self.lib.emph_conv1d_winograd4_position(
                        x.data_ptr(), ldx, y.data_ptr(), ldy,
                        layer.winograd4.data_ptr(), bias, layer.c_in,
                        layer.c_out, runtime.ACTIVATIONS[activation],
                        tiles.data_ptr(), size // runtime.TILE_FIELDS,
                        self.position_rows.data_ptr(), cfg.MAX_POSITIONS,
                        runtime.stream()), 'emph_conv1d_winograd4_position')
                    return True
                runtime.check(self.lib.emph_conv1d_winograd4(
                    x.data_ptr(), ldx, y.data_ptr(), ldy,
                    layer.winograd4.data_ptr(), bias, layer.c_in, layer.c_out,
                    runtime.ACTIVATIONS[activation], tiles.data_ptr(),
                    size // runtime.TILE_FIELDS, runtime.stream()),
                    'emph_conv1d_winograd4')
            return False
        if layer.winograd is not None and axis == runtime.AXIS_FRAMES and \
                block in (32, 64) and not transpose_out:
            # same algorithmic flops; the kernel executes two thirds of them
            with self._timed(name.replace('conv1d', 'conv1d_winograd'), flops):
                runtime.check(self.lib.emph_conv1d_winograd(
                    x.data_ptr(), ldx, y.data_ptr(), ldy,
                    layer.winograd.data_ptr(), bias, layer.c_in, layer.c_out,
                    runtime.ACTIVATIONS[activation], tiles.data_ptr(),
                    size // runtime.TILE_FIELDS, block, runtime.stream()),
                    'emph_conv1d_winograd')
            return False
        with self._timed(name, flops):
            runtime.check(self.lib.emph_conv1d(
                x.data_ptr(), ldx, y.data_ptr(), ldy, layer.pack.data_ptr(),
                bias, layer.c_in, layer.c_out, layer.kernel_size,
                runtime.ACTIVATIONS[activation], tiles.data_ptr(),
                size // runtime.TILE_FIELDS, block, int(transpose_out),
                runtime.stream()), 'emph_conv1d')
        return False

    def features(self, audio, plan, meta, tracks=None, config=None):
        """Feature matrix [num_features, ld_frames] of every segment
        (`data/preprocess/core.py:71-125`).  The pitch tracker (`penn`, a
        third-party neural network) runs outside the library: `tracks` =
        float32 device tensor [2, ld_frames] with its per-frame pitch (Hz) and
        periodicity on the packed frame axis (`batch.pack_tracks`); the
        log2 / normalisation / row placement of core.py:94-106,123 happens
        on the device.  `config`: another configuration's feature switches
        (`data.preprocess.mels.from_audio` asks for the mel rows alone)."""
        config = config or self.config
        rows = config.num_features
        out = self._buffer('features', rows, plan.ld_frames)
        mel_row = 0 if config.mel_feature else -1
        loud_row = rows - 1 if config.loudness_feature else -1
        if config.pitch_feature or config.periodicity_feature:
            if tracks is None or tuple(tracks.shape) != (2, plan.ld_frames):
                raise NotImplementedError(
                    'pitch/periodicity features come from the third-party '
                    '`penn` tracker; pass its outputs as tracks '
                    '[2, ld_frames] (see core.from_alignments_and_audios '
                    'pitch_tracker=)')
            first = cfg.NUM_MELS if config.mel_feature else 0
            pitch_row = first if config.pitch_feature else -1
            periodicity_row = first + int(config.pitch_feature) \
                if config.periodicity_feature else -1
            with self._timed('pitch_rows'):
                runtime.check(self.lib.emph_pitch_rows(
                    tracks[0].data_ptr(), tracks[1].data_ptr(),
                    out.data_ptr(), plan.ld_frames, pitch_row,
                    periodicity_row, int(config.normalize),
                    float(np.log2(np.float32(cfg.FMIN))),
                    float(np.log2(np.float32(cfg.FMAX))), runtime.stream()),
                    'emph_pitch_rows')
        tiles, size = meta[('tiles', runtime.AXIS_FRAMES, FRONTEND_BLOCK)]
        count = size // runtime.TILE_FIELDS
        table = meta['table'][0]
        peak = None
        if config.loudness_feature:
            peak = torch.zeros(
                len(plan), dtype=torch.float32, device=self.device)
            with self._timed('frontend_peak'):
                runtime.check(self.lib.emph_frontend_peak(
                    audio.data_ptr(), audio_format(audio), table.data_ptr(),
                    tiles.data_ptr(),
                    count, self.table.data_ptr(), peak.data_ptr(),
                    runtime.stream()), 'emph_frontend_peak')
        if mel_row >= 0 or loud_row >= 0:
            with self._timed('frontend_logmel'):
                runtime.check(self.lib.emph_logmel(
                    audio.data_ptr(), audio_format(audio), table.data_ptr(),
                    tiles.data_ptr(),
                    count, self.table.data_ptr(), self.mel_start.data_ptr(),
                    self.mel_count.data_ptr(), self.mel_offset.data_ptr(),
                    self.mel_values.data_ptr(), self.mel_nnz, out.data_ptr(),
                    plan.ld_frames, mel_row, loud_row,
                    None if peak is None else peak.data_ptr(),
                    self.a_weights.data_ptr(), int(config.normalize),
                    runtime.stream()), 'emph_logmel')
        return out

    def _transformer(self, layers, x, ld, plan, meta, axis, block, tag,
                     key_counts=None, positioned=False, select=()):
        """`Transformer.forward` (transformer.py:25-30) in place on x
        (`positioned`: the producer of x has added the encoding already).
        `key_counts`: int32 device tensor, real (unpadded) positions per
        segment, for the key-padding mask over zero-padded word pieces.
        `select`: (least, most) positions - only the segments of that size
        (their tiles), the others' columns are left alone."""
        config = self.config
        channels = config.channels
        if block > 32 and ('tiles', axis, 32) + select in meta:
            block = 32
        att_tiles, att_size = meta[('tiles', axis, ATTENTION_BLOCK) + select]
        att_count = att_size // runtime.TILE_FIELDS
        if att_count == 0:
            return x
        counts = plan.frames if axis == runtime.AXIS_FRAMES else plan.words
        if len(counts) and int(counts.max()) > cfg.MAX_POSITIONS:
            # transformer.py:40,51-52: the encoding table has 5000 rows
            raise RuntimeError(
                f'a chunk of {int(counts.max())} positions exceeds the '
                f'{cfg.MAX_POSITIONS}-entry positional encoding; pass a '
                'smaller batch_size')
        if not positioned:
            with self._timed('add_position'):
                runtime.check(self.lib.emph_add_position(
                    x.data_ptr(), ld, self.position.data_ptr(), channels,
                    cfg.MAX_POSITIONS, att_tiles.data_ptr(), att_count,
                    ATTENTION_BLOCK, runtime.stream()), 'emph_add_position')
        qk = self._buffer(tag + '_qk', 2 * channels, ld)
        v = self._buffer(tag + '_v', ld, channels)
        attended = self._buffer(tag + '_attended', channels, ld)
        projected = self._buffer(tag + '_projected', channels, ld)
        attention_flops = 4. * channels * float(
            (counts.astype(np.float64) ** 2).sum())
        # long segments: a workgroup of eight waves shares each key / value
        # block through LDS; short ones (the word axis, word pieces): one wave
        # per 64 queries straight from L2.  By SEGMENT, not by batch: two tile
        # tables, at most two launches per layer.
        launches = [(att_tiles, att_count, ATTENTION_BLOCK)]
        if axis == runtime.AXIS_FRAMES and not select and \
                ('tiles', axis, ATTENTION_GROUP) + LONG in meta:
            launches = []
            for key, size in (((ATTENTION_BLOCK,) + SHORT, ATTENTION_BLOCK),
                              ((ATTENTION_GROUP,) + LONG, ATTENTION_GROUP)):
                tiles, length = meta[('tiles', axis) + key]
                if length:
                    launches.append(
                        (tiles, length // runtime.TILE_FIELDS, size))

        def add_layernorm(norm):
            with self._timed('add_layernorm'):
                runtime.check(self.lib.emph_add_layernorm(
                    x.data_ptr(), projected.data_ptr(), x.data_ptr(), ld,
                    channels, norm[0].data_ptr(), norm[1].data_ptr(),
                    config.layer_norm_eps, 0, ld, runtime.stream()),
                    'emph_add_layernorm')

        # precision='bf16x3' / 'bf16x6': long segments take the bf16 matrix
        # pipe - the layer's keys and values are split once (all 64-position
        # tiles of the axis), the grouped launch reads the pieces
        split_images = None
        if self.attention_pieces and config.channels // config.heads == 40 and \
                any(tile_n == ATTENTION_GROUP for _, _, tile_n in launches):
            size = int(self.lib.emph_split_kv_bytes(
                ld, len(plan), channels, config.heads,
                self.attention_pieces))
            key = (tag + '_split_kv', (size,))
            split_images = self._workspace.get(key)
            if split_images is None:
                for stale in [k for k in self._workspace if k[0] == key[0]]:
                    del self._workspace[stale]
                split_images = torch.empty(
                    size, dtype=torch.uint8, device=self.device)
                self._workspace[key] = split_images

        # ... or the projection kernel writes the pieces itself (no fp32 K and V,
        # no second pass): when every segment of the axis is a long one
        projected_images = split_images is not None and \
            all(tile_n == ATTENTION_GROUP for _, _, tile_n in launches)

        def attend(images_written):
            for tiles, count, tile_n in launches:
                counts_pointer = None if key_counts is None else \
                    key_counts.data_ptr()
                if split_images is not None and tile_n == ATTENTION_GROUP:
                    if not images_written:
                        runtime.check(self.lib.emph_split_kv(
                            qk.data_ptr(), v.data_ptr(), ld, channels,
                            config.heads, att_tiles.data_ptr(), att_count,
                            ATTENTION_BLOCK, self.attention_pieces,
                            split_images.data_ptr(), runtime.stream()),
                            'emph_split_kv')
                    runtime.check(self.lib.emph_attention_split(
                        qk.data_ptr(), split_images.data_ptr(),
                        attended.data_ptr(), ld, channels, config.heads,
                        tiles.data_ptr(), count, tile_n, counts_pointer,
                        self.attention_pieces, runtime.stream()),
                        'emph_attention_split')
                    continue
                runtime.check(self.lib.emph_attention(
                    qk.data_ptr(), v.data_ptr(), attended.data_ptr(), ld,
                    channels, config.heads, tiles.data_ptr(), count, tile_n,
                    counts_pointer, runtime.stream()), 'emph_attention')

        split_tiles, wide = None, self.split_tile
        # (the 16-position kernels address rows with 32-bit byte offsets: a packed
        # axis of 2^22 columns or more - 11 hours of frames in ONE batch - takes the
        # fp32 position-wise kernels)
        if axis == runtime.AXIS_FRAMES and ld < (1 << 22 if wide == 16 else 1 << 29) \
                and ('tiles', axis, wide) + select in meta:
            split_tiles = meta[('tiles', axis, wide) + select]

        def position_wise(layer, following):
            """emph_position_wise_split: this layer's block (layer given) and / or
            the projections of the next (following given), tiles of 16."""
            tiles, size = split_tiles
            block_packs, vectors = layer['block_split'] if layer else (None, None)
            next_packs, next_bias = following['qkv_split'] if following \
                else (None, None)
            pointer = lambda t: None if t is None else t.data_ptr()  # noqa: E731
            runtime.check(self.lib.emph_position_wise_split(
                attended.data_ptr() if layer else None, x.data_ptr(), ld,
                channels, config.heads, pointer(block_packs), pointer(vectors),
                pointer(next_packs), pointer(next_bias), self.linear_pieces,
                self.attention_pieces, config.layer_norm_eps,
                runtime.ACTIVATIONS['relu'], tiles.data_ptr(),
                size // runtime.TILE_FIELDS, 16, qk.data_ptr(), v.data_ptr(),
                split_images.data_ptr() if projected_images and following
                else None, runtime.stream()), 'emph_position_wise_split')

        projected_ahead = False     # this layer's Q, K, V came out of the last block
        images_ahead = False        # ... its K and V as the attention's split images
        for index, layer in enumerate(layers):
            split = layer['block_split'] is not None and \
                split_tiles is not None and block <= 32
            images_written = False
            if projected_ahead:
                images_written = images_ahead
            elif split:
                packs, bias = layer['qkv_split']
                tiles, size = split_tiles
                with self._timed(f'qkv_projection_split_{tag}', 6. * channels *
                                 channels * meta['positions'][axis]):
                    if wide == 16:
                        position_wise(None, layer)
                        images_written = projected_images
                    elif projected_images:
                        runtime.check(self.lib.emph_qkv_projection_split_images(
                            x.data_ptr(), ld, qk.data_ptr(),
                            split_images.data_ptr(), channels, config.heads,
                            packs.data_ptr(), self.linear_pieces,
                            self.attention_pieces, bias.data_ptr(),
                            tiles.data_ptr(), size // runtime.TILE_FIELDS, 32,
                            runtime.stream()),
                            'emph_qkv_projection_split_images')
                        images_written = True
                    else:
                        runtime.check(self.lib.emph_qkv_projection_split(
                            x.data_ptr(), ld, qk.data_ptr(), v.data_ptr(),
                            channels, packs.data_ptr(), self.linear_pieces,
                            bias.data_ptr(), tiles.data_ptr(),
                            size // runtime.TILE_FIELDS, 32, runtime.stream()),
                            'emph_qkv_projection_split')
            elif layer['qkv'] is not None and block <= 32:
                packs, bias = layer['qkv']
                tiles, size = meta[('tiles', axis, block) + select]
                with self._timed(f'qkv_projection_{tag}', 6. * channels * channels *
                                 meta['positions'][axis]):
                    runtime.check(self.lib.emph_qkv_projection(
                        x.data_ptr(), ld, qk.data_ptr(), v.data_ptr(),
                        channels, packs.data_ptr(), bias.data_ptr(),
                        tiles.data_ptr(), size // runtime.TILE_FIELDS, block,
                        runtime.stream()), 'emph_qkv_projection')
            else:
                self._conv(layer['qk'], x, ld, qk, ld, meta, axis, block,
                           None)
                self._conv(layer['v'], x, ld, v, channels, meta, axis, block,
                           None, transpose_out=True)
            projected_ahead = False
            with self._timed(f'attention_{tag}', attention_flops):
                attend(images_written)
            following = layers[index + 1] if index + 1 < len(layers) else None
            if split and following is not None and self.fuse_qkv and \
                    following['block_split'] is not None:
                # ... and the NEXT layer's projections in the same launch (attention
                # has consumed qk / v / the images: the next layer's go there)
                packs, vectors = layer['block_split']
                next_packs, next_bias = following['qkv_split']
                tiles, size = split_tiles
                with self._timed(f'transformer_block_qkv_split_{tag}', 12. *
                                 channels * channels * meta['positions'][axis]):
                    if wide == 16:
                        position_wise(layer, following)
                    else:
                        runtime.check(self.lib.emph_transformer_block_qkv_split(
                            attended.data_ptr(), x.data_ptr(), ld, channels,
                            config.heads, packs.data_ptr(), next_packs.data_ptr(),
                            self.linear_pieces, self.attention_pieces,
                            vectors.data_ptr(), next_bias.data_ptr(),
                            config.layer_norm_eps, runtime.ACTIVATIONS['relu'],
                            tiles.data_ptr(), size // runtime.TILE_FIELDS, 32,
                            qk.data_ptr(), v.data_ptr(),
                            split_images.data_ptr() if projected_images else None,
                            runtime.stream()), 'emph_transformer_block_qkv_split')
                projected_ahead, images_ahead = True, projected_images
                continue
            if split:
                packs, vectors = layer['block_split']
                tiles, size = split_tiles
                with self._timed(f'transformer_block_split_{tag}', 6. * channels *
                                 channels * meta['positions'][axis]):
                    if wide == 16:
                        position_wise(layer, None)
                    else:
                        runtime.check(self.lib.emph_transformer_block_split(
                            attended.data_ptr(), x.data_ptr(), ld, channels,
                            packs.data_ptr(), self.linear_pieces, vectors.data_ptr(),
                            config.layer_norm_eps, runtime.ACTIVATIONS['relu'],
                            tiles.data_ptr(), size // runtime.TILE_FIELDS, 32,
                            runtime.stream()), 'emph_transformer_block_split')
                continue
            if layer['block_qkv'] is not None and block <= 32 and self.fuse_qkv:
                # (attention has consumed qk / v: the next layer's go there)
                packs, vectors = layer['block_qkv']
                tiles, size = meta[('tiles', axis, block) + select]
                with self._timed(f'transformer_block_qkv_{tag}', 12. * channels *
                                 channels * meta['positions'][axis]):
                    runtime.check(self.lib.emph_transformer_block_qkv(
                        attended.data_ptr(), x.data_ptr(), ld, channels,
                        packs.data_ptr(), vectors.data_ptr(),
                        config.layer_norm_eps, runtime.ACTIVATIONS['relu'],
                        tiles.data_ptr(), size // runtime.TILE_FIELDS, block,
                        qk.data_ptr(), v.data_ptr(), runtime.stream()),
                        'emph_transformer_block_qkv')
                projected_ahead = True
                continue
            if layer['block'] is not None and block <= 32:
                packs, vectors = layer['block']
                tiles, size = meta[('tiles', axis, block) + select]
                with self._timed(f'transformer_block_{tag}', 6. * channels *
                                 channels * meta['positions'][axis]):
                    runtime.check(self.lib.emph_transformer_block(
                        attended.data_ptr(), x.data_ptr(), ld, channels,
                        packs.data_ptr(), vectors.data_ptr(),
                        config.layer_norm_eps, runtime.ACTIVATIONS['relu'],
                        tiles.data_ptr(), size // runtime.TILE_FIELDS, block,
                        runtime.stream()), 'emph_transformer_block')
                continue
            self._conv(layer['out'], attended, ld, projected, ld, meta, axis,
                       block, None)
            add_layernorm(layer['norm1'])
            self._conv(layer['linear1'], x, ld, attended, ld, meta, axis,
                       block, 'relu')
            self._conv(layer['linear2'], attended, ld, projected, ld, meta,
                       axis, block, None)
            add_layernorm(layer['norm2'])
        return x

    def _frame_stack(self, features, ld_f, a, b, out, ld_w, plan, meta,
                     frames=True, words=True):
        """Input layer + frame encoder as groups of up to three layers per
        launch (`emph_conv1d_stack`: a workgroup owns a span of positions
        through the layers of a group, activations resident in LDS), then
        `emphases.downsample` into `out` - from running sums the last group
        leaves behind when the per-word sum is folded, else by
        `emph_segment_reduce`.  Same launches as emph_prominence_forward."""
        config = self.config
        channels = config.channels
        spans, span_size = meta['conv_spans']
        total = 1 + len(self.frame_encoder)
        most = int(self.lib.emph_conv_stack_max_layers())
        groups = -(-total // most)
        fold = 'word_sum_tables' in meta
        view = lambda name: meta[('word_sums', name)][0]  # noqa: E731
        sums = self._buffer(
            'word_sums', max(meta.get('n_slots', 0), 1), channels)
        pack = self.input_layer.winograd4.numel()
        relu_layers = config.activation == 'relu'
        source, buffers, done = features, (a, b), 0
        if self.split_conv:
            # bf16x3: up to five layers per launch
            groups = -(-total // 5)
            split_bytes = int(self.lib.emph_conv_split_pack_size())
        for group in range(groups if frames else 0):
            size = -(-(total - done) // (groups - group))
            relu = sum(1 << l for l in range(size)
                       if done + l >= 1 and relu_layers)
            to_sums = fold and group == groups - 1
            target = sums if to_sums else buffers[group & 1]
            flops = 2. * 80 * 80 * 3 * plan.total_frames * size
            if self.split_conv:
                with self._timed('conv1d_split_frames_80x80_k3', flops):
                    runtime.check(self.lib.emph_conv1d_split(
                        source.data_ptr(), ld_f, target.data_ptr(),
                        channels if to_sums else ld_f,
                        self._split_packs[done * split_bytes:].data_ptr(),
                        self._stack_biases[done * channels:].data_ptr(), size,
                        relu, spans.data_ptr(), span_size // 8,
                        view('slot_map').data_ptr() if to_sums else None,
                        runtime.stream()), 'emph_conv1d_split')
                source = target
                done += size
                continue
            with self._timed('conv1d_stack_frames_80x80_k3', flops):
                runtime.check(self.lib.emph_conv1d_stack(
                    source.data_ptr(), ld_f, target.data_ptr(),
                    channels if to_sums else ld_f,
                    self._stack_packs[done * pack:].data_ptr(),
                    self._stack_biases[done * channels:].data_ptr(), size,
                    relu, spans.data_ptr(), span_size // 8,
                    view('slot_map').data_ptr() if to_sums else None,
                    runtime.stream()), 'emph_conv1d_stack')
            source = target
            done += size
        if not words:
            return
        if fold:
            with self._timed('word_sums'):
                runtime.check(self.lib.emph_word_sums(
                    sums.data_ptr(), channels, view('terms').data_ptr(),
                    view('first').data_ptr(), view('lengths').data_ptr(),
                    out.data_ptr(), ld_w, channels, ld_w,
                    runtime.REDUCTIONS[config.downsample_method],
                    runtime.stream()), 'emph_word_sums')
        else:
            with self._timed('segment_reduce'):
                runtime.check(self.lib.emph_segment_reduce(
                    source.data_ptr(), ld_f, meta['bounds'][0].data_ptr(),
                    out.data_ptr(), ld_w, channels,
                    meta['table'][0].data_ptr(),
                    meta['word_segment'][0].data_ptr(), ld_w,
                    runtime.REDUCTIONS[config.downsample_method],
                    runtime.stream()), 'emph_segment_reduce')

    def _word_sums(self, x, ld_f, out, ld_w, plan, meta):
        """The last frame-rate layer + `emphases.downsample` ('sum' /
        'average', `core.py:438-454`) without the layer's output ever being
        written: running sums at the frames the words need, then a few signed
        terms per word."""
        config = self.config
        layer = self.frame_encoder[-1]
        channels = config.channels
        tiles, size = meta[('tiles', runtime.AXIS_FRAMES, 64)]
        sums = self._buffer('word_sums', max(meta['n_slots'], 1), channels)
        view = lambda name: meta[('word_sums', name)][0]  # noqa: E731
        flops = 2. * layer.c_in * layer.c_out * 3 * plan.total_frames
        with self._timed(
                f'conv1d_winograd4_frames_{layer.c_in}x{layer.c_out}_k3', flops):
            runtime.check(self.lib.emph_conv1d_winograd4_word_sums(
                x.data_ptr(), ld_f, sums.data_ptr(), channels,
                layer.winograd4.data_ptr(), layer.bias.data_ptr(), layer.c_in,
                layer.c_out, runtime.ACTIVATIONS[config.activation],
                tiles.data_ptr(), size // runtime.TILE_FIELDS,
                view('slot_map').data_ptr(), runtime.stream()),
                'emph_conv1d_winograd4_word_sums')
        with self._timed('word_sums'):
            runtime.check(self.lib.emph_word_sums(
                sums.data_ptr(), channels, view('terms').data_ptr(),
                view('first').data_ptr(), view('lengths').data_ptr(),
                out.data_ptr(), ld_w, channels, ld_w,
                runtime.REDUCTIONS[config.downsample_method],
                runtime.stream()), 'emph_word_sums')

    def _stack_forward(self, layers, x, other, ld, plan, meta, axis, block,
                       tag, key_counts=None, positioned=False):
        """Frame encoder / word decoder; returns the tensor holding the
        result (x or other)."""
        config = self.config
        if config.architecture == 'convolution':
            for layer in layers:
                self._conv(layer, x, ld, other, ld, meta, axis, block,
                           config.activation)
                x, other = other, x
            return x
        if axis == runtime.AXIS_WORDS and self.word_transformer is not None \
                and key_counts is None and len(plan.words):
            # segments of up to 64 words: the whole decoder in one launch;
            # longer ones: the per-layer kernels on THEIR tiles (disjoint
            # columns of x; which path a segment takes depends on it alone)
            tiles, size = meta[('tiles', axis, ATTENTION_BLOCK) + FEW_WORDS]
            if size:
                with self._timed('word_transformer'):
                    runtime.check(self.lib.emph_word_transformer(
                        x.data_ptr(), ld, self.position.data_ptr(),
                        cfg.MAX_POSITIONS, config.channels, config.heads,
                        self.word_transformer.data_ptr(), len(layers),
                        config.layer_norm_eps, tiles.data_ptr(),
                        size // runtime.TILE_FIELDS, runtime.stream()),
                        'emph_word_transformer')
            if int(plan.words.max()) > FUSED_WORDS:
                self._transformer(
                    layers, x, ld, plan, meta, axis, block, tag, key_counts,
                    positioned, select=MANY_WORDS)
            return x
        return self._transformer(
            layers, x, ld, plan, meta, axis, block, tag, key_counts, positioned)

    ###########################################################################
    # Forward
    ###########################################################################

    def forward(self, audio, plan, meta=None, stages=None, features=None,
                tracks=None):
        """Scores of every word of every segment.

        audio: float32 (or int16 PCM) device tensor, all utterances back to
            back.
        Returns (scores, logits): float32 [ld_words] on the packed word axis
        (`plan.word_columns()` picks the valid entries; other entries are
        undefined).  The tensors are workspace buffers: they are overwritten
        by the next forward() of this engine."""
        config = self.config
        meta = meta or self.upload(plan)
        block = meta['tile']
        channels = config.channels
        ld_f, ld_w = plan.ld_frames, plan.ld_words
        frames, words = runtime.AXIS_FRAMES, runtime.AXIS_WORDS
        if self.model is not None and stages is None and features is None \
                and self.timers is None and not self.split_conv and \
                block in ((64,) if self.quad else (32, 64)) and \
                len(plan):
            # the whole path behind one C call
            check_bounds(plan, config.downsample_method)
            logits = self._buffer('logits', ld_w)
            scores = self._buffer('scores', ld_w)
            tables = meta.get('word_sum_tables')
            spans, span_size = meta.get('conv_spans', (None, 0))
            floats = self.lib.emph_prominence_workspace_floats(
                config.num_features, channels, ld_f, ld_w,
                meta.get('n_slots', 0))
            workspace = self._buffer('fused', int(floats))
            frontend_tiles, frontend_size = meta[
                ('tiles', frames, FRONTEND_BLOCK)]
            frame_tiles, frame_size = meta[('tiles', frames, block)]
            word_tiles, word_size = meta[('tiles',) + self.decoder_tiles]
            runtime.check(self.lib.emph_prominence_forward(
                ctypes.byref(self.model), audio.data_ptr(),
                audio_format(audio), meta['table'][0].data_ptr(),
                frontend_tiles.data_ptr(),
                frontend_size // runtime.TILE_FIELDS, frame_tiles.data_ptr(),
                frame_size // runtime.TILE_FIELDS, block,
                word_tiles.data_ptr(), word_size // runtime.TILE_FIELDS,
                meta['bounds'][0].data_ptr(),
                meta['word_segment'][0].data_ptr(), ld_f, ld_w,
                workspace.data_ptr(), logits.data_ptr(), scores.data_ptr(),
                None if tables is None else ctypes.byref(tables),
                spans.data_ptr() if spans is not None else None,
                span_size // 8 if spans is not None else 0,
                runtime.stream()), 'emph_prominence_forward')
            return scores, logits
        if features is None:
            features = self.features(audio, plan, meta, tracks=tracks)
        table = meta['table'][0]
        logits = self._buffer('logits', ld_w)
        scores = self._buffer('scores', ld_w)
        wa = self._buffer('words_a', channels, ld_w)
        if config.downsample_location == 'input':
            # model/core.py:41-87: gather every word into its own zero-padded
            # piece, encode the pieces as independent sequences, pool each
            # over its padded length into the word's column
            if stages is not None:
                stages['features'] = features.clone()
            piece_meta = meta['pieces']
            piece_plan = piece_meta['plan']
            ld_p = piece_plan.ld_frames
            gathered = self._buffer(
                'piece_features', config.num_features, ld_p)
            with self._timed('gather_columns'):
                runtime.check(self.lib.emph_gather_columns(
                    features.data_ptr(), ld_f, gathered.data_ptr(), ld_p,
                    config.num_features, piece_meta['gather'].data_ptr(),
                    len(piece_plan), runtime.stream()),
                    'emph_gather_columns')
            a = self._buffer('frames_a', channels, ld_p)
            b = self._buffer('frames_b', channels, ld_p)
            self._conv(self.input_layer, gathered, ld_p, a, ld_p, piece_meta,
                       frames, piece_meta['tile'], None)
            encoded = self._stack_forward(
                self.frame_encoder, a, b, ld_p, piece_plan, piece_meta,
                frames, piece_meta['tile'], 'pieces',
                key_counts=piece_meta['key_counts']
                if config.architecture == 'transformer' else None)
            with self._timed('segment_reduce'):
                runtime.check(self.lib.emph_segment_reduce(
                    encoded.data_ptr(), ld_p,
                    piece_meta['piece_bounds'].data_ptr(), wa.data_ptr(),
                    ld_w, channels, piece_meta['table'][0].data_ptr(),
                    piece_meta['word_piece'].data_ptr(), ld_w,
                    runtime.REDUCTIONS[config.downsample_method],
                    runtime.stream()), 'emph_segment_reduce')
        else:
            a = self._buffer('frames_a', channels, ld_f)
            b = self._buffer('frames_b', channels, ld_f)
            if 'conv_spans' in meta and stages is None:
                # the frame-rate layers a few at a time (emph_conv1d_stack),
                # the last launch straight into the per-word sums when folded
                check_bounds(plan, config.downsample_method)
                self._frame_stack(features, ld_f, a, b, wa, ld_w, plan, meta)
            else:
                # (the stage dump wants the input layer's own output)
                positioned = self._conv(
                    self.input_layer, features, ld_f, a, ld_f, meta, frames, block,
                    None, position=config.architecture == 'transformer' and
                    stages is None)
                if stages is not None:
                    stages['features'] = features.clone()
                    stages['input_layer'] = a.clone()
                # (the stage dump wants the encoder's own output: the taps keep
                # the unfolded reduce, which agrees to summation order)
                fold = 'word_sum_tables' in meta and stages is None
                encoded = self._stack_forward(
                    self.frame_encoder[:-1] if fold else self.frame_encoder, a, b,
                    ld_f, plan, meta, frames, block, 'frames',
                    positioned=positioned)
                if stages is not None:
                    stages['encoder'] = encoded.clone()

                check_bounds(plan, config.downsample_method)
                if fold:
                    self._word_sums(encoded, ld_f, wa, ld_w, plan, meta)
                else:
                    with self._timed('segment_reduce'):
                        runtime.check(self.lib.emph_segment_reduce(
                            encoded.data_ptr(), ld_f,
                            meta['bounds'][0].data_ptr(), wa.data_ptr(), ld_w,
                            channels, table.data_ptr(),
                            meta['word_segment'][0].data_ptr(), ld_w,
                            runtime.REDUCTIONS[config.downsample_method],
                            runtime.stream()), 'emph_segment_reduce')
        if stages is not None:
            stages['downsampled'] = wa.clone()
        return self._word_stage(wa, plan, meta, logits, scores)

    def _word_stage(self, wa, plan, meta, logits, scores):
        """Word embeddings -> word decoder -> output layer -> postprocess."""
        config = self.config
        channels = config.channels
        ld_w = plan.ld_words
        words = runtime.AXIS_WORDS
        table = meta['table'][0]
        if self.fused_words:
            tiles, size = meta[('tiles',) + self.decoder_tiles]
            with self._timed('word_decoder', 2. * channels * (
                    channels * config.decoder_kernel_size * self.decoder_layers
                    + config.decoder_kernel_size) * plan.total_words):
                runtime.check(self.lib.emph_word_decoder(
                    wa.data_ptr(), ld_w, tiles.data_ptr(),
                    size // runtime.TILE_FIELDS, channels,
                    None if self.decoder_packs is None
                    else self.decoder_packs.data_ptr(),
                    None if self.decoder_biases is None
                    else self.decoder_biases.data_ptr(),
                    self.decoder_layers, config.decoder_kernel_size,
                    runtime.ACTIVATIONS[config.activation],
                    self.output_weight.data_ptr(), self.output_bias.data_ptr(),
                    config.decoder_kernel_size,
                    runtime.POSTPROCESS[config.loss], logits.data_ptr(),
                    scores.data_ptr(), runtime.stream()), 'emph_word_decoder')
            return scores, logits
        wb = self._buffer('words_b', channels, ld_w)
        decoded = self._stack_forward(
            self.word_decoder, wa, wb, ld_w, plan, meta, words,
            self.word_block, 'words')
        with self._timed('output_layer'):
            runtime.check(self.lib.emph_output_layer(
                decoded.data_ptr(), ld_w, self.output_weight.data_ptr(),
                self.output_bias.data_ptr(), channels,
                config.decoder_kernel_size, table.data_ptr(),
                meta['word_segment'][0].data_ptr(), ld_w, words,
                runtime.POSTPROCESS[config.loss], logits.data_ptr(),
                scores.data_ptr(), runtime.stream()), 'emph_output_layer')
        return scores, logits

    def splittable(self, plan, meta):
        """Whether forward() is `forward_frames()` + `forward_words()` for this
        configuration and layout: the default convolutional path (frame-rate
        layers by emph_conv1d_stack, per-word sums folded, one-launch decoder)."""
        return (self.config.downsample_location != 'input' and
                'conv_spans' in meta and 'word_sum_tables' in meta and
                self.fused_words and len(plan) > 0)

    def forward_frames(self, audio, plan, meta):
        """The frame-rate half of forward(): features and the frame-rate layers,
        up to the running sums the words are made of (left in the workspace)."""
        ld_f = plan.ld_frames
        channels = self.config.channels
        check_bounds(plan, self.config.downsample_method)
        features = self.features(audio, plan, meta)
        a = self._buffer('frames_a', channels, ld_f)
        b = self._buffer('frames_b', channels, ld_f)
        self._frame_stack(features, ld_f, a, b, None, plan.ld_words, plan, meta,
                          words=False)

    def forward_words(self, plan, meta):
        """The word-rate half: per-word sums -> decoder -> scores (same kernels,
        same bits as forward())."""
        ld_w = plan.ld_words
        logits = self._buffer('logits', ld_w)
        scores = self._buffer('scores', ld_w)
        wa = self._buffer('words_a', self.config.channels, ld_w)
        self._frame_stack(None, plan.ld_frames, None, None, wa, ld_w, plan, meta,
                          frames=False)
        return self._word_stage(wa, plan, meta, logits, scores)

    def capture(self, audio, plan, meta=None):
        """Capture forward() for this (audio buffer, plan) into a HIP graph.

        Returns `(replay, scores, logits)`: `replay()` re-enqueues the whole
        kernel sequence with one host call (about 10 us instead of one
        launch per kernel) and refreshes `scores` / `logits` in place.  The
        audio tensor may be refilled between replays; its address and the
        plan must not change."""
        meta = meta or self.upload(plan)
        self.timers = None
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # warm up: attribute calls, buffers
            for _ in range(2):
                self.forward(audio, plan, meta)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        # thread-local capture: the RCCL watchdog thread of a multi-GPU run may
        # query events while this thread captures
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            scores, logits = self.forward(audio, plan, meta)
        # the graph holds raw pointers into the workspace (allocated by the
        # warm-up, outside the graph's pool): keep those tensors alive for as
        # long as the replay closure lives, even if a later forward() with
        # another layout drops them from the workspace
        held = (list(self._workspace.values()), meta, audio)

        def replay():
            graph.replay()
            return held[0] is not None
        return replay, scores, logits


def audio_format(audio):
    """EMPH_AUDIO_* of a packed audio tensor: float32, or 16-bit PCM (the
    front-end applies the x / 32768 of `load.wav` itself)."""
    if audio.dtype == torch.float32:
        return runtime.AUDIO_F32
    if audio.dtype == torch.int16:
        return runtime.AUDIO_PCM16
    raise TypeError(f'packed audio must be float32 or int16, not {audio.dtype}')


def check_bounds(plan, method):
    """Host-side replicas of the two cases where the reference raises
    (`core.py:449-452,459-466`)."""
    if method not in ('max', 'center'):
        return
    for segment in plan.segments:
        starts, ends = segment.bounds
        if method == 'max' and np.any(
                np.minimum(ends, segment.frames) <=
                np.minimum(starts, segment.frames)):
            raise IndexError(
                'max(): Expected reduction dim 1 to have non-zero size '
                '(empty word)')
        if method == 'center' and np.any(
                (starts + ends) // 2 >= segment.frames):
            raise IndexError('word center lies beyond the last frame')
