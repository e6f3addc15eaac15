"""GPU: every C-ABI operator on seeded random ragged inputs against a plain
torch CPU fp32 reference of the same op (one segment at a time, B=1)."""
import math

import numpy as np
import pytest
import torch

from emphases_amd import batch, runtime, synth

pytestmark = pytest.mark.gpu

DEVICE = 'cuda:0'


def ragged_plan(frames, words_per_segment=None, seed=0):
    segments = []
    for index, count in enumerate(frames):
        if words_per_segment is None:
            bounds = synth.word_frames(seed + index, count, 1, 25) \
                if count else np.zeros((2, 0), dtype=np.int64)
        else:
            bounds = words_per_segment[index]
        segments.append(batch.Segment(
            index, 0, bounds.shape[1], 0, 0, count, bounds))
    return batch.Plan(segments, [0] * len(frames), [0] * len(frames))


class Meta:
    def __init__(self, plan, requests):
        host, offsets = plan.pack_metadata(requests)
        self.buffer = torch.from_numpy(host).to(DEVICE)
        self.offsets = offsets

    def view(self, name):
        start, size = self.offsets[name]
        return self.buffer[start:start + size], size


def random_packed(rows, plan, axis, seed):
    ld = plan.ld_frames if axis == runtime.AXIS_FRAMES else plan.ld_words
    values = synth.weights(seed, (rows, ld), 1.0)
    return torch.from_numpy(values)


def spans(plan, axis):
    if axis == runtime.AXIS_FRAMES:
        return list(zip(plan.frame_off, plan.frames))
    return list(zip(plan.word_off, plan.words))


###############################################################################
# word pieces
###############################################################################


def test_gather_columns():
    lib = runtime.library()
    x = torch.from_numpy(synth.weights(3, (81, 300), 1.0))
    table = np.array([[16, 5, 32, 21], [40, 21, 64, 21], [299, 1, 16, 3],
                      [100, 0, 96, 7]], dtype=np.int64)
    y = torch.full((81, 128), 9.0, device=DEVICE)
    x_dev, table_dev = x.to(DEVICE), torch.from_numpy(table).to(DEVICE)
    runtime.check(lib.emph_gather_columns(
        x_dev.data_ptr(), 300, y.data_ptr(), 128, 81, table_dev.data_ptr(),
        len(table), None), 'emph_gather_columns')
    y = y.cpu()
    want = torch.full((81, 128), 9.0)
    for source, length, target, padded in table:
        want[:, target:target + padded] = 0.
        want[:, target:target + length] = x[:, source:source + length]
    assert torch.equal(y, want)


###############################################################################
# conv1d
###############################################################################


ACTIVATIONS = {
    None: lambda x: x, 'relu': torch.relu,
    'gelu': torch.nn.functional.gelu, 'silu': torch.nn.functional.silu,
    'leaky_relu': lambda x: torch.nn.functional.leaky_relu(x, 0.01)}


@pytest.mark.parametrize('c_in,c_out,kernel_size,activation,tile', [
    (80, 80, 3, 'relu', 64), (80, 80, 3, 'relu', 32), (80, 80, 3, None, 16),
    (80, 80, 5, 'gelu', 32), (80, 80, 7, 'silu', 64), (80, 80, 1, None, 32),
    (64, 64, 3, 'leaky_relu', 32), (128, 128, 3, 'relu', 32),
    (81, 80, 3, None, 32), (80, 160, 1, None, 64), (80, 240, 1, 'relu', 16),
    (80, 1, 3, None, 16), (3, 7, 5, 'relu', 16)])
def test_conv1d(c_in, c_out, kernel_size, activation, tile):
    lib = runtime.library()
    plan = ragged_plan([200, 1, 17, 64, 65, 3, 130])
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile)])
    x = random_packed(c_in, plan, axis, 11)
    x[:, :batch.LEAD] = float('nan')           # padding must never be read as data
    weight = synth.weights(5, (c_out, c_in, kernel_size), 0.2)
    bias = synth.weights(6, (c_out,), 0.5)
    y = torch.full((c_out, plan.ld_frames), 7.0, device=DEVICE)
    pack = torch.from_numpy(runtime.conv_pack(weight)).to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    x_dev, bias_dev = x.to(DEVICE), torch.from_numpy(bias).to(DEVICE)
    runtime.check(lib.emph_conv1d(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        pack.data_ptr(), bias_dev.data_ptr(), c_in,
        c_out, kernel_size, runtime.ACTIVATIONS[activation],
        tiles.data_ptr(), size // 4, tile, 0, None), 'emph_conv1d')
    y = y.cpu()
    for off, count in spans(plan, axis):
        want = ACTIVATIONS[activation](torch.nn.functional.conv1d(
            x[None, :, off:off + count], torch.from_numpy(weight),
            torch.from_numpy(bias), padding=(kernel_size - 1) // 2))[0]
        got = y[:, off:off + count]
        assert torch.isfinite(got).all()
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) < 2e-5 * scale
    # columns outside every segment are left untouched
    assert float(y[:, :batch.LEAD].min()) == 7.0


@pytest.mark.parametrize('c_in,c_out,activation,tile', [
    (80, 80, 'relu', 64), (80, 80, None, 32), (81, 80, 'relu', 64),
    (83, 80, None, 32), (80, 96, 'gelu', 64), (128, 128, 'relu', 64),
    (64, 64, 'leaky_relu', 32), (3, 7, None, 64), (80, 1, None, 32)])
def test_conv1d_winograd(c_in, c_out, activation, tile):
    """F(2,3) form == Conv1d(k=3, 'same') on ragged segments, odd lengths,
    channel counts that do not fill the last 4-row group / 16-row tile."""
    lib = runtime.library()
    plan = ragged_plan([200, 1, 2, 17, 64, 65, 3, 130, 31, 33])
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile)])
    x = random_packed(c_in, plan, axis, 11)
    x[:, :batch.LEAD] = float('nan')
    x[:, -batch.TAIL:] = float('nan')
    weight = synth.weights(5, (c_out, c_in, 3), 0.2)
    bias = synth.weights(6, (c_out,), 0.5)
    y = torch.full((c_out, plan.ld_frames), 7.0, device=DEVICE)
    assert runtime.conv_winograd_lds_bytes(c_out, c_in) <= 160 * 1024
    pack = torch.from_numpy(runtime.conv_winograd_pack(weight)).to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    x_dev, bias_dev = x.to(DEVICE), torch.from_numpy(bias).to(DEVICE)
    runtime.check(lib.emph_conv1d_winograd(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        pack.data_ptr(), bias_dev.data_ptr(), c_in, c_out,
        runtime.ACTIVATIONS[activation], tiles.data_ptr(), size // 4, tile,
        None), 'emph_conv1d_winograd')
    y = y.cpu()
    for off, count in spans(plan, axis):
        want = ACTIVATIONS[activation](torch.nn.functional.conv1d(
            x[None, :, off:off + count], torch.from_numpy(weight),
            torch.from_numpy(bias), padding=1))[0]
        got = y[:, off:off + count]
        assert torch.isfinite(got).all()
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) < 2e-5 * scale
    assert float(y[:, :batch.LEAD].min()) == 7.0


@pytest.mark.parametrize('c_in,c_out,activation', [
    (80, 80, 'relu'), (80, 80, None), (64, 64, 'relu'), (64, 96, None),
    (84, 80, 'relu'), (4, 7, None), (80, 1, None), (48, 33, 'relu')])
def test_conv1d_winograd4(c_in, c_out, activation):
    """F(4,3) form == Conv1d(k=3, 'same') on ragged segments, odd lengths,
    channel counts that do not fill the last group / tile / half."""
    lib = runtime.library()
    plan = ragged_plan([200, 1, 2, 3, 4, 5, 17, 64, 65, 130, 63, 66, 127])
    axis, tile = runtime.AXIS_FRAMES, 64
    meta = Meta(plan, [(axis, tile)])
    x = random_packed(c_in, plan, axis, 11)
    x[:, :batch.LEAD] = float('nan')
    x[:, -batch.TAIL:] = float('nan')
    weight = synth.weights(5, (c_out, c_in, 3), 0.2)
    bias = synth.weights(6, (c_out,), 0.5)
    y = torch.full((c_out, plan.ld_frames), 7.0, device=DEVICE)
    assert runtime.conv_winograd4_lds_bytes(c_out, c_in) <= 160 * 1024
    pack = torch.from_numpy(runtime.conv_winograd4_pack(weight)).to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    x_dev, bias_dev = x.to(DEVICE), torch.from_numpy(bias).to(DEVICE)
    runtime.check(lib.emph_conv1d_winograd4(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        pack.data_ptr(), bias_dev.data_ptr(), c_in, c_out,
        runtime.ACTIVATIONS[activation], tiles.data_ptr(), size // 4, None),
        'emph_conv1d_winograd4')
    if c_out > 16:
        # the layer as two independent half launches: the same bits
        split = torch.from_numpy(runtime.conv_winograd4_pack(
            weight, split=True)).to(DEVICE)
        halves = torch.full((c_out, plan.ld_frames), 7.0, device=DEVICE)
        for half in (1, 0):
            runtime.check(lib.emph_conv1d_winograd4_half(
                x_dev.data_ptr(), plan.ld_frames, halves.data_ptr(),
                plan.ld_frames, split.data_ptr(), bias_dev.data_ptr(), c_in,
                c_out, runtime.ACTIVATIONS[activation], tiles.data_ptr(),
                size // 4, None, 0, half, None), 'emph_conv1d_winograd4_half')
        assert torch.equal(halves, y)
    y = y.cpu()
    for off, count in spans(plan, axis):
        want = ACTIVATIONS[activation](torch.nn.functional.conv1d(
            x[None, :, off:off + count], torch.from_numpy(weight),
            torch.from_numpy(bias), padding=1))[0]
        got = y[:, off:off + count]
        assert torch.isfinite(got).all()
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) < 5e-5 * scale
    assert float(y[:, :batch.LEAD].min()) == 7.0
    # transcendental activations and packs beyond the LDS are refused
    assert lib.emph_conv1d_winograd4(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        pack.data_ptr(), None, c_in, c_out, 2, tiles.data_ptr(), size // 4,
        None) == -2
    assert runtime.conv_winograd4_lds_bytes(80, 88) > 160 * 1024
    assert lib.emph_conv1d_winograd4(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        pack.data_ptr(), None, 83, 80, 0, tiles.data_ptr(), size // 4,
        None) == -2                         # c_in must be a multiple of 4


@pytest.mark.parametrize('layers,relu_mask', [
    (1, 0b1), (2, 0b10), (3, 0b110), (3, 0b111), (3, 0), (3, -0b110), (2, -0b01)])
def test_conv1d_stack_equals_layer_by_layer(layers, relu_mask):
    """emph_conv1d_stack (`layers` Conv1d(80, 80, 3) + identity / ReLU in ONE
    launch, activations resident in LDS, one recomputed quad of halo per side)
    == `layers` emph_conv1d_winograd4 launches, BIT FOR BIT, on ragged
    segments around every span boundary the span table can produce: one span
    (1 .. 256 positions), two spans (257, 504), three (505, 600), thirteen
    (3000); NaN in all padding.  Also against torch conv1d (2e-5)."""
    lib = runtime.library()
    frames = [1000, 1, 2, 3, 4, 5, 37, 64, 65, 252, 253, 256, 257, 504, 505,
              600, 3000, 130]
    if relu_mask < 0:
        # (a negative mask: the same layers on forty random lengths)
        relu_mask = -relu_mask
        rng = np.random.default_rng(1000 + layers)
        frames = [int(n) for n in rng.integers(1, 3001, size=40)]
    plan = ragged_plan(frames)
    axis, tile = runtime.AXIS_FRAMES, 64
    meta = Meta(plan, [(axis, tile)])
    spans_host = plan.conv_spans()
    assert spans_host[:, 4].sum() == sum(frames)
    spans_dev = torch.from_numpy(spans_host).to(DEVICE)
    x = random_packed(80, plan, axis, 31)
    x[:, :batch.LEAD] = float('nan')
    x[:, -batch.TAIL:] = float('nan')
    for off, count in spans(plan, axis):
        x[:, off + count:off + count + (-count) % 16] = float('nan')
    weights = [synth.weights(40 + l, (80, 80, 3), 0.12) for l in range(layers)]
    biases = [synth.weights(50 + l, (80,), 0.3) for l in range(layers)]
    packs = torch.from_numpy(np.concatenate(
        [runtime.conv_winograd4_pack(w) for w in weights])).to(DEVICE)
    biases_dev = torch.from_numpy(np.concatenate(biases)).to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    x_dev = x.to(DEVICE)
    # layer by layer
    pack_size = packs.numel() // layers
    current = x_dev
    for l in range(layers):
        out = torch.full((80, plan.ld_frames), 7.0, device=DEVICE)
        runtime.check(lib.emph_conv1d_winograd4(
            current.data_ptr(), plan.ld_frames, out.data_ptr(), plan.ld_frames,
            packs[l * pack_size:].data_ptr(), biases_dev[80 * l:].data_ptr(),
            80, 80, 1 if (relu_mask >> l) & 1 else 0, tiles.data_ptr(),
            size // 4, None), 'emph_conv1d_winograd4')
        current = out
    want = current.cpu()
    # one launch
    y = torch.full((80, plan.ld_frames), 7.0, device=DEVICE)
    runtime.check(lib.emph_conv1d_stack(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        packs.data_ptr(), biases_dev.data_ptr(), layers, relu_mask,
        spans_dev.data_ptr(), len(spans_host), None, None),
        'emph_conv1d_stack')
    got = y.cpu()
    reference = x.clone()
    for off, count in spans(plan, axis):
        assert torch.equal(got[:, off:off + count], want[:, off:off + count]), \
            (count, float((got[:, off:off + count] -
                           want[:, off:off + count]).abs().max()))
        value = reference[None, :, off:off + count]
        for l in range(layers):
            value = torch.nn.functional.conv1d(
                value, torch.from_numpy(weights[l]),
                torch.from_numpy(biases[l]), padding=1)
            if (relu_mask >> l) & 1:
                value = torch.relu(value)
        scale = max(1.0, float(value.abs().max()))
        assert float((got[:, off:off + count] - value[0]).abs().max()) < \
            2e-5 * scale * layers
    # columns outside every segment are left untouched
    assert float(got[:, :batch.LEAD].min()) == 7.0
    assert lib.emph_conv1d_stack(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        packs.data_ptr(), biases_dev.data_ptr(), 4, 0, spans_dev.data_ptr(),
        len(spans_host), None, None) != 0       # (three layers at most)
    assert lib.emph_conv_stack_max_layers() == 3


@pytest.mark.parametrize('layers,relu_mask', [
    (1, 0b1), (2, 0b10), (3, 0b110), (4, 0b1110), (5, 0b11111), (5, -0b11110)])
def test_conv1d_split(layers, relu_mask):
    """emph_conv1d_split (the same group of layers on the bf16 matrix pipe,
    operands split into two bf16 pieces, direct form) against torch conv1d in
    float64 and against the fp32 kernel emph_conv1d_stack, on the span
    boundaries of `test_conv1d_stack_equals_layer_by_layer`; NaN in all
    padding.  Three products per term: 2e-5 x scale per layer (the fp32 kernel
    holds 2e-6)."""
    lib = runtime.library()
    frames = [1000, 1, 2, 3, 4, 5, 37, 64, 65, 252, 253, 256, 257, 504, 505,
              600, 3000, 130]
    if relu_mask < 0:
        relu_mask = -relu_mask
        rng = np.random.default_rng(2000 + layers)
        frames = [int(n) for n in rng.integers(1, 3001, size=40)]
    plan = ragged_plan(frames)
    axis = runtime.AXIS_FRAMES
    spans_host = plan.conv_spans()
    spans_dev = torch.from_numpy(spans_host).to(DEVICE)
    x = random_packed(80, plan, axis, 31)
    x[:, :batch.LEAD] = float('nan')
    x[:, -batch.TAIL:] = float('nan')
    for off, count in spans(plan, axis):
        x[:, off + count:off + count + (-count) % 16] = float('nan')
    weights = [synth.weights(40 + l, (80, 80, 3), 0.12) for l in range(layers)]
    biases = [synth.weights(50 + l, (80,), 0.3) for l in range(layers)]
    packs = torch.from_numpy(np.concatenate(
        [runtime.conv_split_pack(w) for w in weights])).to(DEVICE)
    plain_packs = torch.from_numpy(np.concatenate(
        [runtime.conv_winograd4_pack(w) for w in weights])).to(DEVICE)
    biases_dev = torch.from_numpy(np.concatenate(biases)).to(DEVICE)
    x_dev = x.to(DEVICE)
    y = torch.full((80, plan.ld_frames), 7.0, device=DEVICE)
    runtime.check(lib.emph_conv1d_split(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        packs.data_ptr(), biases_dev.data_ptr(), layers, relu_mask,
        spans_dev.data_ptr(), len(spans_host), None, None), 'emph_conv1d_split')
    # the fp32 kernel beside it (three layers per launch at most)
    plain, source, done = None, x_dev, 0
    pack_floats = plain_packs.numel() // layers
    while done < layers:
        size = min(3, layers - done)
        plain = torch.full((80, plan.ld_frames), 7.0, device=DEVICE)
        runtime.check(lib.emph_conv1d_stack(
            source.data_ptr(), plan.ld_frames, plain.data_ptr(), plan.ld_frames,
            plain_packs[done * pack_floats:].data_ptr(),
            biases_dev[80 * done:].data_ptr(), size, relu_mask >> done,
            spans_dev.data_ptr(), len(spans_host), None, None),
            'emph_conv1d_stack')
        source, done = plain, done + size
    got, plain = y.cpu().double(), plain.cpu().double()
    worst = worst_plain = 0.
    for off, count in spans(plan, axis):
        value = x[None, :, off:off + count].double()
        for l in range(layers):
            value = torch.nn.functional.conv1d(
                value, torch.from_numpy(weights[l]).double(),
                torch.from_numpy(biases[l]).double(), padding=1)
            if (relu_mask >> l) & 1:
                value = torch.relu(value)
        scale = max(1.0, float(value.abs().max()))
        delta = float((got[:, off:off + count] - value[0]).abs().max())
        assert delta < 2e-5 * scale * layers, (count, delta, scale)
        worst = max(worst, delta / scale)
        worst_plain = max(worst_plain, float(
            (plain[:, off:off + count] - value[0]).abs().max()) / scale)
    print(f'{layers} layers: |split - f64| / scale {worst:.2e}, '
          f'|fp32 kernel - f64| / scale {worst_plain:.2e}')
    assert float(y[:, :batch.LEAD].min()) == 7.0      # columns outside: untouched
    assert lib.emph_conv1d_split(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        packs.data_ptr(), biases_dev.data_ptr(), 6, 0, spans_dev.data_ptr(),
        len(spans_host), None, None) != 0         # (five layers at most)


@pytest.mark.parametrize('c_in,c_out,activation,max_positions', [
    (80, 80, None, 5000), (80, 80, 'relu', 150), (64, 64, None, 131),
    (48, 33, None, 5000)])
def test_conv1d_winograd4_position(c_in, c_out, activation, max_positions):
    """Conv1d(k=3, 'same') followed by PositionalEncoding (`x + pe[:T]`,
    transformer.py:45-52) in one launch: the table is channel-major, column
    t goes to position t of EVERY segment, and positions beyond the table
    (the engine raises before it gets there) are left without encoding."""
    lib = runtime.library()
    plan = ragged_plan([200, 1, 2, 3, 4, 5, 17, 64, 65, 130, 63, 66, 127])
    axis, tile = runtime.AXIS_FRAMES, 64
    meta = Meta(plan, [(axis, tile)])
    x = random_packed(c_in, plan, axis, 21)
    x[:, :batch.LEAD] = float('nan')
    weight = synth.weights(7, (c_out, c_in, 3), 0.2)
    bias = synth.weights(8, (c_out,), 0.5)
    table = torch.from_numpy(synth.weights(9, (c_out, max_positions), 1.0))
    y = torch.full((c_out, plan.ld_frames), 7.0, device=DEVICE)
    pack = torch.from_numpy(runtime.conv_winograd4_pack(weight)).to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    x_dev, bias_dev = x.to(DEVICE), torch.from_numpy(bias).to(DEVICE)
    table_dev = table.to(DEVICE)
    runtime.check(lib.emph_conv1d_winograd4_position(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        pack.data_ptr(), bias_dev.data_ptr(), c_in, c_out,
        runtime.ACTIVATIONS[activation], tiles.data_ptr(), size // 4,
        table_dev.data_ptr(), max_positions, None),
        'emph_conv1d_winograd4_position')
    plain = torch.full((c_out, plan.ld_frames), 7.0, device=DEVICE)
    runtime.check(lib.emph_conv1d_winograd4(
        x_dev.data_ptr(), plan.ld_frames, plain.data_ptr(), plan.ld_frames,
        pack.data_ptr(), bias_dev.data_ptr(), c_in, c_out,
        runtime.ACTIVATIONS[activation], tiles.data_ptr(), size // 4, None),
        'emph_conv1d_winograd4')
    y, plain = y.cpu(), plain.cpu()
    for off, count in spans(plan, axis):
        want = ACTIVATIONS[activation](torch.nn.functional.conv1d(
            x[None, :, off:off + count], torch.from_numpy(weight),
            torch.from_numpy(bias), padding=1))[0]
        covered = min(count, max_positions)
        want[:, :covered] += table[:, :covered]
        got = y[:, off:off + count]
        assert torch.isfinite(got).all()
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) < 5e-5 * scale
        # exactly the plain kernel's output plus the table (one fp32 add)
        assert torch.equal(
            got[:, :covered], plain[:, off:off + covered] + table[:, :covered])
    assert float(y[:, :batch.LEAD].min()) == 7.0
    assert lib.emph_conv1d_winograd4_position(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), plan.ld_frames,
        pack.data_ptr(), None, c_in, c_out, 0, tiles.data_ptr(), size // 4,
        None, max_positions, None) == -1


def test_conv1d_winograd_rejects_bad_arguments():
    lib = runtime.library()
    buffer = torch.zeros(4096, device=DEVICE)
    tiles = torch.zeros(4, dtype=torch.int32, device=DEVICE)
    pointer = buffer.data_ptr()
    # tile_n 16 is not a pair tile; 200 input channels do not fit in LDS
    assert lib.emph_conv1d_winograd(
        pointer, 64, pointer, 64, pointer, None, 80, 80, 0,
        tiles.data_ptr(), 1, 16, None) == -2
    assert lib.emph_conv1d_winograd(
        pointer, 64, pointer, 64, pointer, None, 200, 80, 0,
        tiles.data_ptr(), 1, 64, None) == -2
    assert b'LDS' in lib.emph_last_error()
    assert runtime.conv_winograd_lds_bytes(80, 200) > 160 * 1024


def test_conv1d_transposed_output():
    lib = runtime.library()
    plan = ragged_plan([100, 33])
    axis, tile = runtime.AXIS_FRAMES, 32
    meta = Meta(plan, [(axis, tile)])
    x = random_packed(80, plan, axis, 2)
    weight = synth.weights(3, (80, 80, 1), 0.2)
    bias = synth.weights(4, (80,), 0.5)
    y = torch.zeros((plan.ld_frames, 80), device=DEVICE)
    pack = torch.from_numpy(runtime.conv_pack(weight)).to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    x_dev, bias_dev = x.to(DEVICE), torch.from_numpy(bias).to(DEVICE)
    runtime.check(lib.emph_conv1d(
        x_dev.data_ptr(), plan.ld_frames, y.data_ptr(), 80,
        pack.data_ptr(), bias_dev.data_ptr(), 80, 80,
        1, 0, tiles.data_ptr(), size // 4, tile, 1, None), 'emph_conv1d')
    for off, count in spans(plan, axis):
        want = torch.from_numpy(weight[:, :, 0]) @ x[:, off:off + count] + \
            torch.from_numpy(bias)[:, None]
        assert float((y[off:off + count].cpu().T - want).abs().max()) < 2e-5


def test_conv1d_rejects_bad_arguments():
    lib = runtime.library()
    assert lib.emph_conv1d(None, 4, None, 4, None, None, 80, 80, 3, 0, None, 1,
                           32, 0, None) == -1
    x = torch.zeros(16, device=DEVICE)
    p = x.data_ptr()
    assert lib.emph_conv1d(p, 4, p, 4, p, None, 80, 80, 4, 0, p, 1, 32, 0,
                           None) == -2      # even kernel size
    assert b'kernel_size' in lib.emph_last_error()
    assert lib.emph_conv1d(p, 4, p, 4, p, None, 80, 80, 3, 0, p, 1, 48, 0,
                           None) == -2      # tile size
    assert lib.emph_conv1d(p, 4, p, 4, p, None, 80, 80, 3, 0, p, 0, 32, 0,
                           None) == 0       # empty batch is a no-op


###############################################################################
# segment reduce (emphases/core.py:426-469)
###############################################################################


def reduce_reference(x, bounds, mode):
    count = bounds.shape[1]
    out = torch.zeros((x.shape[0], count))
    for j in range(count):
        start, end = int(bounds[0, j]), int(bounds[1, j])
        piece = x[:, start:end]                 # python slice truncation
        if mode == 'sum':
            out[:, j] = piece.sum(1)
        elif mode == 'average':
            out[:, j] = piece.mean(1)
        elif mode == 'max':
            out[:, j] = piece.max(1).values
        else:
            out[:, j] = x[:, (start + end) // 2]
    return out


@pytest.mark.parametrize('mode', ['sum', 'average', 'max', 'center'])
@pytest.mark.parametrize('channels', [80, 7])
def test_segment_reduce(mode, channels):
    lib = runtime.library()
    frames = [300, 40, 1000]
    words = [synth.word_frames(4, 300, 1, 70),
             np.array([[0, 1, 3], [1, 3, 40]]),
             synth.word_frames(5, 1000, 8, 60)]
    plan = ragged_plan(frames, words)
    meta = Meta(plan, [])
    x = random_packed(channels, plan, runtime.AXIS_FRAMES, 9)
    out = torch.full((channels, plan.ld_words), -5.0, device=DEVICE)
    x_dev = x.to(DEVICE)
    runtime.check(lib.emph_segment_reduce(
        x_dev.data_ptr(), plan.ld_frames,
        meta.view('bounds')[0].data_ptr(), out.data_ptr(), plan.ld_words,
        channels, meta.view('table')[0].data_ptr(),
        meta.view('word_segment')[0].data_ptr(), plan.ld_words,
        runtime.REDUCTIONS[mode], None), 'emph_segment_reduce')
    out = out.cpu()
    for (off, count), (woff, wcount), bounds in zip(
            spans(plan, runtime.AXIS_FRAMES), spans(plan, runtime.AXIS_WORDS),
            words):
        want = reduce_reference(x[:, off:off + count], bounds, mode)
        got = out[:, woff:woff + wcount]
        assert float((got - want).abs().max()) < 2e-5
    assert float(out[:, :batch.LEAD].max()) == -5.0


def test_segment_reduce_edge_cases():
    """Empty word: 0 under sum, NaN under average (as the reference); an end
    beyond the chunk is truncated like a Python slice."""
    lib = runtime.library()
    words = [np.array([[0, 10, 10, 25], [10, 10, 25, 60]])]
    plan = ragged_plan([30], words)
    meta = Meta(plan, [])
    x = random_packed(8, plan, runtime.AXIS_FRAMES, 1)
    x_dev = x.to(DEVICE)
    for mode in ('sum', 'average'):
        out = torch.zeros((8, plan.ld_words), device=DEVICE)
        runtime.check(lib.emph_segment_reduce(
            x_dev.data_ptr(), plan.ld_frames,
            meta.view('bounds')[0].data_ptr(), out.data_ptr(), plan.ld_words,
            8, meta.view('table')[0].data_ptr(),
            meta.view('word_segment')[0].data_ptr(), plan.ld_words,
            runtime.REDUCTIONS[mode], None), 'emph_segment_reduce')
        off, woff = int(plan.frame_off[0]), int(plan.word_off[0])
        want = reduce_reference(x[:, off:off + 30], words[0], mode)
        got = out[:, woff:woff + 4].cpu()
        if mode == 'sum':
            assert float(got[:, 1].abs().max()) == 0.0
            assert float((got - want).abs().max()) < 1e-5
        else:
            assert torch.isnan(got[:, 1]).all() and torch.isnan(want[:, 1]).all()
            keep = [0, 2, 3]
            assert float((got[:, keep] - want[:, keep]).abs().max()) < 1e-5


def test_downsample_wrapper_matches_reference_errors():
    import emphases_amd
    from emphases_amd import config as cfg
    xs = torch.from_numpy(synth.weights(3, (2, 80, 50), 1.0))
    bounds = torch.tensor([[[0, 10, 20], [10, 20, 50]],
                           [[0, 5, 5], [5, 5, 0]]])
    lengths = torch.tensor([3, 2])
    got = emphases_amd.downsample(xs, bounds, lengths)
    assert got.shape == (2, 80, 3) and not got.is_cuda
    assert float((got[0, :, 2] - xs[0, :, 20:50].sum(1)).abs().max()) < 2e-5
    assert float(got[1, :, 1].abs().max()) == 0.0       # empty word, sum
    assert float(got[1, :, 2].abs().max()) == 0.0       # beyond word_lengths
    with pytest.raises(IndexError):                     # core.py:449-452
        emphases_amd.downsample(
            xs, bounds, lengths, cfg.Config(downsample_method='max'))


###############################################################################
# output layer + postprocess
###############################################################################


@pytest.mark.parametrize('kernel_size,post', [(3, 'bce'), (1, 'mse'), (5, None)])
def test_output_layer(kernel_size, post):
    lib = runtime.library()
    words = [synth.word_frames(1, 300, 5, 30), np.array([[0], [12]]),
             synth.word_frames(2, 90, 2, 9)]
    plan = ragged_plan([300, 12, 90], words)
    meta = Meta(plan, [])
    axis = runtime.AXIS_WORDS
    x = random_packed(80, plan, axis, 13)
    weight = synth.weights(14, (1, 80, kernel_size), 0.1)
    bias = synth.weights(15, (1,), 0.5)
    logits = torch.zeros(plan.ld_words, device=DEVICE)
    scores = torch.zeros(plan.ld_words, device=DEVICE)
    x_dev = x.to(DEVICE)
    weight_dev = torch.from_numpy(weight).to(DEVICE)
    bias_dev = torch.from_numpy(bias).to(DEVICE)
    runtime.check(lib.emph_output_layer(
        x_dev.data_ptr(), plan.ld_words, weight_dev.data_ptr(),
        bias_dev.data_ptr(), 80, kernel_size,
        meta.view('table')[0].data_ptr(),
        meta.view('word_segment')[0].data_ptr(), plan.ld_words, axis,
        runtime.POSTPROCESS[post], logits.data_ptr(), scores.data_ptr(), None),
        'emph_output_layer')
    for off, count in spans(plan, axis):
        want = torch.nn.functional.conv1d(
            x[None, :, off:off + count], torch.from_numpy(weight),
            torch.from_numpy(bias), padding=(kernel_size - 1) // 2)[0, 0]
        assert float((logits[off:off + count].cpu() - want).abs().max()) < 2e-5
        if post == 'bce':
            want = torch.sigmoid(want)
        elif post == 'mse':
            want = torch.clamp(want, 0., 1.)
        assert float((scores[off:off + count].cpu() - want).abs().max()) < 2e-6


###############################################################################
# transformer pieces
###############################################################################


@pytest.mark.parametrize('tile_n', [64, 256, 512])
@pytest.mark.parametrize('channels,heads', [(80, 2), (64, 2), (128, 2)])
def test_attention(channels, heads, tile_n):
    lib = runtime.library()
    plan = ragged_plan([130, 16, 1, 700, 65])
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile_n)])
    ld = plan.ld_frames
    qk = random_packed(2 * channels, plan, axis, 21) * 3.0
    v = torch.from_numpy(synth.weights(22, (ld, channels), 1.0))
    out = torch.zeros((channels, ld), device=DEVICE)
    tiles, size = meta.view(('tiles', axis, tile_n))
    qk_dev, v_dev = qk.to(DEVICE), v.to(DEVICE)
    runtime.check(lib.emph_attention(
        qk_dev.data_ptr(), v_dev.data_ptr(), out.data_ptr(), ld,
        channels, heads, tiles.data_ptr(), size // 4, tile_n, None, None),
        'emph_attention')
    out = out.cpu()
    d = channels // heads

    def reference(off, count, keys):
        q = qk[:channels, off:off + count].T.reshape(count, heads, d)
        k = qk[channels:, off:off + keys].T.reshape(keys, heads, d)
        vv = v[off:off + keys].reshape(keys, heads, d)
        scores = torch.einsum('qhd,khd->hqk', q, k) / np.sqrt(d)
        return torch.einsum(
            'hqk,khd->qhd', torch.softmax(scores, -1), vv).reshape(
                count, channels).T

    for off, count in spans(plan, axis):
        want = reference(off, count, count)
        assert float((out[:, off:off + count] - want).abs().max()) < 2e-5
    # key-padding mask (transformer.py:26-29): only the leading key_counts[s]
    # positions of a segment are keys; every position is still a query
    key_counts = np.array([50, 16, 1, 333, 17], dtype=np.int32)
    counts_dev = torch.from_numpy(key_counts).to(DEVICE)
    out = torch.zeros((channels, ld), device=DEVICE)
    runtime.check(lib.emph_attention(
        qk_dev.data_ptr(), v_dev.data_ptr(), out.data_ptr(), ld,
        channels, heads, tiles.data_ptr(), size // 4, tile_n,
        counts_dev.data_ptr(), None), 'emph_attention')
    out = out.cpu()
    for (off, count), keys in zip(spans(plan, axis), key_counts):
        want = reference(off, count, int(keys))
        assert float((out[:, off:off + count] - want).abs().max()) < 2e-5


@pytest.mark.parametrize('pieces', [2, 3, 32])
def test_attention_split(pieces):
    """emph_attention_split (bf16 pieces on the bf16 matrix pipe) against a
    float64 reference, with the fp32-MFMA kernel's own error beside it:
    three pieces stay within 2x of the fp32 kernel on inputs built to be
    harsh (keys whose scores tower over the others by more than 2^64, so that
    the lazy softmax reference moves); two pieces do not - that is what
    'bf16x3' trades."""
    lib = runtime.library()
    channels, heads, tile_n = 80, 2, 256
    plan = ragged_plan([130, 16, 1, 700, 65, 1000, 257])
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile_n), (axis, 64)])
    ld = plan.ld_frames
    qk = random_packed(2 * channels, plan, axis, 21) * 3.0
    # keys whose scores tower over the first block's (the lazy softmax
    # reference must MOVE, with a block's scores already issued against the
    # old one): every 97th key of the long segments is 12 x larger
    qk[channels:, 40::97] *= 12.0
    v = torch.from_numpy(synth.weights(22, (ld, channels), 1.0))
    # padding of the packed axes may hold anything
    for off, count in spans(plan, axis):
        qk[:, off + count:off + count + 3] = float('nan')
        v[off + count:off + count + 3] = float('nan')
    tiles, size = meta.view(('tiles', axis, tile_n))
    stage_tiles, stage_size = meta.view(('tiles', axis, 64))
    qk_dev, v_dev = qk.to(DEVICE), v.to(DEVICE)
    d = channels // heads
    image_bytes = lib.emph_split_kv_bytes(
        ld, len(plan.segments), channels, heads, pieces)
    assert image_bytes > 0
    # (stale bytes of another use of the scratch: every NaN pattern)
    images = torch.full((image_bytes // 2,), -1, dtype=torch.int16,
                        device=DEVICE)

    def reference(off, count, keys):
        q = qk[:channels, off:off + count].T.reshape(count, heads, d).double()
        k = qk[channels:, off:off + keys].T.reshape(keys, heads, d).double()
        vv = v[off:off + keys].reshape(keys, heads, d).double()
        scores = torch.einsum('qhd,khd->hqk', q, k) / np.sqrt(d)
        return torch.einsum(
            'hqk,khd->qhd', torch.softmax(scores, -1), vv).reshape(
                count, channels).T

    for key_counts in (None, np.array([50, 16, 1, 333, 17, 999, 200],
                                      dtype=np.int32)):
        counts_dev = None if key_counts is None else \
            torch.from_numpy(key_counts).to(DEVICE)
        pointer = None if key_counts is None else counts_dev.data_ptr()
        plain = torch.full((channels, ld), float('nan'), device=DEVICE)
        split = torch.full((channels, ld), float('nan'), device=DEVICE)
        runtime.check(lib.emph_attention(
            qk_dev.data_ptr(), v_dev.data_ptr(), plain.data_ptr(), ld,
            channels, heads, tiles.data_ptr(), size // 4, tile_n, pointer,
            None), 'emph_attention')
        runtime.check(lib.emph_split_kv(
            qk_dev.data_ptr(), v_dev.data_ptr(), ld, channels, heads,
            stage_tiles.data_ptr(), stage_size // 4, 64, pieces,
            images.data_ptr(), None), 'emph_split_kv')
        runtime.check(lib.emph_attention_split(
            qk_dev.data_ptr(), images.data_ptr(), split.data_ptr(), ld,
            channels, heads, tiles.data_ptr(), size // 4, tile_n, pointer,
            pieces, None), 'emph_attention_split')
        plain, split = plain.cpu().double(), split.cpu().double()
        worst_plain = worst_split = 0.
        for index, (off, count) in enumerate(spans(plan, axis)):
            keys = count if key_counts is None else int(key_counts[index])
            want = reference(off, count, keys)
            worst_plain = max(worst_plain, float(
                (plain[:, off:off + count] - want).abs().max()))
            worst_split = max(worst_split, float(
                (split[:, off:off + count] - want).abs().max()))
        print(f'pieces {pieces}: |split - f64| {worst_split:.2e}, '
              f'|fp32 kernel - f64| {worst_plain:.2e}')
        # (the towering keys make this a harsh regime for fp32 itself)
        assert worst_plain < 2e-5
        # two pieces: the scores' error (2^-17 of sum |q k|) sits in front of
        # an exponential, so it grows with the score range - 3e-4 here, 2e-5
        # without the towering keys; three pieces stay with the fp32 kernel
        # 32 = three pieces for the scores, two behind the softmax: 2^-17 of the
        # values and the probabilities, not amplified
        assert worst_split < (1e-3 if pieces == 2 else 6e-5 if pieces == 32
                              else max(2. * worst_plain, 2e-6))
    with pytest.raises(runtime.LibraryError, match='pieces'):
        runtime.check(lib.emph_attention_split(
            qk_dev.data_ptr(), images.data_ptr(), split.data_ptr(), ld,
            channels, heads, tiles.data_ptr(), size // 4, tile_n, None, 4,
            None), 'emph_attention_split')


def test_add_layernorm_and_position():
    lib = runtime.library()
    plan = ragged_plan([100, 37])
    ld = plan.ld_frames
    axis = runtime.AXIS_FRAMES
    x = random_packed(80, plan, axis, 31) * 4.0
    r = random_packed(80, plan, axis, 32)
    gamma = torch.from_numpy(1.0 + synth.weights(33, (80,), 0.2))
    beta = torch.from_numpy(synth.weights(34, (80,), 0.2))
    y = torch.zeros((80, ld), device=DEVICE)
    x_dev, r_dev = x.to(DEVICE), r.to(DEVICE)
    gamma_dev, beta_dev = gamma.to(DEVICE), beta.to(DEVICE)
    runtime.check(lib.emph_add_layernorm(
        x_dev.data_ptr(), r_dev.data_ptr(), y.data_ptr(), ld, 80,
        gamma_dev.data_ptr(), beta_dev.data_ptr(), 1e-5, 0, ld,
        None), 'emph_add_layernorm')
    want = torch.nn.functional.layer_norm(
        (x + r).T, (80,), gamma, beta, 1e-5).T
    assert float((y.cpu() - want).abs().max()) < 2e-5

    meta = Meta(plan, [(axis, 64)])
    table = torch.from_numpy(synth.weights(35, (5000, 80), 1.0))
    z = x.to(DEVICE).clone()
    table_dev = table.to(DEVICE)
    tiles, size = meta.view(('tiles', axis, 64))
    runtime.check(lib.emph_add_position(
        z.data_ptr(), ld, table_dev.data_ptr(), 80, 5000, tiles.data_ptr(),
        size // 4, 64, None), 'emph_add_position')
    z = z.cpu()
    for off, count in spans(plan, axis):
        want = x[:, off:off + count] + table[:count].T
        assert float((z[:, off:off + count] - want).abs().max()) == 0.0
    assert torch.equal(z[:, :batch.LEAD], x[:, :batch.LEAD])


###############################################################################
# fused position-wise half of a Transformer layer
###############################################################################


@pytest.mark.parametrize('channels,tile', [(80, 32), (80, 16), (64, 32),
                                           (64, 16)])
def test_transformer_block(channels, tile):
    lib = runtime.library()
    plan = ragged_plan([200, 1, 17, 33, 64])
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile)])
    x = random_packed(channels, plan, axis, 21)
    attended = random_packed(channels, plan, axis, 22)
    x[:, :batch.LEAD] = float('nan')
    names = ['out', 'l1', 'l2']
    weight = {n: torch.from_numpy(synth.weights(30 + i, (channels, channels), 0.3))
              for i, n in enumerate(names)}
    vector = {n: torch.from_numpy(synth.weights(40 + i, (channels,), 0.5))
              for i, n in enumerate(
                  ['b_o', 'g1', 'be1', 'b_1', 'b_2', 'g2', 'be2'])}
    vector['g1'] += 1.
    vector['g2'] += 1.
    packs = torch.from_numpy(np.concatenate([
        runtime.linear_chain_pack(weight['out'].numpy(), True),
        runtime.linear_chain_pack(weight['l1'].numpy(), False),
        runtime.linear_chain_pack(weight['l2'].numpy(), False)])).to(DEVICE)
    vectors = torch.cat([vector[n] for n in (
        'b_o', 'g1', 'be1', 'b_1', 'b_2', 'g2', 'be2')]).to(DEVICE)
    x_dev, attended_dev = x.to(DEVICE), attended.to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    runtime.check(lib.emph_transformer_block(
        attended_dev.data_ptr(), x_dev.data_ptr(), plan.ld_frames, channels,
        packs.data_ptr(), vectors.data_ptr(), 1e-5, 1, tiles.data_ptr(),
        size // 4, tile, None), 'emph_transformer_block')
    got = x_dev.cpu()
    norm = torch.nn.functional.layer_norm
    for off, count in spans(plan, axis):
        xs = x[:, off:off + count].T
        a = attended[:, off:off + count].T
        y = norm(xs + a @ weight['out'].T + vector['b_o'], (channels,),
                 vector['g1'], vector['be1'], 1e-5)
        h = torch.relu(y @ weight['l1'].T + vector['b_1'])
        want = norm(y + h @ weight['l2'].T + vector['b_2'], (channels,),
                    vector['g2'], vector['be2'], 1e-5).T
        assert float((got[:, off:off + count] - want).abs().max()) < 5e-5
    assert torch.isnan(got[:, :batch.LEAD]).all()   # padding untouched


@pytest.mark.parametrize('channels,tile', [(80, 32), (80, 16), (64, 32)])
def test_transformer_block_qkv(channels, tile):
    """The position-wise half of an encoder layer AND the next layer's Q, K, V
    projections in one launch: x as emph_transformer_block leaves it (bit for
    bit - the same instructions), Q | K | V of that x as torch computes them."""
    lib = runtime.library()
    plan = ragged_plan([200, 1, 17, 33, 64, 31])
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile)])
    x = random_packed(channels, plan, axis, 23)
    attended = random_packed(channels, plan, axis, 24)
    names = ['out', 'l1', 'l2']
    weight = {n: torch.from_numpy(synth.weights(50 + i, (channels, channels), 0.3))
              for i, n in enumerate(names)}
    vector = {n: torch.from_numpy(synth.weights(60 + i, (channels,), 0.5))
              for i, n in enumerate(
                  ['b_o', 'g1', 'be1', 'b_1', 'b_2', 'g2', 'be2'])}
    vector['g1'] += 1.
    vector['g2'] += 1.
    in_w = torch.from_numpy(synth.weights(70, (3 * channels, channels), 0.3))
    in_b = torch.from_numpy(synth.weights(71, (3 * channels,), 0.5))
    block_packs = np.concatenate([
        runtime.linear_chain_pack(weight['out'].numpy(), True),
        runtime.linear_chain_pack(weight['l1'].numpy(), False),
        runtime.linear_chain_pack(weight['l2'].numpy(), False)])
    packs = torch.from_numpy(np.concatenate([block_packs] + [
        runtime.linear_chain_pack(
            in_w[part * channels:(part + 1) * channels].numpy(), False)
        for part in range(3)])).to(DEVICE)
    seven = torch.cat([vector[n] for n in (
        'b_o', 'g1', 'be1', 'b_1', 'b_2', 'g2', 'be2')])
    vectors = torch.cat([seven, in_b]).to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    attended_dev = attended.to(DEVICE)
    x_dev = x.to(DEVICE)
    qk = torch.full((2 * channels, plan.ld_frames), 7.0, device=DEVICE)
    v = torch.full((plan.ld_frames, channels), 7.0, device=DEVICE)
    runtime.check(lib.emph_transformer_block_qkv(
        attended_dev.data_ptr(), x_dev.data_ptr(), plan.ld_frames, channels,
        packs.data_ptr(), vectors.data_ptr(), 1e-5, 1, tiles.data_ptr(),
        size // 4, tile, qk.data_ptr(), v.data_ptr(), None),
        'emph_transformer_block_qkv')
    plain = x.to(DEVICE)
    plain_packs = torch.from_numpy(block_packs).to(DEVICE)
    plain_vectors = seven.to(DEVICE)
    runtime.check(lib.emph_transformer_block(
        attended_dev.data_ptr(), plain.data_ptr(), plan.ld_frames, channels,
        plain_packs.data_ptr(), plain_vectors.data_ptr(), 1e-5, 1,
        tiles.data_ptr(), size // 4, tile, None), 'emph_transformer_block')
    got, plain, qk, v = x_dev.cpu(), plain.cpu(), qk.cpu(), v.cpu()
    for off, count in spans(plan, axis):
        assert torch.equal(got[:, off:off + count], plain[:, off:off + count])
        want = in_w @ got[:, off:off + count] + in_b[:, None]
        assert float((qk[:, off:off + count] -
                      want[:2 * channels]).abs().max()) < 2e-5
        assert float((v[off:off + count].T -
                      want[2 * channels:]).abs().max()) < 2e-5
    assert float(qk[:, :batch.LEAD].min()) == 7.0
    assert float(v[:batch.LEAD].min()) == 7.0
    assert lib.emph_transformer_block_qkv(
        attended_dev.data_ptr(), x_dev.data_ptr(), plan.ld_frames, channels,
        packs.data_ptr(), vectors.data_ptr(), 1e-5, 1, tiles.data_ptr(),
        size // 4, tile, None, None, None) == -1


@pytest.mark.parametrize('channels,tile', [(80, 32), (64, 16)])
def test_qkv_projection(channels, tile):
    lib = runtime.library()
    plan = ragged_plan([200, 1, 17, 33])
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile)])
    x = random_packed(channels, plan, axis, 5)
    weight = torch.from_numpy(synth.weights(6, (3 * channels, channels), 0.3))
    bias = torch.from_numpy(synth.weights(7, (3 * channels,), 0.5))
    packs = torch.from_numpy(np.concatenate([
        runtime.linear_chain_pack(
            weight[part * channels:(part + 1) * channels].numpy(), True)
        for part in range(3)])).to(DEVICE)
    qk = torch.full((2 * channels, plan.ld_frames), 7.0, device=DEVICE)
    v = torch.full((plan.ld_frames, channels), 7.0, device=DEVICE)
    x_dev, bias_dev = x.to(DEVICE), bias.to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    runtime.check(lib.emph_qkv_projection(
        x_dev.data_ptr(), plan.ld_frames, qk.data_ptr(), v.data_ptr(),
        channels, packs.data_ptr(), bias_dev.data_ptr(), tiles.data_ptr(),
        size // 4, tile, None), 'emph_qkv_projection')
    qk, v = qk.cpu(), v.cpu()
    for off, count in spans(plan, axis):
        want = weight @ x[:, off:off + count] + bias[:, None]
        assert float((qk[:, off:off + count] -
                      want[:2 * channels]).abs().max()) < 2e-5
        assert float((v[off:off + count].T -
                      want[2 * channels:]).abs().max()) < 2e-5
    assert float(qk[:, :batch.LEAD].min()) == 7.0
    assert float(v[:batch.LEAD].min()) == 7.0


@pytest.mark.parametrize('pieces,budget', [(2, 6e-5), (3, 5e-6)])
def test_transformer_block_split(pieces, budget):
    """emph_transformer_block_split (three / six bf16 products per fp32 product)
    against a float64 reference, with the fp32 kernel's own error beside it."""
    lib = runtime.library()
    channels, tile = 80, 32
    plan = ragged_plan([200, 1, 17, 33, 64, 31, 1000])
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile)])
    x = random_packed(channels, plan, axis, 21)
    attended = random_packed(channels, plan, axis, 22)
    x[:, :batch.LEAD] = float('nan')
    names = ['out', 'l1', 'l2']
    weight = {n: torch.from_numpy(synth.weights(30 + i, (channels, channels), 0.3))
              for i, n in enumerate(names)}
    order = ['b_o', 'g1', 'be1', 'b_1', 'b_2', 'g2', 'be2']
    vector = {n: torch.from_numpy(synth.weights(40 + i, (channels,), 0.5))
              for i, n in enumerate(order)}
    vector['g1'] += 1.
    vector['g2'] += 1.
    split_packs = torch.from_numpy(np.concatenate([
        runtime.linear_split_pack(weight[n].numpy(), pieces)
        for n in names])).to(DEVICE)
    plain_packs = torch.from_numpy(np.concatenate([
        runtime.linear_chain_pack(weight['out'].numpy(), True),
        runtime.linear_chain_pack(weight['l1'].numpy(), False),
        runtime.linear_chain_pack(weight['l2'].numpy(), False)])).to(DEVICE)
    vectors = torch.cat([vector[n] for n in order]).to(DEVICE)
    x_dev, plain, attended_dev = x.to(DEVICE), x.to(DEVICE), attended.to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    runtime.check(lib.emph_transformer_block_split(
        attended_dev.data_ptr(), x_dev.data_ptr(), plan.ld_frames, channels,
        split_packs.data_ptr(), pieces, vectors.data_ptr(), 1e-5, 1,
        tiles.data_ptr(), size // 4, tile, None),
        'emph_transformer_block_split')
    runtime.check(lib.emph_transformer_block(
        attended_dev.data_ptr(), plain.data_ptr(), plan.ld_frames, channels,
        plain_packs.data_ptr(), vectors.data_ptr(), 1e-5, 1, tiles.data_ptr(),
        size // 4, tile, None), 'emph_transformer_block')
    got, plain = x_dev.cpu(), plain.cpu()
    norm = torch.nn.functional.layer_norm
    wd = {n: w.double() for n, w in weight.items()}
    vd = {n: w.double() for n, w in vector.items()}
    worst = worst_plain = 0.
    for off, count in spans(plan, axis):
        xs = x[:, off:off + count].T.double()
        a = attended[:, off:off + count].T.double()
        y = norm(xs + a @ wd['out'].T + vd['b_o'], (channels,),
                 vd['g1'], vd['be1'], 1e-5)
        h = torch.relu(y @ wd['l1'].T + vd['b_1'])
        want = norm(y + h @ wd['l2'].T + vd['b_2'], (channels,),
                    vd['g2'], vd['be2'], 1e-5).T
        worst = max(worst, float((got[:, off:off + count] - want).abs().max()))
        worst_plain = max(worst_plain, float(
            (plain[:, off:off + count] - want).abs().max()))
    print(f'block, {pieces} pieces: {worst:.2e}  fp32 kernel {worst_plain:.2e}')
    assert worst < budget
    assert torch.isnan(got[:, :batch.LEAD]).all()   # padding untouched
    # ... and untouched between the segments
    keep = torch.ones(plan.ld_frames, dtype=torch.bool)
    for off, count in spans(plan, axis):
        keep[off:off + count] = False
    keep[:batch.LEAD] = False
    assert torch.equal(got[:, keep], x[:, keep])


@pytest.mark.parametrize('pieces,budget', [(2, 4e-5), (3, 2e-6)])
def test_qkv_projection_split(pieces, budget):
    lib = runtime.library()
    channels, tile = 80, 32
    plan = ragged_plan([200, 1, 17, 33, 700])
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile)])
    x = random_packed(channels, plan, axis, 5)
    weight = torch.from_numpy(synth.weights(6, (3 * channels, channels), 0.3))
    bias = torch.from_numpy(synth.weights(7, (3 * channels,), 0.5))
    packs = torch.from_numpy(np.concatenate([
        runtime.linear_split_pack(
            weight[part * channels:(part + 1) * channels].numpy(), pieces)
        for part in range(3)])).to(DEVICE)
    qk = torch.full((2 * channels, plan.ld_frames), 7.0, device=DEVICE)
    v = torch.full((plan.ld_frames, channels), 7.0, device=DEVICE)
    x_dev, bias_dev = x.to(DEVICE), bias.to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    runtime.check(lib.emph_qkv_projection_split(
        x_dev.data_ptr(), plan.ld_frames, qk.data_ptr(), v.data_ptr(),
        channels, packs.data_ptr(), pieces, bias_dev.data_ptr(),
        tiles.data_ptr(), size // 4, tile, None), 'emph_qkv_projection_split')
    qk, v = qk.cpu(), v.cpu()
    worst = 0.
    for off, count in spans(plan, axis):
        want = weight.double() @ x[:, off:off + count].double() + \
            bias.double()[:, None]
        worst = max(worst, float((qk[:, off:off + count] -
                                  want[:2 * channels]).abs().max()),
                    float((v[off:off + count].T -
                           want[2 * channels:]).abs().max()))
    print(f'qkv, {pieces} pieces: {worst:.2e}')
    assert worst < budget
    assert float(qk[:, :batch.LEAD].min()) == 7.0
    assert float(v[:batch.LEAD].min()) == 7.0


@pytest.mark.parametrize('pieces,attention_pieces', [(2, 32), (3, 3), (2, 2)])
def test_qkv_projection_split_images(pieces, attention_pieces):
    """Q and the images of split keys and values straight out of the projection
    kernel: what emph_split_kv makes of emph_qkv_projection_split's fp32 K and V
    - the constant parts and the zeros behind every segment's end included."""
    lib = runtime.library()
    channels, heads, tile = 80, 2, 32
    # (segments ending in the first and in the second half of a stage, at a
    # stage's end, one shorter than a tile)
    plan = ragged_plan([200, 1, 17, 33, 700, 64, 32, 96, 321])
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile), (axis, 64)])
    ld = plan.ld_frames
    x = random_packed(channels, plan, axis, 5)
    padding = torch.ones(ld, dtype=torch.bool)
    for off, count in spans(plan, axis):
        padding[off:off + count] = False
    x[:, padding] = float('nan')        # (what lies between segments)
    weight = torch.from_numpy(synth.weights(6, (3 * channels, channels), 0.3))
    bias = torch.from_numpy(synth.weights(7, (3 * channels,), 0.5))
    packs = torch.from_numpy(np.concatenate([
        runtime.linear_split_pack(
            weight[part * channels:(part + 1) * channels].numpy(), pieces)
        for part in range(3)])).to(DEVICE)
    x_dev, bias_dev = x.to(DEVICE), bias.to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    stage_tiles, stage_size = meta.view(('tiles', axis, 64))
    image_bytes = lib.emph_split_kv_bytes(
        ld, len(plan.segments), channels, heads, attention_pieces)
    # the two-pass path
    qk = torch.full((2 * channels, ld), 7.0, device=DEVICE)
    v = torch.full((ld, channels), 7.0, device=DEVICE)
    runtime.check(lib.emph_qkv_projection_split(
        x_dev.data_ptr(), ld, qk.data_ptr(), v.data_ptr(), channels,
        packs.data_ptr(), pieces, bias_dev.data_ptr(), tiles.data_ptr(),
        size // 4, tile, None), 'emph_qkv_projection_split')
    want = torch.full((image_bytes // 2,), 0x5555, dtype=torch.int16,
                      device=DEVICE)
    runtime.check(lib.emph_split_kv(
        qk.data_ptr(), v.data_ptr(), ld, channels, heads,
        stage_tiles.data_ptr(), stage_size // 4, 64, attention_pieces,
        want.data_ptr(), None), 'emph_split_kv')
    # ... and the one launch
    q_only = torch.full((2 * channels, ld), 7.0, device=DEVICE)
    got = torch.full((image_bytes // 2,), 0x5555, dtype=torch.int16,
                     device=DEVICE)
    runtime.check(lib.emph_qkv_projection_split_images(
        x_dev.data_ptr(), ld, q_only.data_ptr(), got.data_ptr(), channels,
        heads, packs.data_ptr(), pieces, attention_pieces,
        bias_dev.data_ptr(), tiles.data_ptr(), size // 4, tile, None),
        'emph_qkv_projection_split_images')
    assert torch.equal(q_only[:channels], qk[:channels])
    assert float(q_only[channels:].min()) == 7.0       # K's rows: not touched
    same = torch.equal(got, want)
    print('images bit for bit:', same)
    if not same:
        # the same pieces of values that differ in the last bit of an fp32
        # sum (the V product runs with its operands swapped): decoded
        pk, pv = {2: (2, 2), 3: (3, 3), 32: (3, 2)}[attention_pieces]
        key_halfs, value_halfs = 6 * 64 * 8, 8 * 42 * 8

        def decode(images):
            bits = images.cpu().numpy().view(np.uint16).astype(np.uint32) << 16
            stage = bits.view(np.float32).astype(np.float64).reshape(
                -1, pk * key_halfs + pv * value_halfs)
            keys = stage[:, :pk * key_halfs].reshape(-1, pk, key_halfs).sum(1)
            values = stage[:, pk * key_halfs:].reshape(
                -1, pv, value_halfs).sum(1)
            return keys, values
        for a, b in zip(decode(got), decode(want)):
            assert np.abs(a - b).max() < 2e-6


@pytest.mark.parametrize('pieces,attention_pieces', [
    (3, 32), (2, 32), (3, 3), (2, 2), (3, None), (2, None)])
def test_transformer_block_qkv_split(pieces, attention_pieces):
    """emph_transformer_block_qkv_split = emph_transformer_block_split followed by
    the next layer's emph_qkv_projection_split (attention_pieces None) or
    emph_qkv_projection_split_images, BIT FOR BIT: x, Q, K, V / the images - on a
    ragged axis with more tiles than one round of the grid's waves, segments ending
    anywhere in a stage, NaN between the segments."""
    lib = runtime.library()
    channels, heads, tile = 80, 2, 32
    plan = ragged_plan([200, 1, 17, 33, 700, 64, 32, 96, 321] + [1000] * 40)
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile)])
    ld = plan.ld_frames
    x = random_packed(channels, plan, axis, 21)
    attended = random_packed(channels, plan, axis, 22)
    padding = torch.ones(ld, dtype=torch.bool)
    for off, count in spans(plan, axis):
        padding[off:off + count] = False
    x[:, padding] = float('nan')
    names = ['out', 'l1', 'l2']
    weight = {n: torch.from_numpy(synth.weights(30 + i, (channels, channels), 0.3))
              for i, n in enumerate(names)}
    order = ['b_o', 'g1', 'be1', 'b_1', 'b_2', 'g2', 'be2']
    vector = {n: torch.from_numpy(synth.weights(40 + i, (channels,), 0.5))
              for i, n in enumerate(order)}
    vector['g1'] += 1.
    vector['g2'] += 1.
    block_packs = torch.from_numpy(np.concatenate([
        runtime.linear_split_pack(weight[n].numpy(), pieces)
        for n in names])).to(DEVICE)
    vectors = torch.cat([vector[n] for n in order]).to(DEVICE)
    in_proj = torch.from_numpy(synth.weights(6, (3 * channels, channels), 0.3))
    bias = torch.from_numpy(synth.weights(7, (3 * channels,), 0.5)).to(DEVICE)
    qkv_packs = torch.from_numpy(np.concatenate([
        runtime.linear_split_pack(
            in_proj[part * channels:(part + 1) * channels].numpy(), pieces)
        for part in range(3)])).to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    attended_dev = attended.to(DEVICE)
    image_bytes = 0 if attention_pieces is None else lib.emph_split_kv_bytes(
        ld, len(plan.segments), channels, heads, attention_pieces)

    def buffers():
        return (x.to(DEVICE), torch.full((2 * channels, ld), 7.0, device=DEVICE),
                torch.full((ld, channels), 7.0, device=DEVICE),
                torch.full((max(image_bytes // 2, 8),), 0x5555, dtype=torch.int16,
                           device=DEVICE))
    # two launches
    x_two, qk_two, v_two, images_two = buffers()
    runtime.check(lib.emph_transformer_block_split(
        attended_dev.data_ptr(), x_two.data_ptr(), ld, channels,
        block_packs.data_ptr(), pieces, vectors.data_ptr(), 1e-5, 1,
        tiles.data_ptr(), size // 4, tile, None), 'emph_transformer_block_split')
    if attention_pieces is None:
        runtime.check(lib.emph_qkv_projection_split(
            x_two.data_ptr(), ld, qk_two.data_ptr(), v_two.data_ptr(), channels,
            qkv_packs.data_ptr(), pieces, bias.data_ptr(), tiles.data_ptr(),
            size // 4, tile, None), 'emph_qkv_projection_split')
    else:
        runtime.check(lib.emph_qkv_projection_split_images(
            x_two.data_ptr(), ld, qk_two.data_ptr(), images_two.data_ptr(),
            channels, heads, qkv_packs.data_ptr(), pieces, attention_pieces,
            bias.data_ptr(), tiles.data_ptr(), size // 4, tile, None),
            'emph_qkv_projection_split_images')
    # one launch
    x_one, qk_one, v_one, images_one = buffers()
    runtime.check(lib.emph_transformer_block_qkv_split(
        attended_dev.data_ptr(), x_one.data_ptr(), ld, channels, heads,
        block_packs.data_ptr(), qkv_packs.data_ptr(), pieces,
        attention_pieces or 0, vectors.data_ptr(), bias.data_ptr(), 1e-5, 1,
        tiles.data_ptr(), size // 4, tile, qk_one.data_ptr(), v_one.data_ptr(),
        None if attention_pieces is None else images_one.data_ptr(), None),
        'emph_transformer_block_qkv_split')
    torch.cuda.synchronize()
    live = ~padding
    assert torch.equal(x_one.cpu()[:, live], x_two.cpu()[:, live])
    assert torch.isnan(x_one.cpu()[:, padding]).all()      # padding untouched
    assert torch.equal(qk_one, qk_two)
    assert torch.equal(v_one, v_two)
    assert torch.equal(images_one, images_two)
    assert not torch.isnan(qk_one[:channels].cpu()[:, live]).any()
    if attention_pieces is None:
        assert float(v_one.cpu()[live].min()) != 7.0
    else:
        assert float(qk_one[channels:].min()) == 7.0       # K's rows: not touched
        assert float(v_one.min()) == 7.0
    # a handful of tiles (fewer than the waves of one workgroup) and none at all
    few = 3
    x_few, qk_few, v_few, images_few = buffers()
    runtime.check(lib.emph_transformer_block_qkv_split(
        attended_dev.data_ptr(), x_few.data_ptr(), ld, channels, heads,
        block_packs.data_ptr(), qkv_packs.data_ptr(), pieces,
        attention_pieces or 0, vectors.data_ptr(), bias.data_ptr(), 1e-5, 1,
        tiles.data_ptr(), few, tile, qk_few.data_ptr(), v_few.data_ptr(),
        None if attention_pieces is None else images_few.data_ptr(), None),
        'emph_transformer_block_qkv_split')
    first = plan.frame_off[0]
    assert torch.equal(x_few[:, first:first + 96], x_one[:, first:first + 96])
    assert torch.equal(qk_few[:channels, first:first + 96],
                       qk_one[:channels, first:first + 96])
    assert lib.emph_transformer_block_qkv_split(
        attended_dev.data_ptr(), x_few.data_ptr(), ld, channels, heads,
        block_packs.data_ptr(), qkv_packs.data_ptr(), pieces, 0,
        vectors.data_ptr(), bias.data_ptr(), 1e-5, 1, tiles.data_ptr(), 0, tile,
        qk_few.data_ptr(), v_few.data_ptr(), None, None) == 0


@pytest.mark.parametrize('pieces,attention_pieces,block_budget,qkv_budget', [
    (3, 32, 5e-6, 5e-6), (2, 32, 8e-5, 8e-5), (3, 3, 5e-6, 5e-6),
    (2, 2, 8e-5, 8e-5)])
def test_position_wise_split(pieces, attention_pieces, block_budget, qkv_budget):
    """emph_position_wise_split (tiles of 16 positions, csrc/block_split16.hip) in
    its three shapes on a ragged axis with NaN between the segments and more tiles
    than one round of the grid's waves: the block against a float64 reference; the
    projections of its output against float64 and, as split images, against what
    emph_split_kv makes of the fp32 ones; block + projections as ONE launch bit for
    bit the two."""
    lib = runtime.library()
    channels, heads, tile = 80, 2, 16
    plan = ragged_plan([200, 1, 17, 33, 700, 64, 32, 96, 321, 15, 16, 48, 49] +
                       [1000] * 40)
    axis = runtime.AXIS_FRAMES
    meta = Meta(plan, [(axis, tile), (axis, 64)])
    ld = plan.ld_frames
    x = random_packed(channels, plan, axis, 21)
    attended = random_packed(channels, plan, axis, 22)
    padding = torch.ones(ld, dtype=torch.bool)
    for off, count in spans(plan, axis):
        padding[off:off + count] = False
    live = ~padding
    x[:, padding] = float('nan')
    names = ['out', 'l1', 'l2']
    weight = {n: torch.from_numpy(synth.weights(30 + i, (channels, channels), 0.3))
              for i, n in enumerate(names)}
    order = ['b_o', 'g1', 'be1', 'b_1', 'b_2', 'g2', 'be2']
    vector = {n: torch.from_numpy(synth.weights(40 + i, (channels,), 0.5))
              for i, n in enumerate(order)}
    vector['g1'] += 1.
    vector['g2'] += 1.
    block_packs = torch.from_numpy(np.concatenate([
        runtime.linear_split_pack(weight[n].numpy(), pieces, 16)
        for n in names])).to(DEVICE)
    vectors = torch.cat([vector[n] for n in order]).to(DEVICE)
    in_proj = torch.from_numpy(synth.weights(6, (3 * channels, channels), 0.3))
    bias = torch.from_numpy(synth.weights(7, (3 * channels,), 0.5))
    bias_dev = bias.to(DEVICE)
    qkv_packs = torch.from_numpy(np.concatenate([
        runtime.linear_split_pack(
            in_proj[part * channels:(part + 1) * channels].numpy(), pieces, 16)
        for part in range(3)])).to(DEVICE)
    tiles, size = meta.view(('tiles', axis, tile))
    stage_tiles, stage_size = meta.view(('tiles', axis, 64))
    attended_dev = attended.to(DEVICE)
    image_bytes = lib.emph_split_kv_bytes(
        ld, len(plan.segments), channels, heads, attention_pieces)

    def call(do_block, do_qkv, x_dev, qk, v, images, n_tiles=size // 4):
        runtime.check(lib.emph_position_wise_split(
            attended_dev.data_ptr() if do_block else None, x_dev.data_ptr(), ld,
            channels, heads, block_packs.data_ptr() if do_block else None,
            vectors.data_ptr() if do_block else None,
            qkv_packs.data_ptr() if do_qkv else None,
            bias_dev.data_ptr() if do_qkv else None, pieces, attention_pieces,
            1e-5, 1, tiles.data_ptr(), n_tiles, tile,
            None if qk is None else qk.data_ptr(),
            None if v is None else v.data_ptr(),
            None if images is None else images.data_ptr(), None),
            'emph_position_wise_split')

    def outputs():
        return (torch.full((2 * channels, ld), 7.0, device=DEVICE),
                torch.full((ld, channels), 7.0, device=DEVICE),
                torch.full((image_bytes // 2,), 0x5555, dtype=torch.int16,
                           device=DEVICE))
    # 1. the block alone
    x_block = x.to(DEVICE)
    call(True, False, x_block, None, None, None)
    got = x_block.cpu()
    norm = torch.nn.functional.layer_norm
    wd = {n: w.double() for n, w in weight.items()}
    vd = {n: w.double() for n, w in vector.items()}
    worst = 0.
    for off, count in spans(plan, axis):
        xs = x[:, off:off + count].T.double()
        a = attended[:, off:off + count].T.double()
        y = norm(xs + a @ wd['out'].T + vd['b_o'], (channels,),
                 vd['g1'], vd['be1'], 1e-5)
        h = torch.relu(y @ wd['l1'].T + vd['b_1'])
        want = norm(y + h @ wd['l2'].T + vd['b_2'], (channels,),
                    vd['g2'], vd['be2'], 1e-5).T
        worst = max(worst, float((got[:, off:off + count] - want).abs().max()))
    print(f'block16, {pieces} pieces: {worst:.2e}')
    assert worst < block_budget
    assert torch.isnan(got[:, padding]).all()            # padding untouched
    # 2. the projections of that x alone, as fp32 rows
    qk_rows, v_rows, _ = outputs()
    call(False, True, x_block, qk_rows, v_rows, None)
    qk_host, v_host = qk_rows.cpu(), v_rows.cpu()
    worst = 0.
    for off, count in spans(plan, axis):
        want = in_proj.double() @ got[:, off:off + count].double() + \
            bias.double()[:, None]
        worst = max(worst,
                    float((qk_host[:, off:off + count] -
                           want[:2 * channels]).abs().max()),
                    float((v_host[off:off + count].T -
                           want[2 * channels:]).abs().max()))
    print(f'qkv16, {pieces} pieces: {worst:.2e}')
    assert worst < qkv_budget
    assert float(qk_host[:, padding].min()) == 7.0
    assert float(v_host[padding].min()) == 7.0
    # 3. ... as Q + split images: what emph_split_kv makes of the fp32 K and V
    want_images = torch.full((image_bytes // 2,), 0x5555, dtype=torch.int16,
                             device=DEVICE)
    runtime.check(lib.emph_split_kv(
        qk_rows.data_ptr(), v_rows.data_ptr(), ld, channels, heads,
        stage_tiles.data_ptr(), stage_size // 4, 64, attention_pieces,
        want_images.data_ptr(), None), 'emph_split_kv')
    q_only, v_unused, got_images = outputs()
    call(False, True, x_block, q_only, None, got_images)
    assert torch.equal(q_only[:channels], qk_rows[:channels])
    assert float(q_only[channels:].min()) == 7.0           # K's rows: not touched
    same = torch.equal(got_images, want_images)
    print('images16 bit for bit:', same)
    assert same
    # 4. block + projections in ONE launch: bit for bit the two
    for with_images in (False, True):
        x_one = x.to(DEVICE)
        qk_one, v_one, images_one = outputs()
        call(True, True, x_one, qk_one, None if with_images else v_one,
             images_one if with_images else None)
        torch.cuda.synchronize()
        assert torch.equal(x_one.cpu()[:, live], got[:, live])
        assert torch.isnan(x_one.cpu()[:, padding]).all()
        if with_images:
            assert torch.equal(qk_one, q_only)
            assert torch.equal(images_one, got_images)
        else:
            assert torch.equal(qk_one, qk_rows)
            assert torch.equal(v_one, v_rows)
    # a handful of tiles (fewer than the waves of one workgroup) and none at all
    x_few = x.to(DEVICE)
    qk_few, v_few, _ = outputs()
    call(True, True, x_few, qk_few, v_few, None, n_tiles=3)
    first = plan.frame_off[0]
    assert torch.equal(x_few[:, first:first + 48], x_block[:, first:first + 48])
    assert torch.equal(qk_few[:, first:first + 48], qk_rows[:, first:first + 48])
    call(True, True, x_few, qk_few, v_few, None, n_tiles=0)
    assert lib.emph_position_wise_split(
        None, x_few.data_ptr(), ld, channels, heads, None, None, None, None,
        pieces, attention_pieces, 1e-5, 1, tiles.data_ptr(), 3, tile, None, None,
        None, None) == -1
    # rows are addressed with 32-bit byte offsets: a packed axis of 2^22 columns is
    # refused (EMPH_ERANGE; the engine hands such a batch to the fp32 kernels)
    assert lib.emph_position_wise_split(
        attended_dev.data_ptr(), x_few.data_ptr(), 1 << 22, channels, heads,
        block_packs.data_ptr(), vectors.data_ptr(), None, None, pieces,
        attention_pieces, 1e-5, 1, tiles.data_ptr(), 3, tile, None, None, None,
        None) == -2
    assert b'2^22' in lib.emph_last_error()


###############################################################################
# feature rows
###############################################################################


@pytest.mark.parametrize('normalize', [False, True])
def test_pitch_rows(normalize):
    """emph_pitch_rows against the torch ops of data/preprocess/core.py:94-106."""
    lib = runtime.library()
    ld = 1000 + 144
    pitch = torch.from_numpy(40.0 + 510.0 * np.abs(
        synth.weights(71, (ld,), 1.0)))
    periodicity = torch.from_numpy(np.abs(synth.weights(72, (ld,), 1.0)))
    out = torch.full((83, ld), 5.0, device=DEVICE)
    pitch_dev, periodicity_dev = pitch.to(DEVICE), periodicity.to(DEVICE)
    logfmin = torch.log2(torch.tensor(40.))
    logfmax = torch.log2(torch.tensor(550.))
    runtime.check(lib.emph_pitch_rows(
        pitch_dev.data_ptr(), periodicity_dev.data_ptr(), out.data_ptr(), ld,
        80, 81, int(normalize), float(logfmin), float(logfmax), None),
        'emph_pitch_rows')
    out = out.cpu()
    want = torch.log2(pitch)
    if normalize:
        want = (want - logfmin) / (logfmax - logfmin)
    assert float((out[80] - want).abs().max()) < 1e-6
    assert torch.equal(out[81], periodicity)
    assert float(out[:80].min()) == 5.0 and float(out[82].min()) == 5.0
    # periodicity only, in row 80 (PITCH_FEATURE off)
    out = torch.full((81, ld), 5.0, device=DEVICE)
    runtime.check(lib.emph_pitch_rows(
        None, periodicity_dev.data_ptr(), out.data_ptr(), ld, -1, 80, 0,
        float(logfmin), float(logfmax), None), 'emph_pitch_rows')
    assert torch.equal(out[80].cpu(), periodicity)
    assert lib.emph_pitch_rows(
        None, None, out.data_ptr(), ld, 80, -1, 0, 0., 1., None) == -1


def test_logmel_against_torch_stft():
    """emph_logmel alone against the ATen ops of mels.py:16-109 (reflect pad,
    torch.stft, magnitude, mel matmul, log) on ragged chunks: a chunk that
    starts inside the zero padding, one that ends in it, a minimal 433-sample
    chunk, and interior chunks of one utterance."""
    from emphases_amd import engine as engine_module, melbasis
    engine = engine_module.Engine(device=0)
    lib = runtime.library()
    samples = 160 * 230
    audio = torch.from_numpy(synth.audio(61, 230))[0]
    padded = torch.nn.functional.pad(audio, (432, 432))
    # (start in the padded signal, length)
    chunks = [(0, 160 * 50), (160 * 50, 160 * 77), (160 * 127, 433),
              (160 * 130, samples + 864 - 160 * 130), (160 * 3, 160 * 200 + 7)]
    segments = [batch.Segment(
        0, 0, 1, start, length, 1 + (length + 864 - 1024) // 160,
        np.array([[0], [1]], dtype=np.int64)) for start, length in chunks]
    plan = batch.Plan(segments, [0], [samples])
    meta = engine.upload(plan)
    features = engine.features(audio.to(DEVICE), plan, meta).cpu()
    basis = torch.from_numpy(melbasis.default().dense)
    window = torch.hann_window(1024)
    for (start, length), off, count in zip(
            chunks, plan.frame_off, plan.frames):
        piece = torch.nn.functional.pad(
            padded[None, None, start:start + length], (432, 432),
            mode='reflect')[0, 0]
        spectrum = torch.stft(
            piece, 1024, hop_length=160, window=window, center=False,
            return_complex=True)
        magnitude = torch.sqrt(
            torch.view_as_real(spectrum).pow(2).sum(-1) + 1e-6)
        want = torch.log(torch.clamp(basis @ magnitude, min=1e-5))
        assert want.shape == (80, count)
        got = features[:, off:off + count]
        assert float((got - want).abs().max()) < 5e-4
        loud = want > -6.0          # away from the 1e-6 magnitude floor
        assert float((got - want)[loud].abs().max()) < 2e-5


@pytest.mark.parametrize('kernel_size,layers,activation,post', [
    (3, 6, 'relu', 'bce'), (5, 6, 'gelu', 'mse'), (1, 6, 'silu', 'bce'),
    (3, 0, 'relu', 'bce'), (5, 3, 'leaky_relu', None)])
def test_word_decoder(kernel_size, layers, activation, post):
    """emph_word_decoder alone against torch conv1d stacks: segments shorter
    than, equal to and several times longer than the workgroup's word window
    (halo recompute between windows), NaN in the padding columns."""
    lib = runtime.library()
    channels = 80
    block = int(lib.emph_word_decoder_block(layers, kernel_size, kernel_size))
    # ... and every length from 31 to 52: at most 32 words are one narrow
    # window, 33 .. 2 (32 - halo) two of them, longer ones a full 64-word window
    counts = [block, 1, 2, block + 1, 3 * block + 5, 7, 2 * block, 64, 65] + \
        list(range(31, 53))
    bounds = [np.stack([np.arange(n), np.arange(n) + 1]).astype(np.int64)
              for n in counts]
    plan = ragged_plan(counts, bounds)
    axis = runtime.AXIS_WORDS
    request = (runtime.AXIS_DECODER, (layers, kernel_size, kernel_size))
    meta = Meta(plan, [request])
    table = plan.tiles(*request)
    halo = (64 - block) // 2
    expected = sum(
        2 if 32 < n <= 2 * (32 - halo) else 1 if n <= 64 else -(-n // block)
        for n in counts)
    assert len(table) == expected
    x = random_packed(channels, plan, axis, 81) * 2.0
    valid = np.zeros(plan.ld_words, dtype=bool)
    for off, count in spans(plan, axis):
        valid[off:off + count] = True
    x[:, ~valid] = float('nan')
    weights_ = [synth.weights(90 + i, (channels, channels, kernel_size),
                              (2.0 / (channels * kernel_size)) ** 0.5 * 1.7)
                for i in range(layers)]
    biases = [synth.weights(100 + i, (channels,), 0.1) for i in range(layers)]
    out_w = synth.weights(110, (1, channels, kernel_size), 0.2)
    out_b = synth.weights(111, (1,), 0.2)
    packs = torch.from_numpy(np.concatenate(
        [runtime.word_decoder_pack(w) for w in weights_])).to(DEVICE) \
        if layers else None
    bias_dev = torch.from_numpy(np.concatenate(biases)).to(DEVICE) \
        if layers else None
    logits = torch.full((plan.ld_words,), 9.0, device=DEVICE)
    scores = torch.full((plan.ld_words,), 9.0, device=DEVICE)
    tiles, size = meta.view(('tiles',) + request)
    x_dev = x.to(DEVICE)
    out_w_dev = torch.from_numpy(out_w).to(DEVICE)
    out_b_dev = torch.from_numpy(out_b).to(DEVICE)
    runtime.check(lib.emph_word_decoder(
        x_dev.data_ptr(), plan.ld_words, tiles.data_ptr(), size // 4,
        channels, None if packs is None else packs.data_ptr(),
        None if bias_dev is None else bias_dev.data_ptr(), layers,
        kernel_size, runtime.ACTIVATIONS[activation], out_w_dev.data_ptr(),
        out_b_dev.data_ptr(), kernel_size, runtime.POSTPROCESS[post],
        logits.data_ptr(), scores.data_ptr(), None), 'emph_word_decoder')
    logits, scores = logits.cpu(), scores.cpu()
    pad = (kernel_size - 1) // 2
    for off, count in spans(plan, axis):
        h = x[None, :, off:off + count]
        for w, b in zip(weights_, biases):
            h = ACTIVATIONS[activation](torch.nn.functional.conv1d(
                h, torch.from_numpy(w), torch.from_numpy(b), padding=pad))
        want = torch.nn.functional.conv1d(
            h, torch.from_numpy(out_w), torch.from_numpy(out_b),
            padding=pad)[0, 0]
        got = logits[off:off + count]
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) < 2e-5 * scale
        if post == 'bce':
            assert float((scores[off:off + count] -
                          torch.sigmoid(want)).abs().max()) < 1e-5
        elif post == 'mse':
            assert float((scores[off:off + count] -
                          want.clamp(0., 1.)).abs().max()) < 2e-5 * scale
    assert float(logits[~torch.from_numpy(valid)].min()) == 9.0


###############################################################################
# evaluation metrics
###############################################################################


@pytest.mark.parametrize('loss', ['bce', 'mse'])
def test_word_metrics(loss):
    """emphases_amd.metrics (emph_word_metrics) against the CPU restatement
    of emphases/evaluate/metrics.py over several ragged batches."""
    from emphases_amd import metrics as metrics_module
    from oracle import metrics as oracle_metrics
    generator = torch.Generator().manual_seed(5)
    batches = []
    for lengths in ([31, 7, 1, 19], [64, 2], [5]):
        lengths = torch.tensor(lengths)
        width = int(lengths.max())
        logits = torch.randn(len(lengths), 1, width, generator=generator) * 2
        targets = torch.rand(len(lengths), 1, width, generator=generator)
        batches.append((logits, targets, lengths))
    # dataset statistics first (evaluate/core.py:31-46), then the metrics
    predicted = metrics_module.Statistics(0)
    target = metrics_module.Statistics(0)
    flat_p, flat_t = [], []
    for logits, targets, lengths in batches:
        scores = oracle_metrics.postprocess(logits, loss)
        predicted.update(scores, lengths)
        target.update(targets, lengths)
        mask = oracle_metrics.mask_from_lengths(lengths)
        flat_p += scores[mask].tolist()
        flat_t += targets[mask].tolist()
    for got, want in ((predicted(), oracle_metrics.mean_std(flat_p)),
                      (target(), oracle_metrics.mean_std(flat_t))):
        assert abs(got[0] - want[0]) < 1e-6 and abs(got[1] - want[1]) < 1e-6
    mine = metrics_module.Metrics(predicted, target, gpu=0, loss=loss)
    theirs = oracle_metrics.Metrics(predicted(), target(), loss)
    for logits, targets, lengths in batches:
        mine.update(logits, targets, lengths)
        theirs.update(logits, targets, lengths)
    got, want = mine(), theirs()
    assert set(got) == {'pearson_correlation', 'bce', 'mse'}
    for key in want:
        assert abs(got[key] - want[key]) < 2e-6 * max(1., abs(want[key])), key
    mine.reset()
    assert math.isnan(mine()['bce'])
    # the engine's own layout: packed rows + word_segment, no padding round trip
    plan = ragged_plan([40, 9, 77], [np.stack([np.arange(n), np.arange(n) + 1])
                                     for n in (12, 1, 30)])
    logits = random_packed(1, plan, runtime.AXIS_WORDS, 91)[0] * 3.0
    targets = torch.rand(plan.ld_words, generator=generator)
    mine.update_packed(
        logits.to(DEVICE), targets.to(DEVICE),
        torch.from_numpy(plan.word_segment).to(DEVICE))
    theirs.reset()
    columns = torch.from_numpy(plan.word_columns())
    theirs.update(logits[columns][None, None], targets[columns][None, None],
                  torch.tensor([len(columns)]))
    got, want = mine(), theirs()
    for key in want:
        assert abs(got[key] - want[key]) < 2e-6 * max(1., abs(want[key])), key


@pytest.mark.parametrize('rate', [8000, 22050, 44100, 48000])
def test_resample(rate):
    """emph_resample against oracle/resample.py (the independent per-sample
    float64 restatement of torchaudio's Resample, itself anchored on closed-form
    answers in tests/test_oracle.py): float32 and 16-bit PCM input, a ragged
    batch of utterances incl. one of 5 and one of 1 sample, and a 1 kHz tone
    whose amplitude and phase must come back."""
    import emphases_amd
    from emphases_amd import load
    from oracle import resample as oracle_resample
    lib = runtime.library()
    kernel, orig, new, width = load.resample_kernel(rate)
    assert (orig, new) == oracle_resample.reduced(rate, 16000)
    lengths = [rate // 3 + 17, 5, rate // 2, 1]
    audios = [torch.from_numpy(synth.weights(200 + i, (n,), 0.9))
              for i, n in enumerate(lengths)]
    audios[2] = torch.from_numpy(np.sin(
        2 * np.pi * 1000. * np.arange(lengths[2]) / rate).astype(np.float32))
    targets = [oracle_resample.output_length(n, rate, 16000) for n in lengths]
    assert targets == [load.resampled_length(n, orig, new) for n in lengths]
    table = np.stack([np.cumsum([0] + lengths)[:-1], lengths,
                      np.cumsum([0] + targets)[:-1], targets], axis=1)
    table_dev = torch.from_numpy(table.astype(np.int64)).to(DEVICE)
    kernel_dev = kernel.reshape(new, -1).contiguous().to(DEVICE)
    for pcm in (False, True):
        if pcm:
            host = [torch.from_numpy(np.rint(a.numpy() * 32767.).astype(np.int16))
                    for a in audios]
            source = torch.cat(host).to(DEVICE)
            reference = [h.to(torch.float32) / 32768. for h in host]
        else:
            source = torch.cat(audios).to(DEVICE)
            reference = audios
        out = torch.full((sum(targets) + 8,), 7.0, device=DEVICE)
        runtime.check(lib.emph_resample(
            source.data_ptr(), int(pcm), table_dev.data_ptr(), len(lengths),
            max(targets), kernel_dev.data_ptr(), orig, new, width,
            out.data_ptr(), None), 'emph_resample')
        out = out.cpu()
        assert float(out[sum(targets):].min()) == 7.0
        cursor = 0
        for audio, count in zip(reference, targets):
            want = oracle_resample.resample(audio.numpy(), rate)
            assert want.shape == (count,)
            got = out[cursor:cursor + count].numpy().astype(np.float64)
            assert np.abs(got - want).max() < 2e-6
            cursor += count
        # closed form: the tone keeps its phase and (to the filter's 4e-4
        # passband ripple) its amplitude
        first = sum(targets[:2])
        tone = out[first:first + targets[2]].numpy()[2000:6000]
        want = np.sin(2 * np.pi * 1000. * np.arange(2000, 6000) / 16000.)
        assert np.abs(tone - want * (32767. / 32768. if pcm else 1.)).max() < 6e-4
    # through the public API: resampled on the device == the oracle's
    # resampling first, then the 16 kHz path
    frames = 240
    words = emphases_amd.Alignment.from_frames(synth.word_frames(9, frames))
    native = torch.from_numpy(synth.weights(77, (1, frames * rate // 100), 0.3))
    on_device = emphases_amd.from_alignment_and_audio(words, native, rate)
    resampled = torch.from_numpy(oracle_resample.resample(
        native[0].numpy(), rate).astype(np.float32))[None]
    on_host = emphases_amd.from_alignment_and_audio(words, resampled, 16000)
    assert on_device.shape == on_host.shape == (1, len(words))
    assert float((on_device - on_host).abs().max()) < 1e-5


def test_resample_many_utterances():
    """More utterances than a grid's y dimension holds (65 535)."""
    from emphases_amd import load
    from oracle import resample as oracle_resample
    lib = runtime.library()
    rate, count, length = 8000, 70000, 6
    kernel, orig, new, width = load.resample_kernel(rate)
    audio = torch.from_numpy(synth.weights(31, (count * length,), 0.9))
    starts = np.arange(count) * length
    table = np.stack([starts, np.full(count, length), starts * 2,
                      np.full(count, length * 2)], axis=1).astype(np.int64)
    out = torch.zeros(count * length * 2, device=DEVICE)
    audio_dev = audio.to(DEVICE)
    table_dev = torch.from_numpy(table).to(DEVICE)
    kernel_dev = kernel.reshape(new, -1).contiguous().to(DEVICE)
    runtime.check(lib.emph_resample(
        audio_dev.data_ptr(), 0, table_dev.data_ptr(), count, length * 2,
        kernel_dev.data_ptr(), orig, new, width, out.data_ptr(), None),
        'emph_resample')
    out = out.cpu().numpy().reshape(count, length * 2)
    for index in (0, 1, 65534, 65535, 65536, count - 1):
        want = oracle_resample.resample(
            audio[index * length:(index + 1) * length].numpy(), rate)
        assert np.abs(out[index] - want).max() < 2e-6, index


@pytest.mark.parametrize('loss', ['bce', 'mse'])
def test_word_metrics_match_reference(loss):
    """emphases_amd.metrics (emph_word_metrics) against what the reference's
    own `Statistics` / `Metrics` produced (tests/golden/metrics.npz,
    evaluate/metrics.py:12-110 run by tests/golden/generate.py)."""
    import os
    from emphases_amd import metrics as metrics_module
    golden = np.load(os.path.join(
        os.path.dirname(__file__), 'golden', 'metrics.npz'))
    count = int(golden[f'{loss}/batches'])
    batches = [tuple(torch.from_numpy(golden[f'{loss}/{i}/{key}'])
                     for key in ('logits', 'targets', 'word_lengths'))
               for i in range(count)]
    predicted = metrics_module.Statistics(0)
    target = metrics_module.Statistics(0)
    for logits, targets, lengths in batches:
        scores = torch.sigmoid(logits) if loss == 'bce' else logits.clamp(0., 1.)
        predicted.update(scores, lengths)
        target.update(targets, lengths)
    np.testing.assert_allclose(
        predicted(), golden[f'{loss}/predicted_stats'], rtol=1e-6)
    np.testing.assert_allclose(
        target(), golden[f'{loss}/target_stats'], rtol=1e-6)
    # (the reference hands the Statistics objects over; so do we)
    total = metrics_module.Metrics(predicted, target, gpu=0, loss=loss)
    single = metrics_module.Metrics(predicted, target, gpu=0, loss=loss)
    for index, (logits, targets, lengths) in enumerate(batches):
        single.reset()
        single.update(logits, targets, lengths)
        total.update(logits, targets, lengths)
        got = single()
        np.testing.assert_allclose(
            [got['pearson_correlation'], got['bce'], got['mse']],
            golden[f'{loss}/{index}/result'], rtol=3e-6, atol=2e-7)
    got = total()
    np.testing.assert_allclose(
        [got['pearson_correlation'], got['bce'], got['mse']],
        golden[f'{loss}/result'], rtol=3e-6, atol=2e-7)


###############################################################################
# torch.library operator seams: torch.ops.emphases_amd.*
###############################################################################


def _cu(counts):
    return torch.tensor(np.concatenate([[0], np.cumsum(counts)]),
                        dtype=torch.int32)


def test_torch_ops_logmel_conv_and_reduce():
    """The varlen ops on back-to-back segments (no alignment, no padding
    between them) against torch CPU fp32 / the oracle, one segment at a time."""
    import emphases_amd  # noqa: F401  (registers the ops)
    from oracle import prominence as oracle
    ops = torch.ops.emphases_amd
    # logmel: chunks of 1 s, 0.37 s and 2.5 s back to back
    samples = [16000, 5920, 40000]
    audios = [torch.from_numpy(synth.audio(7 + i, n // 160))[0, :n]
              for i, n in enumerate(samples)]
    mel = ops.logmel(torch.cat(audios).to(DEVICE), _cu(samples))
    frames = [n // 160 for n in samples]
    assert mel.shape == (80, sum(frames)) and mel.is_cuda
    start = 0
    for audio, count in zip(audios, frames):
        want = oracle.logmel(audio[None])
        got = mel[:, start:start + count].cpu()
        assert got.shape == want.shape
        assert float((got - want).abs().max()) < 5e-4
        start += count
    # conv1d_same_act: the three kernel families
    counts = [130, 1, 64, 257, 1000, 3]
    x = torch.from_numpy(synth.weights(3, (80, sum(counts)), 1.0))
    for c_in, c_out, k, act in ((80, 80, 3, 'relu'), (80, 80, 3, 'gelu'),
                                (80, 64, 5, 'none'), (80, 1, 3, 'none')):
        weight = torch.from_numpy(synth.weights(4 + k, (c_out, c_in, k), 0.1))
        bias = torch.from_numpy(synth.weights(5, (c_out,), 0.3))
        y = ops.conv1d_same_act(x[:c_in].to(DEVICE), weight.to(DEVICE),
                                bias.to(DEVICE), _cu(counts), act)
        assert y.shape == (c_out, sum(counts))
        start = 0
        for count in counts:
            want = torch.nn.functional.conv1d(
                x[None, :c_in, start:start + count], weight, bias,
                padding=(k - 1) // 2)[0]
            if act == 'relu':
                want = torch.relu(want)
            elif act == 'gelu':
                want = torch.nn.functional.gelu(want)
            delta = float((y[:, start:start + count].cpu() - want).abs().max())
            assert delta < 5e-5 * max(1., float(want.abs().max())), (k, act, count)
            start += count
    with pytest.raises(ValueError, match='Activation'):
        ops.conv1d_same_act(x.to(DEVICE), weight.to(DEVICE), bias.to(DEVICE),
                            _cu(counts), 'tanh')
    # segment_reduce
    words = [synth.word_frames(i, n, 1, 25) if n else np.zeros((2, 0), np.int64)
             for i, n in enumerate(counts)]
    bounds = torch.from_numpy(np.concatenate(words, axis=1))
    cu_words = _cu([w.shape[1] for w in words])
    for mode in ('sum', 'average', 'max', 'center'):
        got = ops.segment_reduce(x.to(DEVICE), bounds, _cu(counts), cu_words, mode)
        start = first = 0
        for count, own in zip(counts, words):
            want = oracle.downsample(
                x[:, start:start + count], torch.from_numpy(own), mode)
            assert float((got[:, first:first + own.shape[1]].cpu() -
                          want).abs().max()) < 2e-5, mode
            start, first = start + count, first + own.shape[1]
    with pytest.raises(ValueError, match='Interpolation'):
        ops.segment_reduce(x.to(DEVICE), bounds, _cu(counts), cu_words, 'median')


def test_torch_ops_encoder_layer_and_forward(cases):
    import emphases_amd
    from conftest import case_inputs
    ops = torch.ops.emphases_amd
    # encoder_layer against nn.TransformerEncoderLayer, a segment at a time
    channels, heads = 80, 2
    torch.manual_seed(5)
    layer = torch.nn.TransformerEncoderLayer(channels, heads, channels, 0.1).eval()
    counts = [300, 17, 1, 140]
    x = torch.from_numpy(synth.weights(9, (channels, sum(counts)), 1.0))
    parameters = [p.detach().to(DEVICE) for p in (
        layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias,
        layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias,
        layer.norm1.weight, layer.norm1.bias, layer.linear1.weight,
        layer.linear1.bias, layer.linear2.weight, layer.linear2.bias,
        layer.norm2.weight, layer.norm2.bias)]
    got = ops.encoder_layer(x.to(DEVICE), *parameters, _cu(counts), heads)
    assert got.shape == x.shape
    start = 0
    with torch.no_grad():
        for count in counts:
            want = layer(x[:, start:start + count].T[:, None])[:, 0].T
            assert float((got[:, start:start + count].cpu() -
                          want).abs().max()) < 2e-5, count
            start += count
    # prominence_forward: the chunks `preprocess` hands to `infer`, two
    # utterances back to back; bitwise what the public API returns
    chunks, all_bounds, words_per_chunk, want = [], [], [], []
    for name in ('utt_2p5s', 'utt_10s'):
        audio, bounds, _ = case_inputs(cases, name)
        alignment = emphases_amd.Alignment.from_frames(bounds)
        segments = batch.chunk_utterance(alignment, audio.shape[1])
        assert len(segments) == 1
        padded = np.pad(audio[0], (432, 432))
        segment = segments[0]
        chunks.append(torch.from_numpy(
            padded[segment.start_sample:segment.start_sample + segment.length]))
        all_bounds.append(segment.bounds)
        words_per_chunk.append(segment.bounds.shape[1])
        want.append(emphases_amd.from_alignment_and_audio(
            alignment, torch.from_numpy(audio), 16000))
    scores = ops.prominence_forward(
        torch.cat(chunks).to(DEVICE), _cu([len(c) for c in chunks]),
        torch.from_numpy(np.concatenate(all_bounds, axis=1)),
        _cu(words_per_chunk))
    assert torch.equal(scores.cpu(), torch.cat([w[0] for w in want]))
    with pytest.raises(runtime.LibraryError, match='HIP device only'):
        ops.logmel(torch.zeros(16000), _cu([16000]))
