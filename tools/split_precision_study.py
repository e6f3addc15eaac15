"""What would a bf16-split matrix path cost in accuracy?  (round-4 review, next #3)

CPU emulation, before any kernel: the attention products of the oracle's Transformer
stack (S = Q K^T and O = P V, emphases/model/layers/transformer.py:18-30) with every
operand split into n bf16 pieces and the kept cross terms accumulated in float32 - what
v_mfma_f32_32x32x16_bf16 computes (bf16 x bf16 products are exact in fp32).
    terms 3: hi*hi + hi*lo + lo*hi              (two pieces; drops 2^-16)
    terms 6: three pieces, every product of pieces i + j <= 2  (drops 2^-24)
Compared with the plain float32 oracle and a float64 run of the same stack on
BASELINE configs[2] utterances (10 s, seeded weights).  Test infrastructure only.

usage: python tools/split_precision_study.py [utterances]
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emphases_amd import config as cfg, synth, weights  # noqa: E402
from oracle import prominence as oracle  # noqa: E402


def pieces(x, count):
    """x (float32) as `count` bf16-valued float32 tensors, most significant first."""
    out, rest = [], x
    for _ in range(count):
        piece = rest.to(torch.bfloat16).to(torch.float32)
        out.append(piece)
        rest = rest - piece
    return out


def split_matmul(a, b, terms):
    """a @ b with the operands split; float32 accumulation."""
    if terms == 0:
        return a @ b
    count = 2 if terms == 3 else 3
    pa, pb = pieces(a, count), pieces(b, count)
    total = None
    # least significant products first, as a kernel would order them
    pairs = [(i, j) for i in range(count) for j in range(count) if i + j < count]
    for i, j in sorted(pairs, key=lambda p: -(p[0] + p[1])):
        term = pa[i] @ pb[j]
        total = term if total is None else total + term
    return total


def transformer_stack(x, state, prefix, layers, heads, terms_s, terms_o, terms_linear=0,
                      dtype=torch.float32):
    channels, length = x.shape
    head_dim = channels // heads
    state = {k: v.to(dtype) for k, v in state.items() if k.startswith(prefix)}
    h = (x.T + oracle.positional_encoding(length, channels)).to(dtype)
    mm = (lambda a, b: a @ b) if dtype == torch.float64 else \
        (lambda a, b: split_matmul(a, b, terms_linear))
    for i in range(layers):
        p = f'{prefix}.model.layers.{i}.'
        qkv = mm(h, state[p + 'self_attn.in_proj_weight'].T) + state[p + 'self_attn.in_proj_bias']
        q, k, v = qkv.split(channels, dim=1)
        q = q.reshape(length, heads, head_dim).transpose(0, 1)
        k = k.reshape(length, heads, head_dim).transpose(0, 1)
        v = v.reshape(length, heads, head_dim).transpose(0, 1)
        if dtype == torch.float64:
            scores = (q / math.sqrt(head_dim)) @ k.transpose(1, 2)
            attention = torch.softmax(scores, dim=-1) @ v
        else:
            scores = split_matmul(q / math.sqrt(head_dim), k.transpose(1, 2), terms_s)
            # the kernel's form: unnormalised probabilities, one division at the end
            top = scores.max(dim=-1, keepdim=True).values
            prob = torch.exp(scores - top)
            attention = split_matmul(prob, v, terms_o) / prob.sum(-1, keepdim=True)
        attention = attention.transpose(0, 1).reshape(length, channels)
        attention = mm(attention, state[p + 'self_attn.out_proj.weight'].T) + \
            state[p + 'self_attn.out_proj.bias']
        h = torch.nn.functional.layer_norm(
            h + attention, (channels,), state[p + 'norm1.weight'], state[p + 'norm1.bias'], 1e-5)
        ff = torch.relu(mm(h, state[p + 'linear1.weight'].T) + state[p + 'linear1.bias'])
        ff = mm(ff, state[p + 'linear2.weight'].T) + state[p + 'linear2.bias']
        h = torch.nn.functional.layer_norm(
            h + ff, (channels,), state[p + 'norm2.weight'], state[p + 'norm2.bias'], 1e-5)
    return h.T


def forward(feats, bounds, state, variant):
    """model/core.py:89-138, Transformer encoder and decoder, 'intermediate' / 'sum'."""
    dtype = torch.float64 if variant == 'f64' else torch.float32
    terms = {'f32': (0, 0, 0), 'f64': (0, 0, 0), 'x3': (3, 3, 0), 'x6': (6, 6, 0),
             's6o3': (6, 3, 0), 'x3_all': (3, 3, 3), 'x6_all': (6, 6, 6)}[variant]
    st = {k: v.to(dtype) for k, v in state.items()}
    x = torch.nn.functional.conv1d(
        feats[None].to(dtype), st['input_layer.weight'], st['input_layer.bias'], padding='same')[0]
    x = transformer_stack(x, state, 'frame_encoder', 6, 2, *terms, dtype=dtype)
    x = oracle.downsample(x, bounds, 'sum')
    x = transformer_stack(x, state, 'word_decoder', 6, 2, *terms, dtype=dtype)
    return torch.nn.functional.conv1d(
        x[None], st['output_layer.weight'], st['output_layer.bias'], padding='same')[0, 0]


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    config = cfg.Config(architecture='transformer')
    state = {k: torch.from_numpy(v) for k, v in weights.random_state(config, seed=0).items()}
    variants = ('f32', 'x3', 's6o3', 'x6', 'x3_all', 'x6_all')
    worst = {v: [0., 0.] for v in variants}
    torch.set_num_threads(8)
    for index in range(count):
        audio = torch.from_numpy(synth.audio(index, 1000))
        bounds = torch.from_numpy(synth.word_frames(index, 1000).astype(np.int64))
        feats = oracle.features(audio, {'architecture': 'transformer'})[0] \
            if isinstance(oracle.features(audio, {}), tuple) else oracle.features(audio, {})
        feats = feats.reshape(80, -1)
        with torch.no_grad():
            exact = forward(feats, bounds, state, 'f64')
            for v in variants:
                logits = forward(feats, bounds, state, v).double()
                dl = float((logits - exact).abs().max())
                ds = float((torch.sigmoid(logits) - torch.sigmoid(exact)).abs().max())
                worst[v][0], worst[v][1] = max(worst[v][0], dl), max(worst[v][1], ds)
        print(f'utterance {index}: |logit| up to {float(exact.abs().max()):.2f}; ' + '  '.join(
            f'{v}: dlogit {worst[v][0]:.2e} dscore {worst[v][1]:.2e}' for v in variants), flush=True)
    print('worst against float64 over', count, 'utterances (logit, score):')
    for v in variants:
        print(f'  {v:8s} {worst[v][0]:.3e}  {worst[v][1]:.3e}')


if __name__ == '__main__':
    main()
