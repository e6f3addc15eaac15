"""ORACLE — CPU restatement of the reference's prominence-inference path.

TEST INFRASTRUCTURE ONLY.  Imported by `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py`; never by the product (`emphases_amd`),
which must fail loudly when its HIP library is missing.

What it restates (paths relative to /root/reference):
  chunks()            emphases/core.py:345-418 (`preprocess`), convert.py:9-36
  logmel()            emphases/data/preprocess/mels.py:16-59,94-109
  loudness()          emphases/data/preprocess/loudness.py:59-120
  downsample()        emphases/core.py:426-469
  segment()           emphases/core.py:552-586
  conv_stack()        emphases/model/layers/convolution.py:13-37
  transformer_stack() emphases/model/layers/transformer.py:13-52
  forward()           emphases/model/core.py:39-138
  postprocess()       emphases/core.py:335-342
  from_alignment_and_audio()  emphases/core.py:223-265

Precision: float32 on CPU with the same ATen ops the reference calls
(`torch.stft`, `conv1d`, `matmul`), autocast OFF — the reference's shipped
bf16/fp16 autocast (`core.py:594-607`) cannot be matched to 1e-4 by anything,
so parity is defined against its own modules in fp32.

Pinning: first-party behaviour is pinned by `tests/golden/*.npz`, captured by
importing the unmodified reference (`tests/golden/generate.py`).  Third-party
arithmetic (librosa mel basis / A-weighting, pypar bounds, penn) is PARITY
UNPINNED: those packages are not under /root/reference nor installed here.
"""
import math

import numpy as np
import torch

from . import librosa_mel

SAMPLE_RATE = 16000
HOPSIZE = 160
NUM_FFT = 1024
WINDOW_SIZE = 1024
NUM_MELS = 80


###############################################################################
# Host arithmetic (convert.py)
###############################################################################


def seconds_to_frames(seconds):
    """convert.py:19-31: `(seconds * 16000) // 160` in float64."""
    return (seconds * SAMPLE_RATE) // HOPSIZE


###############################################################################
# Chunking (core.py:345-418)
###############################################################################


def chunks(words, num_samples, batch_size=None):
    """Word-boundary chunk plan.

    words: list of (start_seconds, end_seconds) python floats.
    Returns a list of dicts with `start_word, end_word, start_sample,
    end_sample` (indices into the 432-zero-padded signal) and `bounds`
    (int64 [2, Wc], chunk-relative frames).  Chunks whose audio is too short
    for the reflect pad (core.py:403-415 swallows the RuntimeError) carry
    `dropped=True`."""
    padding = int((WINDOW_SIZE - HOPSIZE) / 2)                 # core.py:357
    padded = num_samples + 2 * padding
    total_frames = int(padded / HOPSIZE)                       # core.py:359
    batch_size = total_frames if batch_size is None else batch_size
    plan = []
    start = 0
    while start < len(words):                                  # core.py:365
        frames = 0.
        end = start + 1
        while end < len(words):
            duration = words[end - 1][1] - words[end - 1][0]
            frames += seconds_to_frames(duration)              # core.py:373
            if int(frames) > batch_size:                       # core.py:377
                break
            end += 1
        origin = int(words[start][0] * SAMPLE_RATE / HOPSIZE)
        bounds = np.array([
            [int(s * SAMPLE_RATE / HOPSIZE) - origin for s, _ in
             words[start:end]],
            [int(e * SAMPLE_RATE / HOPSIZE) - origin for _, e in
             words[start:end]]], dtype=np.int64)
        start_sample = int(int(seconds_to_frames(words[start][0])) * HOPSIZE)
        end_sample = int(int(seconds_to_frames(words[end - 1][1])) * HOPSIZE)
        start_sample = min(start_sample, padded)
        end_sample = min(end_sample, padded)                   # python slicing
        length = max(0, end_sample - start_sample)
        plan.append(dict(
            start_word=start, end_word=end, start_sample=start_sample,
            end_sample=end_sample, bounds=bounds,
            dropped=length <= padding))     # reflect pad needs len > 432
        start = end
    return plan


###############################################################################
# Features
###############################################################################


_BASIS = None


def mel_basis():
    global _BASIS
    if _BASIS is None:
        _BASIS = torch.from_numpy(librosa_mel.mel(
            sr=SAMPLE_RATE, n_fft=NUM_FFT, n_mels=NUM_MELS))
    return _BASIS


def logmel(audio, normalize=False):
    """mels.py:16-59: audio float32 [1, S] -> [80, F]."""
    size = (NUM_FFT - HOPSIZE) // 2
    audio = torch.nn.functional.pad(
        audio[None], (size, size), mode='reflect')[0]          # mels.py:31-36
    window = torch.hann_window(WINDOW_SIZE, dtype=audio.dtype)
    stft = torch.stft(
        audio, NUM_FFT, hop_length=HOPSIZE, window=window, center=False,
        normalized=False, onesided=True, return_complex=True)
    stft = torch.view_as_real(stft)[0]                         # mels.py:48
    spectrogram = torch.sqrt(stft.pow(2).sum(-1) + 1e-6)       # mels.py:51
    mels = torch.log(torch.clamp(
        torch.matmul(mel_basis(), spectrogram), min=1e-5))     # mels.py:106-109
    if normalize:
        return (mels + 10.) / 10.                              # mels.py:57-58
    return mels


def a_weights():
    """loudness.py:110-120 — A-weighting evaluated on penn's 8 kHz / 1024 grid
    (a quirk of the reference), minus REF_DB=20."""
    frequencies = np.fft.rfftfreq(n=1024, d=1.0 / 8000)
    f_sq = frequencies ** 2.0
    const = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2.0
    with np.errstate(divide='ignore'):
        weights = 2.0 + 20.0 * (
            np.log10(const[0]) + 2 * np.log10(f_sq)
            - np.log10(f_sq + const[0]) - np.log10(f_sq + const[1])
            - 0.5 * np.log10(f_sq + const[2])
            - 0.5 * np.log10(f_sq + const[3]))
    return np.maximum(-80.0, weights)[:, None] - 20.


def loudness(audio, normalize=False):
    """loudness.py:59-107: audio float32 [1, S] -> [1, F] (numpy on CPU)."""
    p = (NUM_FFT - HOPSIZE) // 2
    padded = torch.nn.functional.pad(
        audio[:, None], (p, p), 'reflect').squeeze(1)[0].numpy()
    n = np.arange(WINDOW_SIZE)
    window = (0.5 - 0.5 * np.cos(
        2.0 * np.pi * n / WINDOW_SIZE)).astype(np.float32)
    frames = 1 + (len(padded) - WINDOW_SIZE) // HOPSIZE
    index = np.arange(WINDOW_SIZE)[:, None] + HOPSIZE * np.arange(frames)[None]
    stft = np.fft.rfft(
        window[:, None] * padded[index], axis=0).astype(np.complex64)
    magnitude = np.abs(stft)
    db = 10.0 * np.log10(np.maximum(1e-10, magnitude * magnitude))
    db = np.maximum(db, db.max() - 80.0)       # top_db over the whole chunk
    weighted = db + a_weights()
    weighted[weighted < -100.] = -100.                     # loudness.py:97
    result = torch.from_numpy(weighted.mean(axis=0)).float()[None]
    if normalize:
        return (result + 100.) / 100.
    return result


def features(audio, cfg, pitch_tracker=None):
    """data/preprocess/core.py:71-125 -> [1, NUM_FEATURES, F].  Pitch and
    periodicity come from `penn` (a neural tracker, not restatable):
    `pitch_tracker(audio [1, S]) -> (pitch [1, F] Hz, periodicity [1, F])`
    stands in for `penn.from_audio` (core.py:84-92); what the reference does
    with its outputs (core.py:94-106) is restated here."""
    rows = []
    if cfg.get('mel_feature', True):
        rows.append(logmel(audio, cfg.get('normalize', False)))
    if cfg.get('pitch_feature') or cfg.get('periodicity_feature'):
        if pitch_tracker is None:
            raise NotImplementedError('penn pitch is third-party')
        pitch, periodicity = pitch_tracker(audio)
        if cfg.get('pitch_feature'):
            if cfg.get('normalize', False):
                logfmin = torch.log2(torch.tensor(40.))     # static.py:33-36
                logfmax = torch.log2(torch.tensor(550.))
                rows.append(
                    (torch.log2(pitch) - logfmin) / (logfmax - logfmin))
            else:
                rows.append(torch.log2(pitch))
        if cfg.get('periodicity_feature'):
            rows.append(periodicity)
    if cfg.get('loudness_feature'):
        rows.append(loudness(audio, cfg.get('normalize', False)))
    return (rows[0] if len(rows) == 1 else torch.cat(rows))[None]


###############################################################################
# Word/frame resampling
###############################################################################


def downsample(x, bounds, method='sum'):
    """core.py:426-469 for one utterance: x [C, T], bounds [2, W] -> [C, W]."""
    channels = x.shape[0]
    count = bounds.shape[1]
    if method == 'center':
        index = (bounds[0] + bounds[1]) // 2                   # core.py:462
        return x[:, torch.as_tensor(index, dtype=torch.long)]
    result = torch.zeros((channels, count), dtype=x.dtype)
    for j in range(count):
        start, end = int(bounds[0, j]), int(bounds[1, j])
        piece = x[:, start:end]
        if method == 'average':
            result[:, j] = piece.mean(dim=1)
        elif method == 'max':
            result[:, j] = piece.max(dim=1).values
        elif method == 'sum':
            result[:, j] = piece.sum(dim=1)
        else:
            raise ValueError(
                f'Interpolation method {method} is not defined')
    return result


###############################################################################
# Layers
###############################################################################


def activation(x, name):
    if name == 'relu':
        return torch.relu(x)
    if name == 'gelu':
        return torch.nn.functional.gelu(x)
    if name == 'silu':
        return torch.nn.functional.silu(x)
    if name == 'leaky_relu':
        return torch.nn.functional.leaky_relu(x, 0.01)
    raise ValueError(name)


def conv(x, weight, bias):
    """Conv1d(padding='same'), x [C, T] one sequence (model/core.py:17-21)."""
    return torch.nn.functional.conv1d(
        x[None], weight, bias, padding=(weight.shape[-1] - 1) // 2)[0]


def conv_stack(x, state, prefix, layers, act):
    """convolution.py:25-37: layers x [Conv1d 'same', activation]."""
    for i in range(layers):
        x = activation(conv(
            x, state[f'{prefix}.{2 * i}.weight'],
            state[f'{prefix}.{2 * i}.bias']), act)
    return x


def positional_encoding(length, channels):
    """transformer.py:43-48 (max_len 5000)."""
    if length > 5000:
        raise RuntimeError('sequence exceeds the 5000-position table')
    index = torch.arange(length).unsqueeze(1)
    frequency = torch.exp(
        torch.arange(0, channels, 2) * (-math.log(10000.0) / channels))
    encoding = torch.zeros(length, channels)
    encoding[:, 0::2] = torch.sin(index * frequency)
    encoding[:, 1::2] = torch.cos(index * frequency)
    return encoding


def layer_norm(x, weight, bias, eps=1e-5):
    return torch.nn.functional.layer_norm(
        x, (x.shape[-1],), weight, bias, eps)


def transformer_stack(x, state, prefix, layers, heads=2, valid=None):
    """transformer.py:25-30 for one sequence: x [C, T] -> [C, T].
    `nn.TransformerEncoderLayer` defaults: post-LN, ReLU, eps 1e-5; dropout is
    the identity in eval mode.  `valid`: number of leading positions that are
    real; the rest are padding that `src_key_padding_mask` hides as KEYS
    (transformer.py:26-29) while they are still computed as queries — only the
    zero-padded word pieces of DOWNSAMPLE_LOCATION='input' have any."""
    channels, length = x.shape
    head_dim = channels // heads
    h = x.T + positional_encoding(length, channels)            # [T, C]
    for i in range(layers):
        p = f'{prefix}.model.layers.{i}.'
        qkv = h @ state[p + 'self_attn.in_proj_weight'].T + \
            state[p + 'self_attn.in_proj_bias']
        q, k, v = qkv.split(channels, dim=1)
        q = q.reshape(length, heads, head_dim).transpose(0, 1)
        k = k.reshape(length, heads, head_dim).transpose(0, 1)
        v = v.reshape(length, heads, head_dim).transpose(0, 1)
        scores = (q / math.sqrt(head_dim)) @ k.transpose(1, 2)
        if valid is not None and valid < length:
            scores[:, :, valid:] = float('-inf')
        attention = torch.softmax(scores, dim=-1) @ v          # [H, T, D]
        attention = attention.transpose(0, 1).reshape(length, channels)
        attention = attention @ state[p + 'self_attn.out_proj.weight'].T + \
            state[p + 'self_attn.out_proj.bias']
        h = layer_norm(
            h + attention, state[p + 'norm1.weight'], state[p + 'norm1.bias'])
        ff = torch.relu(
            h @ state[p + 'linear1.weight'].T + state[p + 'linear1.bias'])
        ff = ff @ state[p + 'linear2.weight'].T + state[p + 'linear2.bias']
        h = layer_norm(
            h + ff, state[p + 'norm2.weight'], state[p + 'norm2.bias'])
    return h.T


def stack(x, state, prefix, cfg, valid=None):
    if cfg.get('architecture', 'convolution') == 'convolution':
        return conv_stack(
            x, state, prefix, cfg.get('layers', 6),
            cfg.get('activation', 'relu'))
    return transformer_stack(
        x, state, prefix, cfg.get('layers', 6), valid=valid)


###############################################################################
# Model (model/core.py:39-138), one utterance or chunk (B=1)
###############################################################################


def forward(feats, bounds, state, cfg=None, stages=None):
    """feats [C_in, T] float32, bounds int64 [2, W] -> logits [W].

    `stages`, when a dict, receives the intermediate tensors."""
    cfg = cfg or {}
    location = cfg.get('downsample_location', 'intermediate')
    method = cfg.get('downsample_method', 'sum')
    bounds = torch.as_tensor(np.asarray(bounds), dtype=torch.long)
    stages = {} if stages is None else stages
    if location == 'input':
        # model/core.py:41-87 — every word is its own zero-padded sequence of
        # `max_length` frames (core.py:552-586), pooled over the PADDED axis
        count = bounds.shape[1]
        lengths = bounds[1] - bounds[0]
        max_length = int(lengths.max())
        words = []
        for j in range(count):
            piece = torch.zeros((feats.shape[0], max_length))
            piece[:, :int(lengths[j])] = feats[
                :, int(bounds[0, j]):int(bounds[1, j])]
            embedding = stack(
                conv(piece, state['input_layer.weight'],
                     state['input_layer.bias']),
                state, 'frame_encoder', cfg, valid=int(lengths[j]))
            if method == 'average':
                words.append(embedding.mean(dim=1))
            elif method == 'max':
                words.append(embedding.max(dim=1).values)
            elif method == 'sum':
                words.append(embedding.sum(dim=1))
            elif method == 'center':
                words.append(embedding[:, int(lengths[j]) // 2])
            else:
                raise ValueError(method)
        word_embeddings = torch.stack(words, dim=1)
        stages['downsampled'] = word_embeddings
        word_embeddings = stack(word_embeddings, state, 'word_decoder', cfg)
    else:
        hidden = conv(
            feats, state['input_layer.weight'], state['input_layer.bias'])
        stages['input_layer'] = hidden
        frame_embeddings = stack(hidden, state, 'frame_encoder', cfg)
        stages['encoder'] = frame_embeddings
        word_embeddings = downsample(frame_embeddings, bounds, method)
        stages['downsampled'] = word_embeddings
        if location == 'intermediate':
            word_embeddings = stack(
                word_embeddings, state, 'word_decoder', cfg)
        elif location not in ('loss', 'inference'):
            raise ValueError(
                f'Downsample location {location} not recognized')
    stages['decoder'] = word_embeddings
    logits = conv(
        word_embeddings, state['output_layer.weight'],
        state['output_layer.bias'])[0]
    stages['logits'] = logits
    return logits


def postprocess(logits, loss='bce'):
    """core.py:335-342"""
    if loss == 'bce':
        return torch.sigmoid(logits)
    return torch.clamp(logits, 0., 1.)


###############################################################################
# API (core.py:223-265)
###############################################################################


def from_alignment_and_audio(words, audio, state, cfg=None, batch_size=None,
                             pitch_tracker=None):
    """words: [(start_s, end_s)], audio float32 [1, S] at 16 kHz.
    Returns scores float32 [1, sum Wc]."""
    cfg = cfg or {}
    padding = int((WINDOW_SIZE - HOPSIZE) / 2)
    padded = torch.nn.functional.pad(audio[:1], (padding, padding))
    scores = []
    with torch.no_grad():
        for chunk in chunks(words, audio.shape[-1], batch_size):
            if chunk['dropped']:
                continue
            piece = padded[:, chunk['start_sample']:chunk['end_sample']]
            feats = features(piece, cfg, pitch_tracker)[0]
            logits = forward(feats, chunk['bounds'], state, cfg)
            scores.append(postprocess(logits, cfg.get('loss', 'bce'))[None])
    return torch.cat(scores, 1)
