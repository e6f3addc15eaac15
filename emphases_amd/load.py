"""Audio file loading and sample-rate conversion (the I/O boundary of
`emphases/load.py:11-17` and `emphases/core.py:613-619`).

The reference delegates both to `torchaudio` (third-party, absent here), so
this module carries its own RIFF/WAVE reader (PCM 8/16/24/32-bit and IEEE
float) and a restatement of torchaudio's default `Resample` — windowed-sinc
polyphase interpolation, Hann window, `lowpass_filter_width=6`,
`rolloff=0.99` — from its published algorithm (parity-unpinned).  Both run on
the host: they are file plumbing, not part of the accelerated path.
"""
import math
import struct

import numpy as np
import torch

from . import config as cfg


def wav(file, raw=False):
    """Read a RIFF/WAVE file -> (float32 tensor [channels, samples], rate).
    `raw`: 16-bit PCM comes back as an int16 tensor (x / 32768 is what the
    float form holds; the HIP front-end applies it on the device)."""
    with open(file, 'rb') as handle:
        data = handle.read()
    if data[:4] != b'RIFF' or data[8:12] != b'WAVE':
        raise ValueError(f'{file} is not a RIFF/WAVE file')
    cursor = 12
    fmt = None
    samples = None
    while cursor + 8 <= len(data):
        tag = data[cursor:cursor + 4]
        size = struct.unpack('<I', data[cursor + 4:cursor + 8])[0]
        body = data[cursor + 8:cursor + 8 + size]
        if tag == b'fmt ':
            code, channels, rate, _, _, bits = struct.unpack(
                '<HHIIHH', body[:16])
            if code == 0xFFFE and len(body) >= 26:      # WAVE_FORMAT_EXTENSIBLE
                code = struct.unpack('<H', body[24:26])[0]
            fmt = (code, channels, rate, bits)
        elif tag == b'data':
            samples = body
        cursor += 8 + size + (size & 1)
    if fmt is None or samples is None:
        raise ValueError(f'{file} has no fmt/data chunk')
    code, channels, rate, bits = fmt
    if code == 1 and bits == 8:
        values = (np.frombuffer(samples, dtype=np.uint8).astype(np.float32)
                  - 128.) / 128.
    elif code == 1 and bits == 16 and raw:
        values = np.frombuffer(samples, dtype='<i2').astype(np.int16)
    elif code == 1 and bits == 16:
        values = np.frombuffer(samples, dtype='<i2').astype(np.float32) / 32768.
    elif code == 1 and bits == 24:
        raw = np.frombuffer(samples[:len(samples) // 3 * 3], dtype=np.uint8)
        raw = raw.reshape(-1, 3).astype(np.int32)
        values = (raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16))
        values = np.where(values >= 1 << 23, values - (1 << 24), values)
        values = values.astype(np.float32) / float(1 << 23)
    elif code == 1 and bits == 32:
        values = np.frombuffer(samples, dtype='<i4').astype(np.float32) / \
            float(1 << 31)
    elif code == 3 and bits == 32:
        values = np.frombuffer(samples, dtype='<f4').astype(np.float32)
    elif code == 3 and bits == 64:
        values = np.frombuffer(samples, dtype='<f8').astype(np.float32)
    else:
        raise ValueError(f'{file}: unsupported WAVE format {code}/{bits}')
    values = values[:len(values) // channels * channels]
    return torch.from_numpy(
        np.ascontiguousarray(values.reshape(-1, channels).T)), rate


def wav_info(file):
    """(sample rate, channels, samples per channel) from the headers of a
    RIFF/WAVE file alone: the chunk list is walked with seeks, the samples are
    not read (what a sharded run plans from, `dist.from_files_to_files`)."""
    with open(file, 'rb') as handle:
        head = handle.read(12)
        if head[:4] != b'RIFF' or head[8:12] != b'WAVE':
            raise ValueError(f'{file} is not a RIFF/WAVE file')
        fmt = None
        while True:
            header = handle.read(8)
            if len(header) < 8:
                break
            tag, size = header[:4], struct.unpack('<I', header[4:])[0]
            if tag == b'fmt ':
                body = handle.read(size)
                _, channels, rate, _, _, bits = struct.unpack(
                    '<HHIIHH', body[:16])
                fmt = (rate, channels, bits)
                handle.seek(size & 1, 1)
            elif tag == b'data':
                if fmt is None:
                    break
                rate, channels, bits = fmt
                # (a data chunk that claims more than the file holds: `wav`
                # reads what is there)
                here = handle.tell()
                handle.seek(0, 2)
                size = min(size, handle.tell() - here)
                return rate, channels, size // (bits // 8) // channels
            else:
                handle.seek(size + (size & 1), 1)
    raise ValueError(f'{file} has no fmt/data chunk')


def save_wav(file, audio, sample_rate=cfg.SAMPLE_RATE):
    """Write float audio [channels, samples] as 16-bit PCM."""
    audio = np.asarray(audio, dtype=np.float32)
    audio = audio[None] if audio.ndim == 1 else audio
    pcm = np.clip(np.rint(audio.T * 32768.), -32768, 32767).astype('<i2')
    body = pcm.tobytes()
    channels = audio.shape[0]
    header = b'RIFF' + struct.pack('<I', 36 + len(body)) + b'WAVEfmt ' + \
        struct.pack('<IHHIIHH', 16, 1, channels, sample_rate,
                    sample_rate * channels * 2, channels * 2, 16) + \
        b'data' + struct.pack('<I', len(body))
    with open(file, 'wb') as handle:
        handle.write(header + body)


def audio(file, raw=False):
    """Load audio and maybe resample (`emphases/load.py:11-17`).  `raw`: a
    16 kHz 16-bit PCM file is returned as int16 (no conversion, half the
    bytes); anything else as float32 like the reference."""
    samples, rate = wav(file, raw)
    if samples.dtype == torch.int16:
        if rate == cfg.SAMPLE_RATE:
            return samples
        samples = samples.to(torch.float32) / 32768.
    return resample(samples, rate)


def resample_kernel(sample_rate, target_rate=cfg.SAMPLE_RATE,
                    lowpass_filter_width=6, rolloff=0.99):
    """Polyphase windowed-sinc kernel of torchaudio.transforms.Resample's
    defaults, restated from its published algorithm (third-party,
    parity-unpinned): `(kernel float32 [new, 1, 2 width + orig], orig, new,
    width)` with the two rates divided by their gcd."""
    sample_rate, target_rate = int(sample_rate), int(target_rate)
    gcd = math.gcd(sample_rate, target_rate)
    orig, new = sample_rate // gcd, target_rate // gcd
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    index = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] \
        / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + \
        index
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernel = torch.where(t == 0, torch.ones_like(t), t.sin() / t)
    kernel = (kernel * window * (base / orig)).to(torch.float32)
    return kernel, orig, new, width


def resampled_length(length, orig, new):
    return int(math.ceil(new * length / orig))


def resample(audio, sample_rate, target_rate=cfg.SAMPLE_RATE,
             lowpass_filter_width=6, rolloff=0.99):
    """Windowed-sinc resampling on the HOST (torchaudio.transforms.Resample
    defaults): file plumbing and the checker of the device version
    (`emph_resample`, which the batch API uses)."""
    sample_rate, target_rate = int(sample_rate), int(target_rate)
    if sample_rate == target_rate:
        return audio
    kernel, orig, new, width = resample_kernel(
        sample_rate, target_rate, lowpass_filter_width, rolloff)
    shape = audio.shape
    flat = audio.reshape(-1, shape[-1]).to(torch.float32).cpu()
    length = flat.shape[-1]
    flat = torch.nn.functional.pad(flat, (width, width + orig))
    result = torch.nn.functional.conv1d(flat[:, None], kernel, stride=orig)
    result = result.transpose(1, 2).reshape(flat.shape[0], -1)
    target = resampled_length(length, orig, new)
    return result[..., :target].reshape(shape[:-1] + (target,))
