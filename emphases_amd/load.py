"""Audio file loading and sample-rate conversion (the I/O boundary of
`emphases/load.py:11-17` and `emphases/core.py:613-619`).

The reference delegates both to `torchaudio` (third-party, absent here), so
this module carries its own RIFF/WAVE reader (PCM 8/16/24/32-bit and IEEE
float) and a restatement of torchaudio's default `Resample` — windowed-sinc
polyphase interpolation, Hann window, `lowpass_filter_width=6`,
`rolloff=0.99` — from its published algorithm (parity-unpinned).  The reader
runs on the host (file plumbing); of the resampler only the TABLE is built
here (`resample_kernel`, float64 on the host like every weight pack): the
arithmetic is `emph_resample` on the device, for every caller.
"""
import math
import struct

import numpy as np
import torch

from . import config as cfg


# (format code, bits per sample) the reader accepts
FORMATS = {(1, 8), (1, 16), (1, 24), (1, 32), (3, 32), (3, 64)}


def _walk(handle, file):
    """ONE chunk walker for `wav` and `wav_info` (seeks, no sample is read):
    `(code, channels, rate, bits, data offset, data bytes)`.  The LAST `fmt `
    and the LAST `data` chunk count, in any order; a data chunk that claims
    more bytes than the file holds is cut to what is there; odd-sized chunks
    are padded to even offsets."""
    head = handle.read(12)
    if len(head) < 12 or head[:4] != b'RIFF' or head[8:12] != b'WAVE':
        raise ValueError(f'{file} is not a RIFF/WAVE file')
    handle.seek(0, 2)
    end = handle.tell()
    cursor = 12
    fmt = data = None
    while cursor + 8 <= end:
        handle.seek(cursor)
        header = handle.read(8)
        tag, size = header[:4], struct.unpack('<I', header[4:])[0]
        if tag == b'fmt ':
            body = handle.read(min(size, 40))
            if len(body) < 16:
                raise ValueError(f'{file}: fmt chunk of {len(body)} bytes')
            code, channels, rate, _, _, bits = struct.unpack(
                '<HHIIHH', body[:16])
            if code == 0xFFFE and len(body) >= 26:      # WAVE_FORMAT_EXTENSIBLE
                code = struct.unpack('<H', body[24:26])[0]
            fmt = (code, channels, rate, bits)
        elif tag == b'data':
            data = (cursor + 8, min(size, end - cursor - 8))
        cursor += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f'{file} has no fmt/data chunk')
    code, channels, rate, bits = fmt
    if (code, bits) not in FORMATS or channels < 1:
        raise ValueError(
            f'{file}: unsupported WAVE format {code}/{bits} '
            f'({channels} channels)')
    return code, channels, rate, bits, data[0], data[1]


def wav(file, raw=False):
    """Read a RIFF/WAVE file -> (float32 tensor [channels, samples], rate).
    `raw`: 16-bit PCM comes back as an int16 tensor (x / 32768 is what the
    float form holds; the HIP front-end applies it on the device)."""
    with open(file, 'rb') as handle:
        code, channels, rate, bits, offset, nbytes = _walk(handle, file)
        frame = bits // 8 * channels
        nbytes = nbytes // frame * frame
        handle.seek(offset)
        samples = np.fromfile(handle, dtype=np.uint8, count=nbytes)
    if code == 1 and bits == 8:
        values = (samples.astype(np.float32) - 128.) / 128.
    elif code == 1 and bits == 16 and raw:
        values = samples.view('<i2').astype(np.int16, copy=False)
    elif code == 1 and bits == 16:
        values = samples.view('<i2').astype(np.float32) / 32768.
    elif code == 1 and bits == 24:
        triples = samples.reshape(-1, 3).astype(np.int32)
        values = triples[:, 0] | (triples[:, 1] << 8) | (triples[:, 2] << 16)
        values = np.where(values >= 1 << 23, values - (1 << 24), values)
        values = values.astype(np.float32) / float(1 << 23)
    elif code == 1 and bits == 32:
        values = samples.view('<i4').astype(np.float32) / float(1 << 31)
    elif code == 3 and bits == 32:
        values = samples.view('<f4').astype(np.float32, copy=False)
    else:
        values = samples.view('<f8').astype(np.float32)
    return torch.from_numpy(
        np.ascontiguousarray(values.reshape(-1, channels).T)), rate


def wav_info(file):
    """(sample rate, channels, samples per channel) from the headers of a
    RIFF/WAVE file alone: the chunk list is walked with seeks, the samples are
    not read (what a sharded run plans from, `dist.from_files_to_files`).
    Same walker, same validation as `wav`: the two cannot disagree."""
    with open(file, 'rb') as handle:
        _, channels, rate, bits, _, nbytes = _walk(handle, file)
    return rate, channels, nbytes // (bits // 8 * channels)


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
    bytes); anything else as float32 like the reference.

    A file that is not at 16 kHz NEEDS THE GPU, unlike the reference's
    `load.audio`: the package has exactly one resampler, `emph_resample`, so
    that every entry point agrees bit for bit at any input rate, and no CPU
    fallback (on a host without an MI355X this raises `runtime.LibraryError`;
    `wav(file)` reads any rate on the host, unresampled)."""
    samples, rate = wav(file, raw)
    if rate == cfg.SAMPLE_RATE:
        return samples
    from . import core
    return core.resample(samples, rate)       # on the device: emph_resample


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
