"""The file boundary of `from_files_to_files` (`emphases/core.py:115-179`), a
batch at a time, on the library's pool of host threads (`csrc/files.hip`).

The reference loads one alignment (`pypar.Alignment(text_file)`, `core.py:49`)
and one audio file (`emphases.load.audio`, `load.py:11-17`) at a time on the
Python thread and saves `<prefix>.TextGrid` / `<prefix>.pt` the same way
(`core.py:111-112`).  The device path behind it takes a few microseconds per
utterance, so at corpus scale the files set the rate: in Python alone a 10 s
file costs 0.5 ms to parse, 0.5 ms to read and 0.2 ms to save.  A `FileBatch`
opens a batch of (TextGrid, WAV) pairs in one library call (parsed and
header-walked in parallel), hands the planner the word times as arrays
(`Alignment` objects that build their `Word`s only when somebody asks), lets
the session read the samples of mono 16-bit PCM / float32 files STRAIGHT into
its pinned staging buffer (`FileAudio` stands in for the tensor), and writes
the outputs of a batch in one call.

Files the fast path does not take - JSON alignments, multi-channel or 8 / 24 /
32-bit integer WAVE files - go through `alignment.py` / `load.py` one by one,
with the same results; a file that cannot be read raises what those readers
raise.
"""
import ctypes
import os

import numpy as np
import torch

from . import alignment as alignment_module
from . import load
from . import runtime


def _cpu_budget():
    """CPUs this process may keep busy: the cgroup quota when there is one
    (a container that shows 128 cores under a 16-CPU quota is THROTTLED -
    frozen for the rest of the period - as soon as its threads together use
    more than that), else the affinity mask."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as file:
            quota, period = file.read().split()
        if quota != 'max':
            return max(1, int(float(quota) / float(period)))
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


# threads of the library's file pool per call: half the budget (two batches are
# in flight, the Python thread and the HIP runtime's threads need CPUs too)
THREADS = int(os.environ.get(
    'EMPHASES_FILE_THREADS', max(2, min(16, _cpu_budget() // 2))))


def stage_threads(openers=1):
    """(threads of an open call, threads of a write call) for the file API's
    pipeline, where `openers` open calls, a write call, the calling thread and
    the HIP runtime's own threads run at once: together they stay under the
    CPU budget (a throttled cgroup loses whole scheduler periods).  Opening
    has the most to do (parse + read), so it gets the larger share."""
    if 'EMPHASES_OPEN_THREADS' in os.environ or \
            'EMPHASES_WRITE_THREADS' in os.environ:
        return (int(os.environ.get('EMPHASES_OPEN_THREADS', THREADS)),
                int(os.environ.get('EMPHASES_WRITE_THREADS', THREADS)))
    if 'EMPHASES_FILE_THREADS' in os.environ:
        return THREADS, THREADS
    spare = max(2, _cpu_budget() - 4)        # the caller, HIP's threads, the helpers
    # (measured on a 16-CPU budget, fresh alignments, tools/files_batchsize.py: 3 / 5
    # threads 55 k files/s, 8 / 4 57-64 k, 12 / 4 59-65 k.  The two openers are
    # seldom inside the library at once - each spends a third of a batch in Python -
    # so every open call gets the whole share)
    opening = max(1, min(12, spare * 2 // 3))
    writing = max(1, min(8, spare - opening))
    return opening, writing


def cpus_near(device_index):
    """CPUs of the NUMA node GPU `device_index` hangs off, among those this
    process may run on - or None when that cannot be told (one node, no sysfs,
    EMPHASES_NUMA=0).  A batch's samples are copied into pinned memory by host
    threads and read from there by the GPU's DMA engine: from the far socket
    of a two-socket host the file API loses 10 % and lands on either side from
    run to run."""
    if os.environ.get('EMPHASES_NUMA', '1') == '0':
        return None
    try:
        props = torch.cuda.get_device_properties(device_index)
        address = '%04x:%02x:%02x.0' % (
            props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        with open(f'/sys/bus/pci/devices/{address}/numa_node') as file:
            node = int(file.read())
        if node < 0:
            return None
        with open(f'/sys/devices/system/node/node{node}/cpulist') as file:
            text = file.read().strip()
        cpus = set()
        for part in text.split(','):
            first, _, last = part.partition('-')
            cpus.update(range(int(first), int(last or first) + 1))
        allowed = os.sched_getaffinity(0)
        cpus &= allowed
        if not cpus or cpus == allowed:
            return None
        return sorted(cpus)
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def pool_near(cpus):
    """The library's file pool - its own threads - onto `cpus`."""
    array = np.asarray(cpus, dtype=np.int32)
    runtime.check(runtime.library().emph_files_affinity(
        array.ctypes.data, len(array)), 'emph_files_affinity')


class FileAudio:
    """Stands in for the 1-D tensor of a mono 16-bit PCM / float32 WAVE file
    whose samples have not been read: the session reads them straight into its
    staging buffer (`Session` looks at `shape`, `dtype`, `is_cuda` only)."""
    __slots__ = ('batch', 'index', 'shape', 'dtype', 'rate', 'staged')
    is_cuda = False

    def __init__(self, batch, index, samples, dtype, rate):
        self.batch, self.index = batch, index
        self.shape = (int(samples),)
        self.dtype = dtype
        self.rate = int(rate)
        self.staged = None      # byte offset in `batch.staging` once read

    def dim(self):
        return 1

    def reshape(self, *shape):
        return self

    def tensor(self):
        """The samples as a tensor (read through `load.wav`): for callers
        that need values on the host, e.g. a mixed int16 / float32 batch."""
        samples, _ = load.wav(self.batch.audio_files[self.index], raw=True)
        return samples[0]


class FileBatch:
    """`count` (alignment file, audio file) pairs opened by the library."""

    def __init__(self, text_files, audio_files, threads=None):
        self.text_files = [str(file) for file in text_files]
        self.audio_files = [str(file) for file in audio_files]
        self.count = len(self.text_files)
        self.threads = int(threads or THREADS)
        lib = runtime.library()
        self._lib = lib
        self._handle = ctypes.c_void_p()
        self._text = (ctypes.c_char_p * max(self.count, 1))(
            *[file.encode() for file in self.text_files])
        self._audio = (ctypes.c_char_p * max(self.count, 1))(
            *[file.encode() for file in self.audio_files])
        runtime.check(lib.emph_files_open(
            self._text, self._audio, self.count, self.threads,
            ctypes.byref(self._handle)), 'emph_files_open')
        sizes = np.zeros((self.count, 12), dtype=np.int64)
        runtime.check(lib.emph_files_sizes(
            self._handle, sizes.ctypes.data), 'emph_files_sizes')
        self.sizes = sizes
        self.status = sizes[:, 0]
        words = sizes[:, 1]
        self.word_first = np.concatenate([[0], np.cumsum(words)])
        self.times = np.zeros((int(words.sum()), 2), dtype=np.float64)
        runtime.check(lib.emph_files_alignments(
            self._handle, self.times.ctypes.data, None, None, None, None,
            None, None, None, None), 'emph_files_alignments')
        self._labels = None
        self._audios = [None] * self.count
        self.staging = None

    def close(self):
        if self._handle:
            self._lib.emph_files_close(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass

    def error(self, index):
        return self._lib.emph_files_error(self._handle, index).decode(
            'utf-8', 'replace')

    ###########################################################################
    # Alignments
    ###########################################################################

    def labels(self):
        """Word / phoneme labels and tier names of the whole batch (fetched
        once, on the first request: the planner needs the times only)."""
        if self._labels is None:
            sizes = self.sizes
            phones = np.maximum(sizes[:, 2], 0)
            word_blob = ctypes.create_string_buffer(int(sizes[:, 3].sum()) + 1)
            phone_blob = ctypes.create_string_buffer(int(sizes[:, 4].sum()) + 1)
            word_end = np.zeros(int(sizes[:, 1].sum()), dtype=np.int64)
            phone_end = np.zeros(int(phones.sum()), dtype=np.int64)
            phone_times = np.zeros((int(phones.sum()), 2), dtype=np.float64)
            phone_word = np.zeros(int(phones.sum()), dtype=np.int32)
            tier_blob = ctypes.create_string_buffer(
                int(self._lib.emph_files_tier_name_bytes(self._handle)) + 1)
            tier_end = np.zeros(self.count, dtype=np.int64)
            runtime.check(self._lib.emph_files_alignments(
                self._handle, None, word_blob, word_end.ctypes.data,
                phone_times.ctypes.data, phone_blob, phone_end.ctypes.data,
                phone_word.ctypes.data, tier_blob, tier_end.ctypes.data),
                'emph_files_alignments')
            self._labels = dict(
                word_blob=word_blob.raw, word_end=word_end,
                phone_blob=phone_blob.raw, phone_end=phone_end,
                phone_times=phone_times, phone_word=phone_word,
                phone_first=np.concatenate([[0], np.cumsum(phones)]),
                tier_blob=tier_blob.raw, tier_end=tier_end)
        return self._labels

    def words(self, index):
        """`Word` objects (with their `Phoneme`s) of file `index`."""
        labels = self.labels()

        def texts(blob, ends, lo, hi):
            edges = np.concatenate([[ends[lo - 1] if lo else 0], ends[lo:hi]])
            return [blob[a:b].decode('utf-8')
                    for a, b in zip(edges[:-1], edges[1:])]
        lo, hi = self.word_first[index], self.word_first[index + 1]
        has_phones = self.sizes[index, 2] >= 0
        words = [alignment_module.Word(name, a, b, [] if has_phones else None)
                 for name, (a, b) in zip(
                     texts(labels['word_blob'], labels['word_end'], lo, hi),
                     self.times[lo:hi])]
        if has_phones:
            plo = labels['phone_first'][index]
            phi = labels['phone_first'][index + 1]
            names = texts(labels['phone_blob'], labels['phone_end'], plo, phi)
            for name, (a, b), word in zip(
                    names, labels['phone_times'][plo:phi],
                    labels['phone_word'][plo:phi]):
                words[word].phonemes.append(
                    alignment_module.Phoneme(name, a, b))
        return words

    def tiers(self, index):
        labels = self.labels()
        begin = labels['tier_end'][index - 1] if index else 0
        word_tier, phone_tier = labels['tier_blob'][
            begin:labels['tier_end'][index]].decode('utf-8').split('\n')
        return (word_tier, phone_tier, bool(self.sizes[index, 11]))

    def alignment(self, index):
        """`Alignment` of file `index` (its `Word`s are built on demand); a
        file the library could not parse goes through `alignment.Alignment`,
        which raises what it always raised."""
        if self.status[index] & 1:
            return alignment_module.Alignment(self.text_files[index])
        lo, hi = self.word_first[index], self.word_first[index + 1]
        return alignment_module.Alignment.lazy(
            self.times[lo:hi], lambda: (self.words(index), self.tiers(index)))

    ###########################################################################
    # Audio
    ###########################################################################

    def audio(self, index):
        """`(samples, rate)`: a `FileAudio` for mono 16-bit PCM / float32
        files, else what `load.wav(file, raw=True)` returns."""
        if self.status[index] & 2:
            return load.wav(self.audio_files[index], raw=True)   # raises
        if self._audios[index] is not None:
            return self._audios[index], self._audios[index].rate
        _, _, _, _, _, code, channels, rate, bits, _, nbytes, _ = \
            self.sizes[index].tolist()
        if channels == 1 and (code, bits) in ((1, 16), (3, 32)):
            dtype = torch.int16 if code == 1 else torch.float32
            self._audios[index] = FileAudio(
                self, index, nbytes // (bits // 8), dtype, rate)
            return self._audios[index], int(rate)
        return load.wav(self.audio_files[index], raw=True)

    def all_alignments(self):
        """`alignment(i)` for every file, without per-file numpy work."""
        first = self.word_first.tolist()
        bad = (self.status & 1).tolist()
        times, lazy = self.times, alignment_module.Alignment.lazy
        return [
            self.alignment(i) if bad[i] else lazy(
                times[first[i]:first[i + 1]],
                lambda i=i: (self.words(i), self.tiers(i)))
            for i in range(self.count)]

    def all_times(self):
        """Word times of every file as float64 [W, 2] views of one table (what
        `batch.plan_batch` and `session.layout_key` take in place of alignment
        objects); `alignment(i)` itself for a file the library does not vouch
        for."""
        first = self.word_first.tolist()
        bad = (self.status & 1).tolist()
        times = self.times
        return [self.alignment(i) if bad[i] else times[first[i]:first[i + 1]]
                for i in range(self.count)]

    def all_audios(self):
        """`audio(i)` for every file: `[(FileAudio | tensor, rate)]`."""
        rows = self.sizes.tolist()
        result = []
        for index, row in enumerate(rows):
            status, code, channels, rate, bits, nbytes = \
                row[0], row[5], row[6], row[7], row[8], row[10]
            if status & 2 or channels != 1 or \
                    (code, bits) not in ((1, 16), (3, 32)):
                result.append(self.audio(index))
                continue
            if self._audios[index] is None:
                self._audios[index] = FileAudio(
                    self, index, nbytes // (bits // 8),
                    torch.int16 if code == 1 else torch.float32, rate)
            result.append((self._audios[index], rate))
        return result

    def read_all(self, staging):
        """Read the samples of every `FileAudio` of the batch back to back
        into `staging` (a pinned uint8 tensor, on whatever thread): the
        session then sends them to the device straight from there
        (`Session` needs no copy of its own).  Returns the bytes used."""
        members = [audio for audio in self._audios if audio is not None]
        nbytes = [audio.shape[0] * (2 if audio.dtype == torch.int16 else 4)
                  for audio in members]
        where = np.concatenate([[0], np.cumsum(
            [(n + 3) // 4 * 4 for n in nbytes])]).astype(np.int64)
        if int(where[-1]) > staging.numel():
            raise ValueError('staging buffer too small')
        if members:
            self.read([audio.index for audio in members], where[:-1], nbytes,
                      staging.data_ptr())
        for audio, offset in zip(members, where[:-1]):
            audio.staged = int(offset)
        self.staging = staging
        return int(where[-1])

    def native_audio(self):
        """bool [count]: files whose samples the library reads itself (mono
        16-bit PCM or float32; `all_audios` gives a `FileAudio` for them)."""
        sizes = self.sizes
        return ((sizes[:, 0] & 2) == 0) & (sizes[:, 6] == 1) & (
            ((sizes[:, 5] == 1) & (sizes[:, 8] == 16)) |
            ((sizes[:, 5] == 3) & (sizes[:, 8] == 32)))

    def staged_format(self, sample_rate):
        """torch.int16 / torch.float32 when EVERY file of the batch is one the
        library reads itself, at `sample_rate`, in that one sample format (then
        `read_staged` serves the whole batch); None otherwise."""
        sizes = self.sizes
        if not self.count or self.status.any() or \
                not bool(self.native_audio().all()) or \
                not bool((sizes[:, 7] == sample_rate).all()):
            return None
        code = int(sizes[0, 5])
        if not bool((sizes[:, 5] == code).all()):
            return None
        return torch.int16 if code == 1 else torch.float32

    def read_staged(self, staging):
        """`read_all` for a batch with a `staged_format`: the samples of every
        file back to back (4-byte aligned) into `staging`; returns (where int64
        [count] byte offsets, lengths int64 [count] samples) - no object per
        file."""
        sizes = self.sizes
        nbytes = np.ascontiguousarray(sizes[:, 10])
        lengths = nbytes // (sizes[:, 8] // 8)
        padded = (nbytes + 3) // 4 * 4
        where = np.cumsum(padded) - padded
        if int(where[-1] + padded[-1]) > staging.numel():
            raise ValueError('staging buffer too small')
        self.read(np.arange(self.count, dtype=np.int32), where, nbytes,
                  staging.data_ptr())
        self.staging = staging
        return where, lengths

    def audio_bytes(self):
        """Bytes `read_all` needs (every file's samples, 4-byte aligned)."""
        native = self.native_audio()
        return int(((self.sizes[native, 10] + 3) // 4 * 4).sum())

    def read(self, indices, where, nbytes, destination):
        """Samples of files `indices` to host address `destination +
        where[k]` (the pinned staging buffer)."""
        which = np.asarray(indices, dtype=np.int32)
        where = np.asarray(where, dtype=np.int64)
        nbytes = np.asarray(nbytes, dtype=np.int64)
        runtime.check(self._lib.emph_files_read_audio(
            self._handle, which.ctypes.data, where.ctypes.data,
            nbytes.ctypes.data, len(which), destination, self.threads),
            'emph_files_read_audio')

    ###########################################################################
    # Outputs
    ###########################################################################

    def write(self, indices, prefixes, scores):
        """`<prefix>.TextGrid` + `<prefix>.pt` for files `indices`
        (`core.py:111-112`); scores: list of float32 CPU tensors [1, W]."""
        native = [k for k, i in enumerate(indices)
                  if not self.status[i] & 1]
        for k, index in enumerate(indices):
            if self.status[index] & 1:      # (a JSON alignment, say)
                from . import core
                core._save(self.alignment(index), scores[k], prefixes[k])
        if not native:
            return
        if hasattr(scores, 'flat') and \
                len(native) == len(indices) == len(scores) and \
                scores.flat.dtype == torch.float32 and \
                not scores.flat.is_cuda and scores.flat.numel():
            # (`session.Scores`: the batch's dense row as it is)
            flat = scores.flat.reshape(-1).contiguous()
            first = scores.first
        else:
            flat = torch.cat(
                [scores[k].reshape(-1) for k in native] +
                [torch.zeros(1)]).to(torch.float32).contiguous()
            first = np.concatenate([[0], np.cumsum(
                [scores[k].numel() for k in native])]).astype(np.int64)
        which = np.array([indices[k] for k in native], dtype=np.int32)
        paths = (ctypes.c_char_p * len(native))(
            *[str(prefixes[k]).encode() for k in native])
        runtime.check(self._lib.emph_files_write(
            self._handle, which.ctypes.data, paths, flat.data_ptr(),
            first.ctypes.data, len(native), self.threads), 'emph_files_write')
