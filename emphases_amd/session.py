"""Batches in flight: host staging, H2D, kernels and D2H of consecutive
batches overlap.

The reference processes one file at a time, synchronously
(`emphases/core.py:169-179`; audio goes to the device inside
`data/preprocess/core.py:74`, scores come back in `core.py:112`).  A `Session`
keeps `depth` lanes, each with its own HIP stream, engine workspace, pinned
staging buffer and pinned result buffer:

    submit(batch i+1)   plan on the host, gather the utterances into pinned
                        memory (`emph_host_gather`: the library's pool of copy
                        threads), enqueue the host-to-device copy + the kernels
                        + the copy of the scores back, all on the lane's
                        stream, and return
    result(batch i)     wait for that lane's event, split the scores

so the PCIe transfer and the host planning of one batch run under the kernels
of the previous one.  16-bit PCM tensors (what a WAV file holds; `load.pcm`)
are staged and transferred as they are - half the bytes - and converted by the
front-end kernel (identical scores: x / 32768 is exact).
"""
import collections
import os
import threading

import numpy as np
import torch

from . import batch
from . import config as cfg
from . import load

# (`emph_host_gather` takes 1 .. 64 copy threads)
COPY_THREADS = min(64, max(1, int(os.environ.get('EMPHASES_COPY_THREADS', 16))))
# A synchronous call with more audio than twice this is run as consecutive
# sub-batches of about this size over the lanes, so that the kernels of one
# run under the PCIe transfer of the next (a rank's 1.28 GB share of BASELINE
# configs[3]: 23 ms of DMA at 55 GB/s in front of 5.4 ms of kernels).
SPLIT_BYTES = int(os.environ.get('EMPHASES_SPLIT_BYTES', 256 << 20))
# pinned buffers the file API's openers rotate through (`core.files_to_scores`
# keeps FILE_BUFFERS - 2 batches being opened ahead of the one in flight)
# (at least three: the batch in flight, the one finishing, one being opened)
FILE_BUFFERS = max(3, int(os.environ.get("EMPHASES_FILE_BUFFERS", 4)))


def host_float32(audio):
    """float32 copy of a host tensor of another dtype, through numpy: a
    `Tensor.to` of a long signal is a torch CPU parallel region (runtime.py)."""
    if audio.is_cuda:
        return audio.to(torch.float32)
    return torch.from_numpy(audio.detach().numpy().astype(np.float32))


def host_contiguous(audio):
    """`audio.contiguous()` of a host tensor without a torch copy kernel."""
    if audio.is_cuda or audio.is_contiguous():
        return audio.contiguous()
    return torch.from_numpy(np.ascontiguousarray(audio.detach().numpy()))


def host_pcm_to_float(audio):
    """16-bit PCM -> float32 (x / 32768, exact) on the host, through numpy."""
    if audio.dtype != torch.int16:
        return audio if audio.dtype == torch.float32 else host_float32(audio)
    if audio.is_cuda:
        return audio.to(torch.float32) / 32768.
    return torch.from_numpy(
        audio.detach().numpy().astype(np.float32) * np.float32(1. / 32768.))


def mono(audio):
    """1-D tensor of channel 0 (`mels.py:48` featurises channel 0 only), at
    the caller's sample rate; int16 stays int16."""
    if not torch.is_tensor(audio):
        return audio                  # files.FileAudio: mono by construction
    audio = audio[0] if audio.dim() == 2 else audio.reshape(-1)
    if audio.dtype not in (torch.float32, torch.int16):
        audio = host_float32(audio)
    return audio


class _Lane:
    """One batch in flight."""

    def __init__(self, engine):
        self.engine = engine.lane()
        self.device = engine.device
        self.stream = torch.cuda.Stream(device=self.device)
        self.done = torch.cuda.Event()
        self.staging = None           # pinned uint8
        self.audio = None             # device uint8
        self.raw = None               # device uint8: audio before resampling
        self.result = None            # pinned float32
        self.pending = None
        # `Pending.result()` of this lane's batch against a `submit` that takes
        # the lane again from another thread
        self.lock = threading.RLock()
        # recurring batch layouts: key -> _Layout (plan, device metadata and,
        # from the second sighting on, the captured HIP graph of the forward)
        self.layouts = collections.OrderedDict()

    def _reserve(self, nbytes, words):
        if self.staging is None or self.staging.numel() < nbytes:
            size = max(nbytes, 1) * 5 // 4
            self.staging = torch.empty(size, dtype=torch.uint8).pin_memory()
            self.audio = torch.empty(
                size, dtype=torch.uint8, device=self.device)
            self.layouts.clear()      # captured graphs point at the old buffer
        if self.result is None or self.result.numel() < words:
            self.result = torch.empty(
                max(words, 1) * 5 // 4, dtype=torch.float32).pin_memory()

    def stage(self, audios, lengths, dtype, raw=False):
        """Packed device tensor of all utterances (dtype float32 or int16),
        enqueued on this lane's stream.  `raw`: into the buffer that holds
        audio at the caller's rate, in front of the resampler."""
        item = 2 if dtype == torch.int16 else 4
        total = int(sum(lengths))
        offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        if raw:
            if self.raw is None or self.raw.numel() < total * item:
                self.raw = torch.empty(
                    max(total * item, 1) * 5 // 4, dtype=torch.uint8,
                    device=self.device)
            buffer = self.raw
        else:
            buffer = self.audio
        device_view = buffer[:max(total, 1) * item].view(dtype)[:total]
        on_host = [i for i, a in enumerate(audios) if not a.is_cuda]
        host_view = self.staging[:max(total, 1) * item].view(dtype)[:total]
        if any(not torch.is_tensor(audios[i]) for i in on_host):
            return self._stage_files(
                audios, lengths, item, offsets, host_view, device_view)
        # The gather runs in the library (`emph_host_gather`: a persistent pool
        # of copy threads; numpy copies from a Python thread pool reach 58 GB/s
        # and torch's CPU copy_ into a slice of a large tensor a tenth of that,
        # tools/h2d_paths.py).  A few pieces, each sent to the device as soon as
        # it is gathered, so the DMA of one runs under the gather of the next.
        from . import runtime
        lib = runtime.library()
        sources = [host_contiguous(audios[i]) for i in on_host]
        pointers = np.array([a.data_ptr() for a in sources], dtype=np.int64)
        nbytes = np.array([lengths[i] * item for i in on_host], dtype=np.int64)
        where = np.array([offsets[i] * item for i in on_host], dtype=np.int64)
        threads = max(1, min(COPY_THREADS, total * item // (3 << 20)))
        pieces = max(1, min(4, len(on_host), total * item // (8 << 20)))
        edges = np.linspace(0, len(on_host), pieces + 1).astype(int)
        base = host_view.data_ptr()
        for lo, hi in zip(edges[:-1], edges[1:]):
            if hi == lo:
                continue
            runtime.check(lib.emph_host_gather(
                pointers[lo:hi].ctypes.data, nbytes[lo:hi].ctypes.data,
                where[lo:hi].ctypes.data, int(hi - lo), base, int(threads)),
                'emph_host_gather')
            # contiguous runs of the piece go to the device
            first = lo
            for k in range(lo + 1, hi + 1):
                if k == hi or on_host[k] != on_host[k - 1] + 1:
                    start = int(offsets[on_host[first]])
                    stop = int(offsets[on_host[k - 1] + 1])
                    device_view[start:stop].copy_(
                        host_view[start:stop], non_blocking=True)
                    first = k
        for i, audio in enumerate(audios):
            if audio.is_cuda:
                device_view[offsets[i]:offsets[i + 1]].copy_(
                    audio, non_blocking=True)
        return device_view


    def _stage_files(self, audios, lengths, item, offsets, host_view,
                     device_view):
        """`stage` for a batch whose audio is still on disk
        (`files.FileAudio`): the library's threads read the data chunks
        straight into the pinned staging buffer (`emph_files_read_audio`), a
        few pieces at a time, each sent to the device as soon as it is read;
        tensors among them take the copy path."""
        from . import runtime
        count = len(audios)
        if all(not torch.is_tensor(audio) and audio.staged is not None
               for audio in audios):
            # already read into the batch's own pinned buffer (by the opener
            # thread of core.files_to_scores): one DMA per run of files that
            # sit back to back there
            dtype = device_view.dtype
            first = 0
            for k in range(1, count + 1):
                previous = audios[k - 1]
                if k < count and audios[k].batch is previous.batch and \
                        audios[k].staged == previous.staged + \
                        lengths[k - 1] * item:
                    continue
                head = audios[first]
                start, stop = int(offsets[first]), int(offsets[k])
                source = head.batch.staging[
                    head.staged:head.staged + (stop - start) * item]
                device_view[start:stop].copy_(
                    source.view(dtype), non_blocking=True)
                first = k
            return device_view
        base = host_view.data_ptr()
        pieces = max(1, min(4, count, int(offsets[-1]) * item // (8 << 20)))
        edges = np.linspace(0, count, pieces + 1).astype(int)
        for lo, hi in zip(edges[:-1], edges[1:]):
            if hi == lo:
                continue
            by_batch = {}
            tensors = []
            for i in range(lo, hi):
                audio = audios[i]
                if torch.is_tensor(audio):
                    if not audio.is_cuda:
                        tensors.append(i)
                    continue
                by_batch.setdefault(id(audio.batch), (audio.batch, []))[1] \
                    .append(i)
            for batch_files, members in by_batch.values():
                batch_files.read(
                    [audios[i].index for i in members],
                    [int(offsets[i]) * item for i in members],
                    [lengths[i] * item for i in members], base)
            if tensors:
                sources = [host_contiguous(audios[i]) for i in tensors]
                # (named: an array that is only a temporary of the argument
                # list is freed before the call reads it)
                pointers = np.array(
                    [a.data_ptr() for a in sources], dtype=np.int64)
                sizes = np.array(
                    [lengths[i] * item for i in tensors], dtype=np.int64)
                where = np.array(
                    [int(offsets[i]) * item for i in tensors], dtype=np.int64)
                runtime.check(runtime.library().emph_host_gather(
                    pointers.ctypes.data, sizes.ctypes.data, where.ctypes.data,
                    len(tensors), base, COPY_THREADS), 'emph_host_gather')
            start, stop = int(offsets[lo]), int(offsets[hi])
            device_view[start:stop].copy_(
                host_view[start:stop], non_blocking=True)
        for i, audio in enumerate(audios):
            if torch.is_tensor(audio) and audio.is_cuda:
                device_view[offsets[i]:offsets[i + 1]].copy_(
                    audio, non_blocking=True)
        return device_view


class Staged:
    """A batch of 16 kHz mono utterances whose samples already sit in a pinned
    buffer (`files.FileBatch.read_staged`): utterance u's `lengths[u]` samples
    of `dtype` (int16 = 16-bit PCM, or float32) start at byte `where[u]` of
    `buffer`.  What `Session.submit_staged` takes in place of one object per
    utterance."""
    __slots__ = ('buffer', 'where', 'lengths', 'dtype')

    def __init__(self, buffer, where, lengths, dtype):
        self.buffer = buffer
        self.where = np.asarray(where, dtype=np.int64)
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.dtype = dtype

    def __len__(self):
        return len(self.lengths)


class _Layout:
    """What can be kept when the same batch layout (word times, utterance
    lengths, batch_size, sample format) comes back: the plan, its metadata on
    the device, and the forward captured into a HIP graph."""
    KEEP = 8

    def __init__(self, plan):
        self.plan = plan
        self.meta = None
        self.replay = None
        self.scores = None
        self.seen = 0


def layout_key(alignments, lengths, batch_size, dtype):
    """Hashable identity of a batch layout, or None when an alignment is not
    one whose word times can be read as an array."""
    from . import alignment as alignment_module
    tables = []
    for item in alignments:
        if type(item) is alignment_module.Alignment:
            tables.append(item.times())
        elif isinstance(item, np.ndarray):
            tables.append(np.asarray(item, dtype=np.float64).reshape(-1, 2))
        else:
            return None
    times = np.concatenate(tables) if tables else np.zeros((0, 2))
    return (hash(times.tobytes()), tuple(len(t) for t in tables),
            tuple(lengths), batch_size, dtype)


class Scores:
    """The scores of a batch as a sequence of per-utterance [1, W_u] views of
    ONE dense row (`flat`, with `first[u] .. first[u + 1]` the columns of
    utterance u): the views are made when asked for, so a consumer that wants
    the row itself (`files.FileBatch.write`) creates no object per utterance."""

    def __init__(self, dense, counts):
        self.flat = dense                            # [1, total]
        self.first = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)

    def __len__(self):
        return len(self.first) - 1

    def __getitem__(self, index):
        if isinstance(index, slice):
            return [self[i] for i in range(*index.indices(len(self)))]
        if index < 0:
            index += len(self)
        if not 0 <= index < len(self):
            raise IndexError(index)
        return self.flat[:, int(self.first[index]):int(self.first[index + 1])]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Pending:
    """Scores of a submitted batch; `result()` waits for them (a list of
    per-utterance tensors), `scores()` for the same as a `Scores`."""

    def __init__(self, lane, plan, count, on_device, ld_words):
        self._lane = lane
        self._plan = plan
        self._count = count
        self._on_device = on_device
        self._ld_words = ld_words
        self._scores = None
        self._value = None
        self._dense = None

    def result(self):
        if self._value is None:
            self._value = list(self.scores())
        return self._value

    def scores(self):
        if self._dense is not None:
            return self._dense
        lane, plan = self._lane, self._plan
        if plan is None or not len(plan):
            empty = torch.zeros((1, 0))
            if self._on_device:
                empty = empty.to(lane.device)
            self._dense = Scores(empty, np.zeros(self._count, dtype=np.int64))
            return self._dense
        with lane.lock:
            if self._dense is not None:
                return self._dense
            lane.done.synchronize()
            if self._on_device:
                packed = self._scores
                # allocated on the lane's stream, consumed on the caller's: the
                # caching allocator must not hand the block to the lane's next
                # clone while the caller's kernels still read it
                packed.record_stream(torch.cuda.current_stream(lane.device))
                columns = torch.from_numpy(plan.word_columns())
                dense = packed[columns.to(packed.device)][None]
            else:
                # (numpy: the gather of a corpus-sized batch through torch would
                # be a CPU parallel region, runtime.py; the fancy index copies,
                # so the pinned buffer is free for the lane's next batch)
                dense = torch.from_numpy(
                    lane.result[:self._ld_words].numpy()[
                        plan.word_columns()])[None]
            if lane.pending is self:
                lane.pending = None
        # one gather of the valid word columns, one split: per-utterance views
        # [1, W_u] of a dense row (an utterance's chunks are consecutive)
        counts = np.bincount(
            plan.utterance, weights=plan.words, minlength=self._count)
        self._dense = Scores(dense, counts.astype(np.int64))
        self._scores = None
        return self._dense


class Session:
    """`depth` batches in flight on one device."""

    def __init__(self, engine, depth=2):
        self.engine = engine
        self.lanes = [_Lane(engine) for _ in range(max(1, depth))]
        self._cursor = 0
        self._lock = threading.Lock()
        self._kernels = {}        # (rate, target) -> resampling kernel on the device
        self._file_buffers = []   # pinned uint8 tensors for files.FileBatch.read_all

    def _resample(self, lane, audios, lengths, dtype, sample_rate,
                  target_rate=cfg.SAMPLE_RATE):
        """`emphases.resample` (`core.py:613-619`) for the whole batch on the
        device: stage the audio at its own rate, one `emph_resample` launch
        into the lane's packed buffer at `target_rate`.  Returns (packed
        float32 device tensor, lengths at the target rate)."""
        from . import runtime
        rates = (int(sample_rate), int(target_rate))
        if rates not in self._kernels:
            kernel, orig, new, width = load.resample_kernel(*rates)
            self._kernels[rates] = (
                kernel.reshape(new, -1).contiguous().to(lane.device),
                orig, new, width)
        kernel, orig, new, width = self._kernels[rates]
        targets = [load.resampled_length(n, orig, new) for n in lengths]
        source = np.cumsum([0] + lengths)
        target = np.cumsum([0] + targets)
        table = np.stack(
            [source[:-1], lengths, target[:-1], targets], axis=1).astype(np.int64)
        raw = lane.stage(audios, lengths, dtype, raw=True)
        table_device = lane.engine._pinned(
            'resample', table.view(np.int32).ravel()).to(
                lane.device, non_blocking=True)
        out = lane.audio[:max(int(target[-1]), 1) * 4].view(torch.float32)[
            :int(target[-1])]
        runtime.check(runtime.library().emph_resample(
            raw.data_ptr(), 1 if dtype == torch.int16 else 0,
            table_device.data_ptr(), len(lengths), max(targets + [0]),
            kernel.data_ptr(), orig, new, width, out.data_ptr(),
            runtime.stream()), 'emph_resample')
        return out, targets

    def file_buffer(self, turn, nbytes):
        """Pinned uint8 tensor number `turn % FILE_BUFFERS` of at least
        `nbytes` (four rotate: two being read into by the openers, one in
        flight, one whose batch is being finished); kept by the session, since
        pinning memory takes milliseconds."""
        while len(self._file_buffers) < FILE_BUFFERS:
            self._file_buffers.append(None)
        slot = turn % FILE_BUFFERS
        buffer = self._file_buffers[slot]
        if buffer is None or buffer.numel() < nbytes:
            buffer = torch.empty(
                max(nbytes, 1) * 5 // 4, dtype=torch.uint8).pin_memory()
            self._file_buffers[slot] = buffer
        return buffer

    def resample(self, audios, sample_rate, target_rate=cfg.SAMPLE_RATE,
                 on_device=False):
        """`emphases.resample` (`core.py:613-619`) of 1-D tensors (float32, or
        int16 = 16-bit PCM) on the device - the ONE resampler of the package:
        the batch API, the step API (`core.preprocess`), `load.audio` and the
        pitch tracker's 16 kHz input all come through `emph_resample`.
        Returns float32 tensors (on the device, or on the host)."""
        audios = [audio.reshape(-1) for audio in audios]
        if not audios:
            return []
        pcm = all(audio.dtype == torch.int16 for audio in audios)
        dtype = torch.int16 if pcm else torch.float32
        if not pcm:
            audios = [host_pcm_to_float(audio) for audio in audios]
        lengths = [int(audio.shape[0]) for audio in audios]
        _, orig, new, _ = load.resample_kernel(sample_rate, target_rate)
        targets = [load.resampled_length(n, orig, new) for n in lengths]
        with self._lock:
            lane = self.lanes[self._cursor % len(self.lanes)]
            self._cursor += 1
        with lane.lock:
            if lane.pending is not None:
                lane.pending.result()
            lane._reserve(max(sum(targets) * 4, sum(lengths) * 4), 1)
            if any(audio.is_cuda for audio in audios):
                lane.stream.wait_stream(torch.cuda.current_stream(lane.device))
            with torch.cuda.device(lane.device), \
                    torch.cuda.stream(lane.stream):
                packed, _ = self._resample(
                    lane, audios, lengths, dtype, sample_rate, target_rate)
                packed = packed.clone() if on_device else packed.cpu()
            lane.stream.synchronize()
        if on_device:
            packed.record_stream(torch.cuda.current_stream(lane.device))
        return list(packed.split(targets))

    def submit(self, alignments, audios, sample_rate=cfg.SAMPLE_RATE,
               batch_size=None, on_device=False, pitch_tracker=None,
               plan=None):
        """Enqueue a batch; returns a `Pending`.  The lane it takes is the one
        whose batch was submitted `depth` submissions ago: that batch's
        results are extracted first if the caller has not done so.  `plan`:
        the batch's plan when the caller has it already (`batch.plan_batch` of
        the same alignments and 16 kHz lengths, maybe `Engine.prepare`d on
        another thread): no layout cache lookup, no planning here."""
        with self._lock:
            lane = self.lanes[self._cursor % len(self.lanes)]
            self._cursor += 1
            with lane.lock:
                if lane.pending is not None:
                    lane.pending.result()
                try:
                    return self._enqueue(
                        lane, list(alignments), list(audios), sample_rate,
                        batch_size, on_device, pitch_tracker, plan)
                except BaseException:
                    # Copies and kernels may already be queued on the lane's
                    # stream with no `done` event behind them: drain it before
                    # the pinned staging buffer can be reused, and forget the
                    # layouts (one of them may be half built / half captured).
                    try:
                        lane.stream.synchronize()
                    except Exception:     # noqa: BLE001
                        pass
                    lane.layouts.clear()
                    lane.pending = None
                    raise

    def submit_staged(self, plan, staged, on_device=False):
        """`submit` of a batch that is planned (`batch.plan_batch` of its
        alignments and lengths, maybe `Engine.prepare`d) and whose samples are
        in a pinned buffer already (`Staged`): the DMA of every run of
        utterances that lie back to back there, the kernels, the scores' way
        back.  Nothing here is per utterance."""
        if self.engine.config.pitch_feature or \
                self.engine.config.periodicity_feature:
            raise ValueError(
                'submit_staged: pitch / periodicity features need the samples '
                'on the host (use submit)')
        with self._lock:
            lane = self.lanes[self._cursor % len(self.lanes)]
            self._cursor += 1
            with lane.lock:
                if lane.pending is not None:
                    lane.pending.result()
                try:
                    return self._enqueue_staged(lane, plan, staged, on_device)
                except BaseException:
                    try:
                        lane.stream.synchronize()
                    except Exception:     # noqa: BLE001
                        pass
                    lane.layouts.clear()
                    lane.pending = None
                    raise

    def _enqueue_staged(self, lane, plan, staged, on_device):
        count = len(staged)
        pending = Pending(
            lane, plan, count, on_device,
            plan.ld_words if plan is not None else 0)
        if plan is None or not len(plan):
            return pending
        dtype = staged.dtype
        item = 2 if dtype == torch.int16 else 4
        offsets = np.concatenate([[0], np.cumsum(staged.lengths)])
        total = int(offsets[-1])
        lane._reserve(total * item, plan.ld_words)
        # runs of utterances that lie back to back in the pinned buffer
        nbytes = staged.lengths * item
        cuts = np.nonzero(
            staged.where[1:] != staged.where[:-1] + nbytes[:-1])[0] + 1
        first = np.concatenate([[0], cuts]).tolist()
        last = np.concatenate([cuts, [count]]).tolist()
        with torch.cuda.device(lane.device), torch.cuda.stream(lane.stream):
            packed = lane.audio[:max(total, 1) * item].view(dtype)[:total]
            for lo, hi in zip(first, last):
                start, stop = int(offsets[lo]), int(offsets[hi])
                source = int(staged.where[lo])
                packed[start:stop].copy_(
                    staged.buffer[source:source + (stop - start) * item]
                    .view(dtype), non_blocking=True)
            scores, _ = lane.engine.forward(packed, plan)
            if on_device:
                pending._scores = scores.clone()
            else:
                lane.result[:plan.ld_words].copy_(scores, non_blocking=True)
            lane.done.record(lane.stream)
        lane.pending = pending
        return pending

    def _enqueue(self, lane, alignments, audios, sample_rate, batch_size,
                 on_device, pitch_tracker, ready=None):
        audios = [mono(audio) for audio in audios]
        resampling = int(sample_rate) != cfg.SAMPLE_RATE
        pcm = bool(audios) and all(
            audio.dtype == torch.int16 for audio in audios)
        dtype = torch.int16 if pcm else torch.float32
        if not pcm:
            # int16 is 16-bit PCM wherever it appears (x / 32768, exact): a
            # mixed batch gives every utterance the bits of its own call
            # (a 16-bit file among float32 ones is read on the host for that)
            audios = [audio if torch.is_tensor(audio) or
                      audio.dtype != torch.int16 else audio.tensor()
                      for audio in audios]
            audios = [host_pcm_to_float(audio)
                      if audio.dtype == torch.int16 else audio
                      for audio in audios]
        raw_lengths = [int(audio.shape[0]) for audio in audios]
        lengths = raw_lengths
        if resampling:
            _, orig, new, _ = load.resample_kernel(sample_rate)
            lengths = [load.resampled_length(n, orig, new)
                       for n in raw_lengths]
            dtype_in, dtype = dtype, torch.float32
        layout = None
        key = layout_key(alignments, lengths, batch_size, dtype) \
            if audios and ready is None else None
        if ready is not None:
            layout = _Layout(ready)
        if key is not None:
            layout = lane.layouts.get(key)
            if layout is not None:
                lane.layouts.move_to_end(key)
        fresh = layout is None and bool(audios)
        if fresh:
            layout = _Layout(
                batch.plan_batch(alignments, lengths, batch_size))
        plan = layout.plan if layout is not None else None
        pending = Pending(
            lane, plan, len(audios), on_device,
            plan.ld_words if plan is not None else 0)
        if plan is None or not len(plan):
            return pending
        engine = lane.engine
        # (growing the buffers drops the cached layouts, whose graphs point
        # into the old ones: so before this layout joins the cache)
        lane._reserve(
            max(sum(lengths) * (2 if dtype == torch.int16 else 4),
                sum(raw_lengths) * 4 if resampling else 0),
            plan.ld_words)
        if fresh and key is not None:
            lane.layouts[key] = layout
            while len(lane.layouts) > _Layout.KEEP:
                lane.layouts.popitem(last=False)
        if any(audio.is_cuda for audio in audios):
            # device-resident input may still be being written by kernels on
            # the caller's stream
            lane.stream.wait_stream(torch.cuda.current_stream(lane.device))
        with torch.cuda.device(lane.device), \
                torch.cuda.stream(lane.stream):
            tracks = None
            if engine.config.pitch_feature or \
                    engine.config.periodicity_feature:
                from . import core
                # (the tracker wants the samples on the host)
                audios = [audio if torch.is_tensor(audio) else audio.tensor()
                          for audio in audios]
                # (the tracker runs on the host and wants 16 kHz audio)
                heard = audios
                if resampling:
                    heard, _ = self._resample(
                        lane, audios, raw_lengths, dtype_in, sample_rate)
                    lane.stream.synchronize()
                    heard = list(heard.cpu().split(lengths))
                tracks = core._tracks(
                    engine, plan, heard, pitch_tracker, lane.device.index)
            if resampling:
                packed, _ = self._resample(
                    lane, audios, raw_lengths, dtype_in, sample_rate)
            else:
                packed = lane.stage(audios, lengths, dtype)
            layout.seen += 1
            if tracks is None and layout.replay is None and \
                    layout.seen >= 2 and key in lane.layouts:
                # the layout came back: from now on one graph launch
                layout.meta = engine.upload(plan)
                layout.replay, layout.scores, _ = engine.capture(
                    packed, plan, layout.meta)
            if tracks is None and layout.replay is not None:
                layout.replay()
                scores = layout.scores
            else:
                scores, _ = engine.forward(packed, plan, tracks=tracks)
            if on_device:
                pending._scores = scores.clone()
            else:
                lane.result[:plan.ld_words].copy_(
                    scores, non_blocking=True)
            lane.done.record(lane.stream)
        lane.pending = pending
        return pending

    def run(self, alignments, audios, sample_rate=cfg.SAMPLE_RATE,
            batch_size=None, on_device=False, pitch_tracker=None):
        """submit + result: one synchronous batch (very large ones as a few
        sub-batches in flight, see SPLIT_BYTES; an utterance's scores do not
        depend on its neighbours in a batch)."""
        alignments, audios = list(alignments), list(audios)
        groups = self._groups(audios)
        pendings = [
            self.submit(alignments[lo:hi], audios[lo:hi], sample_rate,
                        batch_size, on_device, pitch_tracker)
            for lo, hi in groups]
        if len(pendings) == 1:
            return pendings[0].result()
        return [scores for pending in pendings for scores in pending.result()]

    def _groups(self, audios):
        """[(first, last + 1)] of the sub-batches of a synchronous call."""
        whole = [(0, len(audios))]
        if len(self.lanes) < 2 or len(audios) < 2:
            return whole
        sizes = np.array(
            [int(a.shape[-1]) * (2 if a.dtype == torch.int16 else 4)
             for a in audios], dtype=np.int64)
        total = int(sizes.sum())
        if total <= 2 * SPLIT_BYTES:
            return whole
        count = -(-total // SPLIT_BYTES)
        # equal shares of the bytes, cut at utterance boundaries
        ends = np.searchsorted(
            np.cumsum(sizes), total * np.arange(1, count + 1) / count, 'left') + 1
        ends = np.unique(np.minimum(ends, len(audios)))
        return list(zip(np.concatenate([[0], ends[:-1]]).tolist(),
                        ends.tolist()))
