"""Drop-in API surface of the prominence-inference path.

Same names, argument meaning and error behaviour as the reference's
`emphases/core.py` (signatures at `core.py:23-29, 76-83, 115-122, 182-189,
223-230`; step functions `preprocess` 345-418, `infer` 295-332, `postprocess`
335-342, `downsample` 426-469), with the hot path executed by the HIP library
instead of ATen.  Deliberate, documented deviations:

* scores are float32 (the reference returns bf16/fp16 because it runs under
  `torch.autocast`, `core.py:594-607`);
* `checkpoint=None` loads the bundled copy of the reference's trained weights
  instead of downloading from HuggingFace (`core.py:307-310`);
* there is no CPU execution path: `gpu=None` means "the current HIP device"
  and the result is returned on the CPU (as the reference would for
  `gpu=None`); with no device visible the call raises;
* `from_alignments_and_audios` is an addition: many utterances in one ragged
  batch (the reference loops one file at a time, `core.py:169-179`).
"""
import contextlib
import functools
import os
import time

import numpy as np
import torch

from . import alignment as alignment_module
from . import batch
from . import config as cfg
from . import engine as engine_module
from . import load
from . import runtime

_ACTIVE = [cfg.DEFAULT]


def configure(config=None, **overrides):
    """Select the active configuration (the reference mutates module globals
    through `yapecs --config file.py`; see `emphases/__init__.py:7-15`)."""
    if config is None:
        config = cfg.Config(**overrides)
    _ACTIVE[0] = config
    return config


def active_config():
    return _ACTIVE[0]


@functools.lru_cache(maxsize=8)
def _engine(checkpoint, device_index, config, conv_tile=None,
            precision='f32'):
    return engine_module.Engine(
        config, checkpoint, device_index, conv_tile=conv_tile,
        precision=precision)


def get_engine(checkpoint=None, gpu=None, config=None, conv_tile=None,
               precision='f32'):
    """Cached `Engine` per (checkpoint, device, config, conv_tile) — the
    explicit form of the reference's function-attribute cache
    (`core.py:298-315`).  `conv_tile`: positions per frame-rate conv tile.
    None (default): the kernel family of a layer follows from the
    configuration alone, so an utterance's scores are BITWISE the same alone,
    in any batch and on any shard; 64 / 32 / 16 pin a tile; 'auto' picks the
    lowest-latency variant by batch size (a few utterances then take the
    direct form, which agrees with the Winograd kernels to 1e-6, not
    bitwise).  `precision`: 'f32', or one of the opt-in split-bf16 names of
    `engine.PRECISIONS`."""
    device = runtime.require_gpu(gpu)
    index = device.index if device.index is not None else \
        torch.cuda.current_device()
    checkpoint = None if checkpoint is None else os.fspath(checkpoint)
    return _engine(checkpoint, index, config or active_config(), conv_tile,
                   precision)


###############################################################################
# Emphasis annotation API
###############################################################################


def penn_tracker(gpu=None):
    """The reference's pitch tracker call (`data/preprocess/core.py:84-92`) as
    a `pitch_tracker` callable; `penn` is third-party and optional."""
    try:
        import penn
    except ImportError as error:
        raise NotImplementedError(
            'PITCH_FEATURE / PERIODICITY_FEATURE need the third-party `penn` '
            'pitch tracker, which is not installed: pass pitch_tracker=, a '
            'callable (audio [1, S] at 16 kHz) -> (pitch [1, F] in Hz, '
            'periodicity [1, F])') from error

    def track(audio):
        return penn.from_audio(
            audio, cfg.SAMPLE_RATE, hopsize=cfg.HOPSIZE_SECONDS,
            fmin=cfg.FMIN, fmax=cfg.FMAX, pad=True, interp_unvoiced_at=50,
            gpu=gpu)
    return track


def _tracks(engine, plan, audios, pitch_tracker, gpu):
    """Device tensor [2, ld_frames] of tracker outputs, or None."""
    config = engine.config
    if not (config.pitch_feature or config.periodicity_feature):
        return None
    tracker = pitch_tracker or penn_tracker(gpu)
    pairs = [
        tracker(batch.chunk_audio(audios[segment.utterance], segment))
        for segment in plan.segments]
    packed = torch.from_numpy(batch.pack_tracks(plan, pairs))
    return packed.pin_memory().to(engine.device, non_blocking=True)


@functools.lru_cache(maxsize=8)
def _session(checkpoint, device_index, config, conv_tile=None,
             precision='f32'):
    from . import session
    return session.Session(
        _engine(checkpoint, device_index, config, conv_tile, precision),
        depth=2)


def get_session(checkpoint=None, gpu=None, config=None, conv_tile=None,
                precision='f32'):
    """Cached `session.Session` (batches in flight on their own streams) of
    the cached engine."""
    device = runtime.require_gpu(gpu)
    index = device.index if device.index is not None else \
        torch.cuda.current_device()
    checkpoint = None if checkpoint is None else os.fspath(checkpoint)
    return _session(checkpoint, index, config or active_config(), conv_tile,
                    precision)


def from_alignments_and_audios(alignments, audios, sample_rate=cfg.SAMPLE_RATE,
                               checkpoint=None, batch_size=None, gpu=None,
                               config=None, pitch_tracker=None,
                               conv_tile=None, precision='f32'):
    """Scores for many utterances in one ragged batch.

    audios: float tensors [1, S] (or [S]); int16 tensors are taken as 16-bit
        PCM (x / 32768, what `load.audio(file, raw=True)` returns) and travel
        to the device as they are.
    pitch_tracker: for configurations with PITCH_FEATURE / PERIODICITY_FEATURE,
        the stand-in for `penn.from_audio` (`data/preprocess/core.py:84-92`):
        `(chunk audio [1, Sc]) -> (pitch [1, Fc] Hz, periodicity [1, Fc])`;
        default: `penn` itself, if installed.
    precision: 'f32' (default), or 'bf16x3' / 'bf16x3_fast' / 'bf16x6'
        (`engine.PRECISIONS`).
    Returns a list of float32 tensors [1, W_i] (CPU if `gpu is None`)."""
    session = get_session(checkpoint, gpu, config, conv_tile, precision)
    return session.run(
        alignments, audios, sample_rate, batch_size,
        on_device=gpu is not None, pitch_tracker=pitch_tracker)


def from_alignment_and_audio(alignment, audio, sample_rate, checkpoint=None,
                             batch_size=None, gpu=None, precision='f32'):
    """Produce emphasis scores for each word (`core.py:223-265`).

    alignment: object with `len()`, `[i]` -> word with `.start()/.end()/
        .duration()` in seconds (the subset of `pypar.Alignment` the reference
        uses, `core.py:366-400`)
    audio: float tensor [1, S] (only channel 0 is featurised, `mels.py:48`)
    sample_rate, checkpoint, batch_size (max frames per chunk), gpu: as in
        the reference.
    precision: 'f32' (default), or one of the opt-in split-bf16 names of
        `engine.PRECISIONS` (an addition; every entry point below takes it).
    Returns float32 scores [1, W]; words of chunks the reference drops
    (`core.py:414-415`) have no score, as there."""
    return from_alignments_and_audios(
        [alignment], [audio], sample_rate, checkpoint, batch_size, gpu,
        precision=precision)[0]


def from_text_and_audio(text, audio, sample_rate, checkpoint=None,
                        batch_size=None, gpu=None):
    """`core.py:182-220` force-aligns the transcript with `pyfoal` (P2FA /
    HTK subprocesses, third-party) before inference; forced alignment is
    outside the accelerated path."""
    raise NotImplementedError(
        'forced alignment (pyfoal/P2FA) is not part of the HIP hot path; '
        'align the transcript first and call from_alignment_and_audio')


def from_file(text_file, audio_file, checkpoint=None, batch_size=None,
              gpu=None, precision='f32'):
    """`core.py:23-73`: scores for an alignment file (.TextGrid / .json) and
    an audio file."""
    if not str(text_file).endswith(('.TextGrid', '.json')):
        with open(text_file, encoding='utf-8') as file:
            return from_text_and_audio(
                file.read(), load.audio(audio_file), cfg.SAMPLE_RATE,
                checkpoint, batch_size, gpu)
    # the file's own rate: a file that is not at 16 kHz is resampled on the
    # device with the rest of the path (`core.py:44` does it on the host)
    samples, rate = load.wav(audio_file, raw=True)
    return from_alignment_and_audio(
        alignment_module.Alignment(text_file), samples, rate, checkpoint,
        batch_size, gpu, precision)


def _save(alignment, scores, output_prefix):
    """`core.py:111-112`: the alignment as .TextGrid, the scores as .pt"""
    alignment.save(f'{output_prefix}.TextGrid')
    torch.save(scores.cpu(), f'{output_prefix}.pt')


def from_file_to_file(text_file, audio_file, output_prefix=None,
                      checkpoint=None, batch_size=None, gpu=None,
                      precision='f32'):
    """`core.py:76-112`"""
    from pathlib import Path
    if output_prefix is None:
        output_prefix = Path(text_file).stem
    scores = from_file(text_file, audio_file, checkpoint, batch_size, gpu,
                       precision)
    _save(alignment_module.Alignment(text_file), scores, output_prefix)


def files_to_scores(text_files, audio_files, session, batch_size=None,
                    utterances_per_batch=256, deliver=None,
                    deliver_batch=None):
    """The loop of `core.py:169-179` over ragged batches of
    `utterances_per_batch` files, two batches in flight.  A batch of files is
    opened by the library in one call (`files.FileBatch`: TextGrids parsed and
    WAVE headers walked on a pool of host threads), the samples of batch i+1
    are read straight into the session's pinned staging buffer while batch i
    computes, 16-bit PCM files travel to the device as 16-bit PCM, files that
    are not at 16 kHz are resampled on the device (one submission per sample
    rate).  Results: `deliver_batch(opened, local indices, global indices,
    scores)` once per submission, or `deliver(index, alignment, scores)` per
    file, in order within a batch."""
    from . import files
    text_files, audio_files = list(text_files), list(audio_files)
    for file in text_files:
        if not str(file).endswith(('.TextGrid', '.json')):
            from_text_and_audio(None, None, None)
    _files_to_scores(text_files, audio_files, session, batch_size,
                     utterances_per_batch, deliver, deliver_batch)


# `files_to_scores` appends (stage, batch, start, end) in perf_counter_ns here when
# it is a list (tools/files_timeline.py): where the three stages of a batch run
TIMELINE = None


def _stamp(stage, position, start):
    if TIMELINE is not None:
        TIMELINE.append((stage, position, start, time.perf_counter_ns()))


def _files_to_scores(text_files, audio_files, session, batch_size,
                     utterances_per_batch, deliver, deliver_batch):
    """Three stages run side by side, each on a thread of its own (what the
    helpers call are library or numpy routines that leave the interpreter lock
    alone; none of them enters a torch CPU parallel region):

        openers  batches i + 1 and i + 2 (two threads: an opener waits on the
                 library's pool and on the interpreter lock in turn, so one of
                 them delivered a batch per 4.9 ms against 2.5 ms alone,
                 profiles/r5_files_timeline.txt): open + parse + header walk
                 (`files.FileBatch`), the samples into a pinned buffer of the
                 session (`read_all`), the plan (`batch.plan_batch`) and its
                 metadata tables (`Engine.prepare`)
        caller   batch i: DMA + kernels (`Session.submit`), then the scores
                 of batch i - 1
        writer   batch i - 1: `<prefix>.TextGrid` + `<prefix>.pt`

    The library's file pool is sized per stage from the CPUs the process may
    keep busy (`files.stage_threads`): a cgroup that allows fewer CPUs than the
    machine shows FREEZES the process for the rest of the period when opener,
    writer, caller and the HIP runtime's threads together exceed the quota.
    A failure anywhere still finishes and writes every batch submitted before
    it - like the reference's loop (`core.py:169-179`), which leaves the
    outputs of every file in front of the bad one - then raises."""
    import collections
    import concurrent.futures
    from . import files
    from . import session as session_module
    engine = session.engine
    open_threads, write_threads = files.stage_threads(session_module.FILE_BUFFERS - 2)
    # (the pitch tracker wants every utterance's samples on the host)
    tracked = engine.config.pitch_feature or engine.config.periodicity_feature

    def open_batch(first, last, turn):
        start = time.perf_counter_ns()
        opened = files.FileBatch(
            text_files[first:last], audio_files[first:last], open_threads)
        _stamp('open.parse', turn, start)
        begin = time.perf_counter_ns()
        dtype = None if tracked else opened.staged_format(cfg.SAMPLE_RATE)
        if dtype is not None:
            # every file read by the library, 16 kHz mono, one sample format:
            # samples, plan and tables without an object per file - the samples
            # straight into one of the session's pinned buffers, the plan from
            # the library's own table of word times
            with torch.cuda.device(session.engine.device):
                staging = session.file_buffer(turn, opened.audio_bytes())
            where, lengths = opened.read_staged(staging)
            _stamp('open.read', turn, begin)
            begin = time.perf_counter_ns()
            plan = engine.prepare(batch.plan_batch(
                opened.all_times, lengths, batch_size,
                tables=(opened.times, opened.sizes[:, 1])))
            _stamp('open.plan', turn, begin)
            _stamp('open', turn, start)
            return opened, [(cfg.SAMPLE_RATE, range(opened.count), (),
                             session_module.Staged(
                                 staging, where, lengths, dtype), plan)]
        loaded = opened.all_audios()
        _stamp('open.objects', turn, begin)
        begin = time.perf_counter_ns()
        # the samples: straight into one of the session's pinned buffers
        with torch.cuda.device(session.engine.device):
            opened.read_all(session.file_buffer(turn, opened.audio_bytes()))
        _stamp('open.read', turn, begin)
        begin = time.perf_counter_ns()
        groups = []
        rates = sorted({rate for _, rate in loaded})
        # (word-time tables, not alignment objects: nothing per file for the
        # interpreter's collector to trace; `deliver` builds what it asks for)
        alignments = opened.all_times()
        for rate in rates:
            chosen = [i for i, (_, r) in enumerate(loaded) if r == rate]
            picked = [alignments[i] for i in chosen]
            audios = [loaded[i][0] for i in chosen]
            plan = None
            if rate == cfg.SAMPLE_RATE and all(
                    a.dim() == 1 or a.shape[0] == 1 for a in audios):
                plan = engine.prepare(batch.plan_batch(
                    picked, [int(a.shape[-1]) for a in audios], batch_size))
            groups.append((rate, chosen, picked, audios, plan))
        _stamp('open.plan', turn, begin)
        _stamp('open', turn, start)
        return opened, groups

    def write(position, opened, chosen, indices, scores):
        start = time.perf_counter_ns()
        opened.threads = write_threads
        if deliver_batch is not None:
            deliver_batch(opened, chosen, indices, scores)
        else:
            for local, index, item in zip(chosen, indices, scores):
                deliver(index, opened.alignment(local), item)
        _stamp('write', position, start)

    # (a quarter and a half batch at both ends, to ramp the pipeline up and down,
    # was measured and is not faster: 57-58 ms against 54-57 for 4 096 files)
    starts = list(range(0, len(text_files), utterances_per_batch))
    ends = [min(first + utterances_per_batch, len(text_files))
            for first in starts]
    ahead = session_module.FILE_BUFFERS - 2
    # the stages' threads (these two pools and the library's file pool: ours, not
    # the caller's) next to the GPU: what they copy into pinned memory its DMA
    # engine reads
    # - when that node has room for them: the pools are sized from the whole CPU
    # budget (`files.stage_threads`), and packed onto a few near CPUs they would
    # lose more than the far socket costs.  The library's pool is shared by the
    # process: it goes back to every allowed CPU when the call ends.
    near = files.cpus_near(engine.device.index)
    if near is not None and len(near) < open_threads + write_threads + 2:
        near = None
    settle, everywhere = None, None
    if near is not None:
        try:
            everywhere = sorted(os.sched_getaffinity(0))
            files.pool_near(near)
        except (runtime.LibraryError, OSError):
            near = everywhere = None                # (placement is a nicety)
    if near is not None:
        def settle():
            try:
                os.sched_setaffinity(0, near)       # (pid 0: this thread only)
            except OSError:
                pass                                # (a cpuset that changed: stay)
    opener = concurrent.futures.ThreadPoolExecutor(
        ahead, thread_name_prefix='emphases-open', initializer=settle)
    writer = concurrent.futures.ThreadPoolExecutor(
        1, thread_name_prefix='emphases-write', initializer=settle)
    writes, in_flight, failure = [], [], None

    def finish(jobs, drain=2):
        for position, pending, opened, chosen, indices in jobs:
            start = time.perf_counter_ns()
            scores = pending.scores()
            _stamp('scores', position, start)
            writes.append(writer.submit(
                write, position, opened, chosen, indices, scores))
        while len(writes) > drain:          # (errors surface; memory bounded)
            writes.pop(0).result()

    def ask(position):
        return opener.submit(
            open_batch, starts[position], ends[position], position)

    try:
        opening = collections.deque(
            ask(position) for position in range(min(ahead, len(starts))))
        for position, first in enumerate(starts):
            opened, groups = opening.popleft().result()
            if position + ahead < len(starts):
                opening.append(ask(position + ahead))
            start = time.perf_counter_ns()
            jobs = [(position,
                     session.submit_staged(plan, audios)
                     if type(audios) is session_module.Staged else
                     session.submit(picked, audios, rate, batch_size, plan=plan),
                     opened, chosen, range(first + chosen[0], first + chosen[-1] + 1)
                     if type(chosen) is range else [first + i for i in chosen])
                    for rate, chosen, picked, audios, plan in groups]
            _stamp('submit', position, start)
            previous, in_flight = in_flight, jobs
            finish(previous)
    except BaseException as error:      # noqa: BLE001
        failure = error
    # what was submitted is finished and written whatever happened after it
    for jobs in (in_flight,):
        try:
            finish(jobs, drain=0)
        except BaseException as error:      # noqa: BLE001
            failure = failure or error
    for pending_write in writes:
        try:
            pending_write.result()
        except BaseException as error:      # noqa: BLE001
            failure = failure or error
    opener.shutdown(wait=True, cancel_futures=True)
    writer.shutdown(wait=True)
    if everywhere is not None:
        try:
            files.pool_near(everywhere)
        except runtime.LibraryError:
            pass
    if failure is not None:
        raise failure


def from_files_to_files(text_files, audio_files, output_prefixes=None,
                        checkpoint=None, batch_size=None, gpu=None,
                        utterances_per_batch=256, conv_tile=None,
                        precision='f32'):
    """`core.py:115-179`, but the files are processed in ragged batches of
    `utterances_per_batch` instead of one at a time, two batches in flight,
    read, parsed and written by the library's host threads
    (`files_to_scores`).  On several GPUs: `dist.from_files_to_files`.
    `precision`: 'f32' or an opt-in name of `engine.PRECISIONS`; the scores
    of a file are bitwise those of the tensor API at the same precision."""
    from pathlib import Path
    text_files, audio_files = list(text_files), list(audio_files)
    if output_prefixes is None:
        output_prefixes = [Path(file).stem for file in text_files]
    output_prefixes = list(output_prefixes)
    session = get_session(checkpoint, gpu, None, conv_tile, precision)
    files_to_scores(
        text_files, audio_files, session, batch_size, utterances_per_batch,
        deliver_batch=lambda opened, chosen, indices, scores: opened.write(
            chosen, [output_prefixes[i] for i in indices], scores))


###############################################################################
# Inference steps
###############################################################################


def preprocess(alignment, audio, sample_rate=cfg.SAMPLE_RATE, batch_size=None,
               gpu=None, pitch_tracker=None):
    """Convert audio to model input (`core.py:345-418`): yields
    `(features [1, NUM_FEATURES, Fc] on the device, word_bounds int64
    [1, 2, Wc] on the CPU)` per chunk."""
    engine = get_engine(None, gpu)
    audio = resample(audio[:1] if audio.dim() == 2 else audio, sample_rate,
                     gpu=gpu).reshape(-1).to(torch.float32)
    segments = batch.chunk_utterance(alignment, int(audio.shape[0]), batch_size)
    if not segments:
        return
    plan = batch.Plan(segments, [0], [int(audio.shape[0])])
    with torch.cuda.device(engine.device), engine.lock:
        meta = engine.upload(plan)
        features = engine.features(
            audio.to(engine.device).contiguous(), plan, meta,
            tracks=_tracks(engine, plan, [audio], pitch_tracker, gpu)).clone()
    for segment, off, count in zip(segments, plan.frame_off, plan.frames):
        yield (features[None, :, off:off + count],
               torch.from_numpy(segment.bounds)[None])


def infer(features, word_bounds, checkpoint=None):
    """Model inference on one chunk (`core.py:295-332`): features
    [1, C_in, F] on the device, word_bounds [1, 2, W] -> logits [1, 1, W]."""
    model = Model(checkpoint=checkpoint, gpu=features.device.index)
    frame_lengths = torch.tensor([features.shape[-1]])
    word_lengths = torch.tensor([word_bounds.shape[-1]])
    return model(features, frame_lengths, word_bounds, word_lengths)


def postprocess(logits, config=None):
    """`core.py:335-342`"""
    config = config or active_config()
    if config.loss == 'bce':
        return torch.sigmoid(logits)
    return torch.clamp(logits, 0., 1.)


def _packed_plan(frame_lengths, word_bounds, word_lengths):
    """Plan for already-featurised, padded `[B, C, T]` inputs."""
    segments = []
    for index, (frames, words) in enumerate(zip(frame_lengths, word_lengths)):
        bounds = np.asarray(
            word_bounds[index].cpu(), dtype=np.int64)[:, :int(words)]
        segments.append(batch.Segment(
            index, 0, int(words), 0, 0, int(frames), bounds))
    return batch.Plan(
        segments, [0] * len(segments), [0] * len(segments))


def _pack(xs, plan, axis_offsets, counts, ld, device):
    packed = torch.zeros(
        (xs.shape[1], ld), dtype=torch.float32, device=device)
    for index, (off, count) in enumerate(zip(axis_offsets, counts)):
        packed[:, off:off + count] = xs[index, :, :count]
    return packed


def downsample(xs, word_bounds, word_lengths, config=None):
    """Interpolate from frame to word resolution (`core.py:426-469`):
    xs [B, C, T], word_bounds [B, 2, W], word_lengths [B] -> [B, C, Wmax]."""
    config = config or active_config()
    device = runtime.require_gpu(xs.device if xs.is_cuda else None)
    lib = runtime.library()
    lengths = [int(n) for n in word_lengths]
    plan = _packed_plan([xs.shape[2]] * xs.shape[0], word_bounds, lengths)
    engine_module.check_bounds(plan, config.downsample_method)
    with torch.cuda.device(device):
        host, offsets = plan.pack_metadata([])
        meta = torch.from_numpy(host).to(device)
        view = lambda name: meta[  # noqa: E731
            offsets[name][0]:offsets[name][0] + offsets[name][1]]
        packed = _pack(
            xs.to(device, torch.float32), plan, plan.frame_off, plan.frames,
            plan.ld_frames, device)
        out = torch.zeros(
            (xs.shape[1], plan.ld_words), dtype=torch.float32, device=device)
        runtime.check(lib.emph_segment_reduce(
            packed.data_ptr(), plan.ld_frames, view('bounds').data_ptr(),
            out.data_ptr(), plan.ld_words, xs.shape[1],
            view('table').data_ptr(), view('word_segment').data_ptr(),
            plan.ld_words, runtime.REDUCTIONS[config.downsample_method],
            runtime.stream()), 'emph_segment_reduce')
    result = torch.zeros(
        (xs.shape[0], xs.shape[1], max(lengths) if lengths else 0),
        dtype=torch.float32, device=device)
    for index, (off, count) in enumerate(zip(plan.word_off, plan.words)):
        result[index, :, :count] = out[:, off:off + count]
    return result if xs.is_cuda else result.cpu()


def segment(xs, word_bounds, word_lengths):
    """Convert acoustic features to word segments (`core.py:552-586`): xs
    [B, C, T], word_bounds [B, 2, W], word_lengths [B] -> (segments
    [B * W, C, L] zero-padded to the longest word L of the batch, bounds
    int64 [B * W, 2, 1] = (0, frames), lengths int64 [B * W]).  Column j >=
    word_lengths[i] of item i repeats the item's last word, as the reference's
    `min(j, words - 1)` does.  One `emph_gather_columns` launch."""
    device = runtime.require_gpu(xs.device if xs.is_cuda else None)
    items, channels, frames = (int(n) for n in xs.shape)
    width = int(word_bounds.shape[2])
    bounds = np.asarray(word_bounds.cpu(), dtype=np.int64)
    spans = bounds[:, 1] - bounds[:, 0]
    longest = int(spans.max()) if spans.size else 0
    count = items * width
    pieces = np.zeros((count, 4), dtype=np.int64)
    lengths = np.zeros(count, dtype=np.int64)
    for item in range(items):
        words = int(word_lengths[item])
        for column in range(width):
            # (Python's negative index when an item has no word at all)
            chosen = min(column, words - 1) % width
            start, end = bounds[item, 0, chosen], bounds[item, 1, chosen]
            if start < 0 or end > frames or end < start:
                # the reference's slice assignment raises on these too
                raise ValueError(
                    f'word bounds ({start}, {end}) outside the {frames} frames')
            index = item * width + column
            pieces[index] = (item * frames + start, end - start,
                             index * longest, longest)
            lengths[index] = end - start
    result_bounds = torch.zeros((count, 2, 1), dtype=torch.long)
    result_bounds[:, 1, 0] = torch.from_numpy(lengths)
    result_lengths = torch.from_numpy(lengths)
    with torch.cuda.device(device):
        # channel-major with the items side by side: a piece is a run of columns
        packed = xs.to(device, torch.float32).permute(1, 0, 2).reshape(
            channels, items * frames).contiguous()
        out = torch.zeros((channels, max(count * longest, 1)),
                          dtype=torch.float32, device=device)
        if count and longest:
            table = torch.from_numpy(pieces).to(device)
            runtime.check(runtime.library().emph_gather_columns(
                packed.data_ptr(), packed.shape[1], out.data_ptr(),
                out.shape[1], channels, table.data_ptr(), count,
                runtime.stream()), 'emph_gather_columns')
        result = out[:, :count * longest].reshape(
            channels, count, longest).permute(1, 0, 2).contiguous().to(xs.dtype)
    if xs.is_cuda:
        return result, result_bounds.to(device), result_lengths.to(device)
    return result.cpu(), result_bounds, result_lengths


class Model:
    """Callable with the reference's `Model.forward` signature
    (`emphases/model/core.py:39`): `(features [B, C_in, T], frame_lengths [B],
    word_bounds [B, 2, W], word_lengths [B]) -> logits [B, 1, Wmax]`.

    Every batch item is computed with its own zero halo (B=1 semantics); the
    reference's padded-batch conv leaks padding into shorter items
    (`convolution.py:35-37`), which is not reproduced."""

    def __init__(self, checkpoint=None, gpu=None, config=None):
        self.config = config or active_config()
        self.engine = get_engine(checkpoint, gpu, self.config)

    def __call__(self, features, frame_lengths, word_bounds, word_lengths):
        return self.forward(features, frame_lengths, word_bounds, word_lengths)

    def forward(self, features, frame_lengths, word_bounds, word_lengths):
        engine = self.engine
        device = engine.device
        plan = _packed_plan(
            [int(n) for n in frame_lengths], word_bounds,
            [int(n) for n in word_lengths])
        with torch.cuda.device(device), engine.lock:
            packed = _pack(
                features.to(device, torch.float32), plan, plan.frame_off,
                plan.frames, plan.ld_frames, device)
            _, logits = engine.forward(None, plan, features=packed)
            logits = logits.clone()
        width = int(max(int(n) for n in word_lengths))
        result = torch.zeros(
            (features.shape[0], 1, width), dtype=torch.float32, device=device)
        for index, (off, count) in enumerate(zip(plan.word_off, plan.words)):
            result[index, 0, :count] = logits[off:off + count]
        return result if features.is_cuda else result.cpu()


###############################################################################
# Utilities
###############################################################################


@contextlib.contextmanager
def inference_context(model=None):
    """`core.py:594-610` sets eval mode, no_grad and autocast; the HIP path is
    inference-only and float32, so only no_grad remains."""
    with torch.no_grad():
        yield


def resample(audio, sample_rate, target_rate=cfg.SAMPLE_RATE, gpu=None):
    """`core.py:613-619` (torchaudio's windowed-sinc `Resample`, third-party)
    on the device: `emph_resample`, the one resampler of the package.  audio
    [..., S] float (or int16 = 16-bit PCM) -> float32 [..., S'], on the
    device the input is on (a host tensor comes back on the host)."""
    if int(sample_rate) == int(target_rate):
        return audio
    session = get_session(None, gpu if gpu is not None or not audio.is_cuda
                          else audio.device.index)
    rows = audio.reshape(-1, audio.shape[-1])
    out = session.resample(
        list(rows), sample_rate, target_rate, on_device=audio.is_cuda)
    return torch.stack(out).reshape(audio.shape[:-1] + (out[0].shape[0],))
