"""Benchmark of the prominence-inference hot path on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path — packed audio already resident in HBM ->
log-mel -> conv frame encoder -> word-boundary reduce -> word decoder ->
scores — over one ragged batch of BASELINE.json's configs[1]: 64 synthetic
10 s / 16 kHz utterances with random alignments, conv config, bundled
checkpoint, per GPU (weak scaling: every rank owns its own 64 utterances and
the only exchange is ONE RCCL all_gather of all steps' per-word scores at the
end of a timed region).

Protocol.  The chip's clocks ramp for tens of milliseconds, so a pre-roll
replays the step until three consecutive 50-step blocks agree to 2 % (at least
0.3 s, at most 3 s); then `--warmup` untimed steps; then `--steps` steps are
timed `--regions` (9) times, each region bracketed by a barrier +
`torch.cuda.synchronize()` on both sides (max over ranks), and the MEDIAN
region is the one reported (`ms_per_step`, `value`), with the fastest and the
slowest beside it.

Prints ONE JSON line on rank 0 with BASELINE.json's metric (utterances/s,
whole job), the roofline of the dominant kernel (fp32-MFMA conv1d, measured
live with HIP events on the launch stream, the committed rocprofv3 average
beside it), the CPU oracle timed on the host cores (threads AND one process
per core), and — on one GPU — side records for BASELINE configs[2]
(Transformer), configs[3] (10 k-utterance corpus) and configs[4] (5-minute
utterances chunked at batch_size 3000) and the public API end to end.

`--workload corpus | longform` is the STRONG-scaling mode for BASELINE
configs[3] / configs[4]: the whole job (10 000 utterances of 2-30 s, or 64
five-minute utterances chunked at batch_size 3000) is sharded over the ranks
by `dist.assign` (LPT by frames), every rank keeps the audio of ITS shard
resident, and one step = one whole job = `dist.exchange_counts` (collective 1)
-> the shard's forward (hipGraph replay) -> `dist.exchange_scores` (collective
2: every rank ends with all scores in input order), both collectives inside
the timed region.  With N > 1 ranks the default line carries both modes as
side records (`configs_3_corpus_sharded`, `configs_4_longform_sharded`).
`--gpus N` without a torchrun environment starts the N workers itself.
"""
import argparse
import csv
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import emphases_amd  # noqa: E402
from emphases_amd import batch, config as cfg, synth  # noqa: E402

UTTERANCES = 64
FRAMES = 1000                 # 10 s at 100 frames/s
PEAK_FP32_MFMA = 157.3        # TFLOP/s, MI355X_MICROARCH.md (dense, f32 in)
PEAK_HBM = 8.0e12             # B/s, same guide
# compulsory traffic and algorithmic flops of the conv path
BYTES_PER_FRAME = 641.
FLOPS_PER_FRAME, FLOPS_PER_WORD = 0.2968e6, 0.2309e6
PROFILE_TAGS = ('r6', 'r5', 'r4', 'r3', 'r2', 'r1')
# kernel name in `Engine.timers` -> substring of the rocprofv3 kernel name
ROCPROF_NAMES = {
    'conv1d_winograd4_frames_80x80_k3': 'conv1d_winograd4_kernel',
    'conv1d_stack_frames_80x80_k3': 'conv1d_stack_kernel',
    'conv1d_split_frames_80x80_k3': 'conv1d_split_kernel',
    'attention_frames': 'attention_group_kernel',
    'frontend_logmel': 'frontend_kernel',
    'segment_reduce': 'segment_reduce_kernel',
    'word_decoder': 'word_decoder_kernel'}
# ... under an opt-in precision
SPLIT_ROCPROF_NAMES = {'attention_frames': 'attention_split_kernel'}


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=200)
    parser.add_argument('--warmup', type=int, default=20)
    parser.add_argument('--regions', type=int, default=9,
                        help='how often the --steps steps are timed; the '
                             'median region is reported')
    parser.add_argument('--config', default='conv',
                        choices=['conv', 'transformer'])
    parser.add_argument('--precision', default='f32',
                        choices=['f32', 'bf16x3', 'bf16x3_fast', 'bf16x6'],
                        help='the opt-in split-bf16 precisions of '
                             'engine.Engine (conv stack / attention); the '
                             'headline is f32')
    parser.add_argument('--workload', default='batch',
                        choices=['batch', 'corpus', 'longform'],
                        help='batch: BASELINE configs[1] per GPU (weak '
                             'scaling, the headline); corpus / longform: '
                             'configs[3] / configs[4] sharded over the ranks '
                             '(strong scaling, both collectives timed)')
    parser.add_argument('--corpus-utterances', type=int, default=10000)
    parser.add_argument('--longform-utterances', type=int, default=64)
    parser.add_argument('--tile', type=int, default=None)
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-api', action='store_true',
                        help='skip the end-to-end public-API measurement')
    parser.add_argument('--no-side', action='store_true',
                        help='skip the side records (configs[2], [3], [4])')
    parser.add_argument('--streams', type=int, default=2,
                        help='batches in flight: consecutive steps alternate '
                             'between this many HIP streams (each with its '
                             'own workspace), so that the VALU-bound front-end '
                             'of one batch overlaps the MFMA-bound encoder of '
                             'the previous one')
    parser.add_argument('--no-winograd', action='store_true',
                        help='direct (3-tap) form of the frame-rate convs '
                             'instead of Winograd')
    parser.add_argument('--backend', default='nccl',
                        help="torch.distributed backend ('nccl' = RCCL; "
                             "'gloo' only to rehearse the N > 1 path on a box "
                             'with fewer GPUs than ranks)')
    parser.add_argument('--no-graph', action='store_true',
                        help='launch kernel by kernel instead of replaying '
                             'the captured HIP graph')
    parser.add_argument('--no-preroll', action='store_true')
    parser.add_argument('--side-records', default=None, metavar='PATH',
                        help='where the FULL record goes (every side '
                             'measurement, notes, per-kernel tables); the '
                             'line on stdout stays under 10 KB and names this '
                             'file and its sha256.  Default: bench_side.json '
                             'next to this script (bench_side_<workload / '
                             'config / precision>.json for the non-default '
                             'commands)')
    parser.add_argument('--cpu-worker', type=float, default=None,
                        help=argparse.SUPPRESS)
    return parser.parse_args()


def workload(rank, count=UTTERANCES, frames=FRAMES):
    """Synthetic utterances `rank*count ..`."""
    first = rank * count
    audios = [synth.audio(first + i, frames) for i in range(count)]
    bounds = [synth.word_frames(first + i, frames) for i in range(count)]
    alignments = [
        emphases_amd.Alignment.from_frames(b, synth.word_names(b.shape[1]))
        for b in bounds]
    return audios, alignments, bounds


def build_plan(audios, alignments):
    lengths = [a.shape[1] for a in audios]
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    segments = []
    for index, (alignment, length) in enumerate(zip(alignments, lengths)):
        segments.extend(batch.chunk_utterance(alignment, length, None, index))
    return batch.Plan(segments, offsets, lengths)


###############################################################################
# CPU baseline (runs BEFORE this process touches the GPU)
###############################################################################


def cpu_model():
    try:
        with open('/proc/cpuinfo') as file:
            for line in file:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def physical_cores():
    """Physical cores this process may run on: distinct (physical id, core id)
    pairs of /proc/cpuinfo among the CPUs of the affinity mask."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = None
    cores = set()
    try:
        processor = physical = core = None
        with open('/proc/cpuinfo') as file:
            for line in file:
                if line.startswith('processor'):
                    processor = int(line.split(':')[1])
                elif line.startswith('physical id'):
                    physical = line.split(':')[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':')[1].strip()
                elif not line.strip():
                    if physical is not None and core is not None and (
                            allowed is None or processor in allowed):
                        cores.add((physical, core))
                    processor = physical = core = None
    except OSError:
        pass
    fallback = len(allowed) if allowed else (os.cpu_count() or 1)
    return len(cores) or fallback


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1
    cfs quota), or None when unlimited: 128 visible cores under a quota of a
    few CPUs behave like those few."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as file:
            quota, period = file.read().split()
        return None if quota == 'max' else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as file:
            quota = float(file.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as file:
            period = float(file.read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def oracle_inputs(count=8):
    from emphases_amd import weights
    state = {k: torch.from_numpy(v) for k, v in weights.load().items()}
    audios = [torch.from_numpy(synth.audio(i, FRAMES)) for i in range(count)]
    words = [[(int(s) / 100., int(e) / 100.) for s, e in
              synth.word_frames(i, FRAMES).T] for i in range(count)]
    return state, audios, words


def cpu_worker(seconds):
    """One child of the process-parallel baseline: one torch thread, the
    oracle on its own utterances, B=1, one after the other
    (`emphases/core.py:169-179`).  Never touches the GPU.  Protocol on
    stdin/stdout: 'ready' when warm, waits for 'go', prints its count."""
    torch.set_num_threads(1)
    from oracle import prominence as oracle
    state, audios, words = oracle_inputs(4)
    oracle.from_alignment_and_audio(words[0], audios[0], state)
    print('ready', flush=True)
    sys.stdin.readline()
    done = 0
    start = time.perf_counter()
    while time.perf_counter() - start < seconds:
        index = done % len(audios)
        oracle.from_alignment_and_audio(words[index], audios[index], state)
        done += 1
    print(json.dumps({'done': done, 'seconds': time.perf_counter() - start}),
          flush=True)


def cpu_processes(count, seconds):
    """`count` fresh single-thread processes running the oracle side by side
    for `seconds`; utterances/s summed over them."""
    env = dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1',
               HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='',
               ROCR_VISIBLE_DEVICES='')
    children = [subprocess.Popen(
        [sys.executable, os.path.abspath(__file__), '--cpu-worker',
         str(seconds)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
        stderr=subprocess.DEVNULL, env=env, text=True, cwd=ROOT)
        for _ in range(count)]
    try:
        for child in children:
            line = child.stdout.readline()
            if line.strip() != 'ready':
                raise RuntimeError(f'cpu worker said {line!r}')
        start = time.perf_counter()
        for child in children:
            child.stdin.write('go\n')
            child.stdin.flush()
        results = [json.loads(child.stdout.readline()) for child in children]
        wall = time.perf_counter() - start
    finally:
        for child in children:
            try:
                child.stdin.close()
                child.wait(timeout=30)
            except Exception:       # noqa: BLE001
                child.kill()
    return {
        'processes': count, 'threads_per_process': 1,
        'value': sum(r['done'] / r['seconds'] for r in results),
        'utterances': sum(r['done'] for r in results),
        'seconds': wall}


def cpu_baseline(seconds=4.0):
    """The CPU oracle (port of the reference's op sequence, B=1 loop exactly
    like `emphases/core.py:169-179`) on this box's host cores:
    in this process at 1 and at 8 torch threads (what the survey timed the
    reference itself with), and as P single-thread PROCESSES for P = 8, 32 and
    all physical cores (P = 8, the cgroup CPU quota and twice the quota when
    the container has one: 128 visible cores under a 16-CPU quota collapse) — the reference's loop is sequential, so a corpus is
    spread over a host by running it once per core, and that is the number the
    GPU rate stands next to."""
    from oracle import prominence as oracle
    state, audios, words = oracle_inputs()
    physical = physical_cores()
    runs = []
    for threads in sorted({1, min(8, physical)}):
        torch.set_num_threads(threads)
        oracle.from_alignment_and_audio(words[0], audios[0], state)  # warm up
        done = 0
        start = time.perf_counter()
        while time.perf_counter() - start < seconds:
            index = done % len(audios)
            oracle.from_alignment_and_audio(
                words[index], audios[index], state)
            done += 1
        elapsed = time.perf_counter() - start
        runs.append({'threads': threads, 'value': done / elapsed,
                     'utterances': done, 'seconds': elapsed})
    torch.set_num_threads(min(8, physical))
    pools = []
    quota = cpu_quota()
    if quota:       # a cgroup CPU quota, not the visible cores, is the capacity
        counts = {min(8, physical), min(physical, max(1, round(quota))),
                  min(physical, max(1, round(2 * quota)))}
    else:
        counts = {min(8, physical), min(32, physical), physical}
    for count in sorted(counts):
        try:
            pools.append(cpu_processes(count, seconds))
        except Exception as error:      # noqa: BLE001
            pools.append({'processes': count, 'error': repr(error)})
    best_pool = max((p for p in pools if 'value' in p),
                    key=lambda p: p['value'], default=None)
    best_run = max(runs, key=lambda run: run['value'])
    if best_pool is not None and best_pool['value'] >= best_run['value']:
        value, cores = best_pool['value'], best_pool['processes']
        sample = (
            f"{best_pool['utterances']} x 10 s utterances in "
            f"{best_pool['seconds']:.1f} s by {cores} single-thread processes "
            f'side by side ({physical} physical cores visible, cgroup CPU '
            f'quota {quota}), each looping oracle/prominence.py one utterance '
            'at a time (B=1), torch CPU fp32')
    else:
        value, cores = best_run['value'], best_run['threads']
        sample = (
            f"{best_run['utterances']} x 10 s utterances in "
            f"{best_run['seconds']:.1f} s at {cores} torch threads, one at a "
            'time (B=1) through oracle/prominence.py, torch CPU fp32')
    return {
        'value': value, 'unit': 'utterances/s', 'cores': cores,
        'kind': 'port', 'sample': sample,
        'threads': {str(run['threads']): run['value'] for run in runs},
        'processes': {str(p['processes']): p.get('value', p.get('error'))
                      for p in pools},
        'physical_cores': physical, 'logical_cores': os.cpu_count(),
        'cgroup_cpu_quota': quota, 'cpu': cpu_model(),
        # BASELINE.md / SURVEY.md §6: the reference itself (bf16 autocast as
        # shipped), measured in the survey container (8 vCPU Xeon @2.1 GHz)
        'reference_survey': {
            'ms_per_utterance_1_thread': 18.8,
            'ms_per_utterance_8_threads': 10.4,
            'utterances_per_s_1_thread': 1000. / 18.8,
            'utterances_per_s_8_threads': 1000. / 10.4,
            'where': 'survey container, 8 vCPU Xeon 2.1 GHz, reference '
                     'from_alignment_and_audio as shipped'}}


###############################################################################
# The timed protocol
###############################################################################


class Runner:
    """`streams` lanes (HIP stream + engine workspace + captured graph) over
    one resident batch; `step()` enqueues one pass on the next lane."""

    def __init__(self, config, state, device, audios, alignments, streams=2,
                 tile=None, winograd=True, graph=True, plan=None, packed=None,
                 precision='f32'):
        """`audios` + `alignments` (configs[1]: one chunk per utterance), or a
        ready `plan` with its `packed` device audio (the sharded workloads)."""
        self.device = device
        self.engine = emphases_amd.engine.Engine(
            config, state, device, conv_tile=tile, winograd=winograd,
            precision=precision)
        if plan is None:
            plan = build_plan(audios, alignments)
            packed = torch.cat(
                [torch.from_numpy(a).reshape(-1) for a in audios]).to(device)
        self.plan, self.packed = plan, packed
        self.meta = self.engine.upload(self.plan)
        self.columns = torch.from_numpy(self.plan.word_columns()).to(device)
        self.lanes = []
        for index in range(max(1, streams)):
            engine = self.engine if index == 0 else \
                emphases_amd.engine.Engine(
                    config, state, device, conv_tile=tile, winograd=winograd,
                    precision=precision)
            stream = torch.cuda.Stream(device=device) if streams > 1 \
                else torch.cuda.current_stream()
            with torch.cuda.stream(stream):
                if graph:
                    replay, scores, _ = engine.capture(
                        self.packed, self.plan, self.meta)
                else:
                    replay, scores = None, None
            self.lanes.append((stream, engine, replay, scores))
        torch.cuda.synchronize()
        self.counter = 0
        self.after = None           # hook(stream, scores, step index)

    def step(self):
        stream, engine, replay, scores = self.lanes[
            self.counter % len(self.lanes)]
        index = self.counter
        self.counter += 1
        with torch.cuda.stream(stream):
            if replay is None:
                scores = engine.forward(self.packed, self.plan, self.meta)[0]
            else:
                replay()
            if self.after is not None:
                self.after(scores, index)
        return scores

    def kernel_times(self, passes=20):
        """Per-kernel durations of eager launches on the launch stream:
        `kernels[name]` = [regions, seconds by recorded HIP events (dispatch
        included), flops, seconds kernel-exact, kernels launched].  Kernel-exact
        = `runtime.LaunchTimer`: events bound to each kernel's own dispatch
        packet, the begin -> end rocprofv3 reads; a named region may hold more
        than one kernel (attention over long and short segments, say)."""
        from emphases_amd import runtime
        engine = self.engine
        engine.timers = []
        # (one untimed pass: every pass below finds its buffers in place)
        engine.forward(self.packed, self.plan, self.meta)
        per_pass = len(engine.timers)
        graph = self.lanes[0][2] is not None

        def hot():
            """The clocks where the timed regions had them: an idle chip runs the
            first passes 7 % slower (tools/timer_check.py: eager passes right behind
            300 replays agree with rocprofv3's trace of the replays to 1-2 %)."""
            begin = time.perf_counter()
            for count in range(300):
                self.step()
                if count >= 20 and count % 10 == 0:
                    torch.cuda.synchronize()
                    if time.perf_counter() - begin > 0.3:
                        break
            torch.cuda.synchronize()
        # creating the timer's events takes a moment: before the pre-roll, so that
        # nothing but the passes follows it (replays are not launches of the
        # library; without graphs the pre-roll would use the timer up)
        capacity = 8 * (passes * max(per_pass, 1) + 4 * passes) + 64
        if not graph:
            hot()
        with runtime.LaunchTimer(capacity) as exact:
            if graph:
                hot()
            engine.timers = []
            for _ in range(passes):
                engine.forward(self.packed, self.plan, self.meta)
            # the empty kernel: what a recorded pair measures around nothing
            for _ in range(4 * passes):
                with engine._timed('launch_probe'):
                    runtime.check(
                        engine.lib.emph_launch_probe(runtime.stream()),
                        'emph_launch_probe')
            torch.cuda.synchronize()
        assert exact.launches <= exact.capacity, (exact.launches, capacity)
        kernels, samples = {}, {}
        for position, (name, flops, begin, end, first, last) in \
                enumerate(engine.timers):
            entry = kernels.setdefault(name, [0, 0., 0., 0., 0])
            entry[0] += 1
            entry[1] += begin.elapsed_time(end) * 1e-3
            entry[2] += flops
            entry[4] += last - first
            # the k-th region of a pass, over the passes: their MEDIAN counts (a
            # region that met a clock step or another process's burst does not)
            slot = position % per_pass if name != 'launch_probe' else 0
            samples.setdefault((name, slot), []).append(
                float(exact.microseconds[first:last].sum()) * 1e-6)
        for (name, slot), values in samples.items():
            kernels[name][3] += float(np.median(values)) * len(values)
        engine.timers = None
        return kernels, passes


def preroll(step, block=50, least=0.3, most=3.0, tolerance=0.02):
    """Replay until the step time is steady: three consecutive `block`-step
    means within `tolerance` of each other."""
    history = []
    begin = time.perf_counter()
    while True:
        start = time.perf_counter()
        for _ in range(block):
            step()
        torch.cuda.synchronize()
        now = time.perf_counter()
        history.append((now - start) / block)
        last = history[-3:]
        steady = len(last) == 3 and max(last) <= min(last) * (1 + tolerance)
        if (steady and now - begin >= least) or now - begin >= most:
            return {'seconds': now - begin, 'blocks': len(history),
                    'steady': bool(steady),
                    'ms_per_step_first_block': history[0] * 1e3,
                    'ms_per_step_last_block': history[-1] * 1e3}


def timed_regions(runner, steps, warmup, regions, world, exchange=None,
                  skip_preroll=False):
    """[seconds] of `regions` timed regions of exactly `steps` steps."""
    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    ramp = None if skip_preroll else preroll(runner.step)
    for _ in range(warmup):
        runner.step()
    if exchange is not None:
        exchange()
    laps = []
    for _ in range(max(1, regions)):
        runner.counter = 0
        barrier()
        start = time.perf_counter()
        for _ in range(steps):
            scores = runner.step()
        if exchange is not None:
            exchange()
        barrier()
        elapsed = time.perf_counter() - start
        if world > 1:
            slowest = torch.tensor(
                [elapsed], dtype=torch.float64, device=runner.device)
            torch.distributed.all_reduce(
                slowest, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(slowest.item())
        laps.append(elapsed)
    return laps, ramp, scores


def profile_file(name):
    for tag in PROFILE_TAGS:
        path = os.path.join(ROOT, 'profiles', f'{tag}_{name}')
        if os.path.exists(path):
            return path
    return None


def stats_file(config, precision='f32', streams=1):
    """The committed rocprofv3 kernel statistics of this command."""
    stem = 'bench' if config == 'conv' else 'transformer'
    if precision != 'f32':
        stem = f'{config}_{precision}'
    return profile_file(f'{stem}_kernel_stats_{streams}stream.csv')


def from_profiles(config, dominant, precision='f32'):
    """What the committed rocprofv3 runs of this same command (same
    `--config`, same `--precision`) say about the dominant kernel - NOT
    measured in this run (`profiles/README.md`)."""
    result = {}
    stats = stats_file(config, precision)
    pattern = ROCPROF_NAMES.get(dominant)
    if precision != 'f32':
        pattern = SPLIT_ROCPROF_NAMES.get(dominant, pattern)
    if stats and pattern:
        # (every instantiation of the kernel: the fused conv launches are
        # conv1d_stack_kernel<false> twice and <true> once per step)
        calls, total = 0, 0.
        with open(stats) as file:
            for row in csv.DictReader(file):
                if pattern in row['Name']:
                    calls += int(row['Calls'])
                    total += float(row['TotalDurationNs'])
        if calls:
            result['rocprof_avg_launch_us'] = total / calls * 1e-3
            result['rocprof_calls'] = calls
            result['kernel_stats_file'] = os.path.relpath(stats, ROOT)
    key = dominant if precision == 'f32' else f'{dominant}@{precision}'
    summary = profile_file('pmc_summary.json' if config == 'conv'
                           else 'transformer_pmc_summary.json')
    if summary:
        with open(summary) as file:
            entry = json.load(file).get(key, {})
        if entry:
            result['traffic'] = entry.get('traffic_bytes')
            result['traffic_raw'] = entry.get('traffic_bytes_raw')
            result['algorithmic_bytes'] = entry.get('algorithmic_bytes')
            result['pmc_file'] = os.path.relpath(summary, ROOT)
    utilisation = profile_file('pmc_utilisation.json')
    if utilisation:
        with open(utilisation) as file:
            entry = json.load(file).get(key, {})
        if entry:
            result['mfma_pipe_busy'] = entry.get('mfma_pipe_busy')
            result['sq_insts_mfma_per_launch'] = \
                entry.get('sq_insts_mfma_per_launch')
            result['mfma_pipe_busy_file'] = os.path.relpath(utilisation, ROOT)
            if result.get('traffic') is None and entry.get('traffic_bytes'):
                result['traffic'] = entry['traffic_bytes']
                result['pmc_file'] = os.path.relpath(utilisation, ROOT)
    return result


def rocprof_kernels(config):
    """Per-kernel rocprofv3 averages (us per launch x launches per step) of the
    committed one-stream and two-stream profiles of this command: what the
    kernels take without the HIP-event dispatch gap, alone and with two
    batches in flight."""
    result = {}
    for streams in (1, 2):
        path = stats_file(config, 'f32', streams)
        if not path:
            continue
        rows = {}
        with open(path) as file:
            for row in csv.DictReader(file):
                if 'emph::' not in row['Name']:
                    continue
                name = row['Name'].split('emph::')[1].split('(')[0]
                rows[name] = {'avg_us': float(row['AverageNs']) * 1e-3,
                              'calls': int(row['Calls'])}
        fewest = min((r['calls'] for r in rows.values()), default=1)
        result[f'{streams}_stream'] = {
            'file': os.path.relpath(path, ROOT),
            'us_per_step': {
                name: r['avg_us'] * r['calls'] / fewest
                for name, r in rows.items()}}
    return result


# flops of one wave-level matrix instruction of each kernel's K loop
MFMA_FLOPS = {'conv1d_stack_frames_80x80_k3': 2 * 16 * 16 * 4,      # v_mfma_f32_16x16x4_f32
              'conv1d_winograd4_frames_80x80_k3': 2 * 16 * 16 * 4,
              'attention_frames': 2 * 16 * 16 * 4}


def executed_matrix_flops(dominant, launches_per_step, committed, spans=None,
                          layers=None):
    """Matrix flops ONE launch of the dominant kernel executes: the committed
    PMC pass's SQ_INSTS_MFMA per launch (`profiles/*_pmc_utilisation.json`, the
    same command as this run) times the flops of the instruction; for the fused
    conv stack cross-checked against the count the span table implies (a
    workgroup per span, 20 k-steps x 30 MFMAs on each of 4 SIMDs per layer)."""
    per = MFMA_FLOPS.get(dominant)
    counted = committed.get('sq_insts_mfma_per_launch')
    result = {}
    if spans and layers and 'stack' in dominant:
        result['mfma_instructions_per_launch_from_spans'] = \
            spans * layers * 20 * 30 * 4 / launches_per_step
    implied = result.get('mfma_instructions_per_launch_from_spans')
    if per and counted and (not implied or abs(counted / implied - 1.) < .02):
        result['mfma_instructions_per_launch'] = counted
        result['mfma_instructions_source'] = committed.get('mfma_pipe_busy_file')
        result['flops'] = counted * per
        if implied:
            result['pmc_vs_span_table'] = counted / implied
    elif per and result:
        # (the committed PMC pass is of the headline batch, this is another job)
        result['flops'] = result['mfma_instructions_per_launch_from_spans'] * per
        result['mfma_instructions_source'] = 'span table (no committed PMC pass)'
    return result


PEAK_BF16_MFMA = 2500.        # TFLOP/s dense, same guide
# flops of one v_mfma_f32_32x32x16_bf16 (a v_mfma_f32_16x16x32_bf16 is half of it)
SPLIT_MFMA_FLOPS = 2 * 32 * 32 * 16
# (products of the scores, of the values) per term of each opt-in precision's attention
SPLIT_TERMS = {'bf16x3': (6, 3), 'bf16x3_fast': (6, 3), 'bf16x6': (6, 6)}


def split_attention_flops_per_instruction(precision):
    """attention_split_kernel per 32 x 32 scores: 3 T_K + 2 T_V instructions of 32 x 32 x
    16 (S^T over a head dimension of 40 -> 48; rows 0 .. 31 of O^T) and 2 T_V of 16 x 16
    x 32 (rows 32 .. 47): the mean flops of an instruction SQ_INSTS_MFMA counts."""
    scores, values = SPLIT_TERMS[precision]
    long_ones, short_ones = 3 * scores + 2 * values, 2 * values
    return SPLIT_MFMA_FLOPS * (long_ones + .5 * short_ones) / (long_ones + short_ones)


# Without a committed PMC count: executed over algorithmic flops of the split attention,
# (3 T_K + 3 T_V) MFMA-equivalents of 32768 flops per 2 x 2 x 32 x 32 x 40
SPLIT_EXECUTED = {name: (3 * scores + 3 * values) * 32768 / 163840.
                  for name, (scores, values) in SPLIT_TERMS.items()}

# What the fields of `roofline` are (kept out of the line: the side-records
# file carries this once).
ROOFLINE_NOTES = {
    'frac': 'EXECUTED matrix flops (MFMA instructions counted by PMC in the '
            'committed pass of this command x flops per instruction) per '
            'second of kernel time over the dense MFMA peak of the pipe the '
            'kernel runs on (fp32: 157.3 TF; the opt-in split kernels: bf16, '
            '2 500 TF). At most 1. The vector work of an f32 kernel runs on '
            'the same multipliers and is not counted (profiles/r5_coexec.txt).',
    'avg_launch_us': 'this run, kernel-exact: events bound to each kernel\'s '
                     'own dispatch packet (emph_launch_timer_*, '
                     'hipExtLaunchKernel) on the launch stream, eager '
                     'launches, one batch in flight - the begin -> end '
                     'rocprofv3 reads, so `rocprof_avg_launch_us` (committed '
                     'trace) must agree',
    'avg_launch_us_events': 'the same launches between two RECORDED HIP '
                            'events: includes the command processor\'s '
                            'dispatch (probe_events_us around an empty kernel '
                            'whose own duration is probe_kernel_us)',
    'useful_tflops': 'algorithmic flops (SURVEY 8d: direct form, fp32 '
                     'problem; no Winograd saving, no split products, no '
                     'padding) per second of kernel time; frac_algorithmic = '
                     'that over the fp32 MFMA peak: above 1 is work the '
                     'kernel avoided or a faster pipe, not a faster fp32 pipe',
    'frac_pmc_pipe_busy': 'SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) '
                          'of the committed PMC pass: the pipe\'s share of '
                          'cycles, independent of the clock',
    'traffic': 'HBM bytes per launch by PMC (separate FETCH_SIZE / '
               'WRITE_SIZE passes, FETCH doubled per the guide), committed '
               'profile of this command'}


def roofline(kernels, passes, ms_per_step, config, spans=None, layers=None,
             precision='f32'):
    """The dominant kernel against the matrix pipe it runs on.  `frac` =
    matrix flops it EXECUTES (MFMA instructions counted by PMC) per second of
    kernel time (kernel-exact, this run) over the dense peak; `useful_tflops`
    / `frac_algorithmic` = the fp32 problem's direct-form flops per second,
    which says how fast the layer is computed and can exceed the fp32 peak."""
    probe = kernels.pop('launch_probe', None)
    exact = all(value[3] > 0. for value in kernels.values())
    clock = 3 if exact else 1
    dominant = max(kernels, key=lambda name: kernels[name][clock])
    regions, by_events, flops, by_kernel, launched = kernels[dominant]
    committed = from_profiles(config, dominant, precision)
    launches_per_step = regions / passes
    events_us = by_events / regions * 1e6
    kernel_us = by_kernel / regions * 1e6 if exact else events_us
    algorithmic_flops = flops / regions
    split = precision != 'f32' and (dominant.startswith('attention') or
                                    dominant.startswith('conv1d_split'))
    peak = PEAK_BF16_MFMA if split else PEAK_FP32_MFMA
    counted = committed.get('sq_insts_mfma_per_launch')
    if split:
        executed = {}
        if counted:
            per_instruction = split_attention_flops_per_instruction(precision) \
                if dominant.startswith('attention') else SPLIT_MFMA_FLOPS
            executed = {'flops': counted * per_instruction,
                        'mfma_instructions_per_launch': counted,
                        'mfma_instructions_source':
                            committed.get('mfma_pipe_busy_file')}
        elif dominant.startswith('attention'):
            executed = {'flops': algorithmic_flops * SPLIT_EXECUTED[precision],
                        'mfma_instructions_source':
                            'formula (no committed PMC pass): 27 / 36 MFMA-'
                            'equivalents per 32 x 32 scores, tile edges not '
                            'counted'}
        else:
            # direct form, three products per term, 96 rows for 80 channels,
            # 256 computed positions per 250 owned
            executed = {'flops': algorithmic_flops * 3. * 96. / 80. * 256. / 250.,
                        'mfma_instructions_source':
                            'formula (no committed PMC pass): direct-form '
                            'flops x 3 products x 96 / 80 rows x 256 / 250'}
    else:
        executed = executed_matrix_flops(
            dominant, launches_per_step, committed, spans, layers)
    executed_flops = executed.get('flops') or algorithmic_flops
    achieved = executed_flops / (kernel_us * 1e-6) / 1e12
    useful = algorithmic_flops / (kernel_us * 1e-6) / 1e12
    result = {
        'bound': 'mfma', 'kernel': dominant,
        'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
        'frac': achieved / peak,
        # HBM bytes per launch by PMC: the committed profile of this command
        'traffic': committed.get('traffic'),
        'traffic_source': committed.get('pmc_file'),
        'avg_launch_us': kernel_us,
        'avg_launch_us_source': 'kernel-exact events, this run' if exact else
        'recorded events, this run (dispatch included)',
        'avg_launch_us_events': events_us,
        'kernels_per_launch': launched / regions,
        'probe_events_us': probe[1] / probe[0] * 1e6 if probe else None,
        'probe_kernel_us': probe[3] / probe[0] * 1e6 if probe else None,
        'launches_per_step': launches_per_step,
        'share_of_step': None if not ms_per_step else
        kernel_us * 1e-6 * launches_per_step / (ms_per_step * 1e-3),
        'executed_mfma_flops_per_launch': executed_flops,
        'executed': executed,
        'algorithmic_flops_per_launch': algorithmic_flops,
        'useful_tflops': useful,
        'frac_algorithmic': useful / PEAK_FP32_MFMA}
    if committed.get('mfma_pipe_busy') is not None:
        result['frac_pmc_pipe_busy'] = committed['mfma_pipe_busy']
    if committed.get('rocprof_avg_launch_us'):
        average = committed['rocprof_avg_launch_us']
        result['rocprof_avg_launch_us'] = average
        result['rocprof_file'] = committed.get('kernel_stats_file')
        result['frac_at_rocprof_avg'] = \
            executed_flops / (average * 1e-6) / 1e12 / peak
    return result, committed


def summary(laps, steps):
    laps = sorted(laps)
    median = laps[len(laps) // 2] if len(laps) % 2 else \
        0.5 * (laps[len(laps) // 2 - 1] + laps[len(laps) // 2])
    return {
        'ms_per_step': median / steps * 1e3,
        'ms_per_step_min': laps[0] / steps * 1e3,
        'ms_per_step_max': laps[-1] / steps * 1e3,
        'timed_region_s': median, 'regions': len(laps)}


###############################################################################
# Side records
###############################################################################


def percentiles(laps):
    laps = np.asarray(laps) * 1e3
    return {
        'ms_per_call': float(np.median(laps)),
        'ms_per_call_p90': float(np.percentile(laps, 90)),
        'ms_per_call_p99': float(np.percentile(laps, 99)),
        'ms_per_call_worst': float(laps.max()),
        'worst_lap_index': int(laps.argmax()),
        'ms_per_call_mean': float(laps.mean()), 'laps': int(laps.size)}


def end_to_end_api(audios, alignments, rounds=200):
    """SURVEY.md §8(d) protocol (ii): the public batch API on the same 64
    utterances, host tensors in, scores out - chunk planning, staging, H2D,
    kernels, D2H and the split into per-utterance tensors all inside the
    clock.  `call` = one synchronous `from_alignments_and_audios` after the
    other (median / p90 / p99 / worst of the laps); `pipelined` =
    `Session.submit` with two batches in flight (what `from_files_to_files`
    does).  float32 = the reference's input type (pageable CPU tensors); pcm16
    = the same audio as 16-bit PCM tensors."""
    floats = [torch.from_numpy(a) for a in audios]
    pcm = [torch.from_numpy(np.rint(a * 32768.).astype(np.int16))
           for a in audios]
    session = emphases_amd.get_session(None, 0)
    result = {}
    reference = None
    for name, tensors in (('float32', floats), ('pcm16', pcm)):
        for _ in range(16):         # both lanes: buffers, layout cache, graph
            scores = emphases_amd.from_alignments_and_audios(
                alignments, tensors, 16000)
        torch.cuda.synchronize()
        laps = []
        for _ in range(rounds):
            start = time.perf_counter()
            scores = emphases_amd.from_alignments_and_audios(
                alignments, tensors, 16000)
            laps.append(time.perf_counter() - start)
        # pipelined: five runs of 40 submissions, the median run counts
        runs = []
        for _ in range(5):
            start = time.perf_counter()
            previous = None
            for _ in range(40):
                pending = session.submit(alignments, tensors, 16000)
                if previous is not None:
                    previous.result()
                previous = pending
            scores = previous.result()
            runs.append((time.perf_counter() - start) / 40)
        piped = float(np.median(runs))
        flat = torch.cat([s.reshape(-1) for s in scores])
        if reference is None:
            reference = flat
        entry = percentiles(laps)
        entry.update({
            'utterances_per_s': len(audios) / entry['ms_per_call'] * 1e3,
            'ms_per_call_pipelined': piped * 1e3,
            'ms_per_call_pipelined_runs': [run * 1e3 for run in runs],
            'utterances_per_s_pipelined': len(audios) / piped,
            'bit_identical_to_float32': bool(torch.equal(flat, reference))})
        result[name] = entry
    # every call a layout never seen before (what a stream of real utterances
    # looks like: the layout cache and its captured graph never hit)
    fresh = [[emphases_amd.Alignment.from_frames(
        synth.word_frames(5000 + 64 * k + i, FRAMES),
        synth.word_names(synth.word_frames(5000 + 64 * k + i, FRAMES).shape[1]))
        for i in range(len(audios))] for k in range(24)]
    for group in fresh[:4]:
        emphases_amd.from_alignments_and_audios(group, floats, 16000)
    laps = []
    for group in fresh[4:]:
        start = time.perf_counter()
        emphases_amd.from_alignments_and_audios(group, floats, 16000)
        laps.append(time.perf_counter() - start)
    start = time.perf_counter()
    previous = None
    for group in fresh[4:]:
        pending = session.submit(group, floats, 16000)
        if previous is not None:
            previous.result()
        previous = pending
    previous.result()
    piped = (time.perf_counter() - start) / len(fresh[4:])
    entry = percentiles(laps)
    entry.update({'ms_per_call_pipelined': piped * 1e3,
                  'utterances_per_s_pipelined': len(audios) / piped})
    result['float32_new_layout_every_call'] = entry
    result['workload'] = (
        'emphases_amd.from_alignments_and_audios on 64 x 10 s host tensors '
        '(BASELINE configs[1] through the public API): planning + staging + '
        'H2D + kernels + D2H; pipelined = 2 batches in flight '
        '(session.Session); ms_per_call = median of the laps')
    return result


def side_transformer(device, audios, alignments, args, precision='f32',
                     baseline=None):
    """BASELINE configs[2]: the same 64 x 10 s batch, Transformer config
    (seeded weights), same protocol as the headline.  `precision` 'bf16x3' /
    'bf16x6': the opt-in split-bf16 attention (never the headline), with its
    worst score difference to the f32 run's scores (`baseline`)."""
    config = cfg.Config(architecture='transformer')
    state = emphases_amd.weights.random_state(config, seed=0)
    runner = Runner(config, state, device, audios, alignments, streams=2,
                    precision=precision)
    steps = max(10, min(args.steps, 40))
    laps, ramp, scores = timed_regions(runner, steps, 5, 5, 1)
    line = summary(laps, steps)
    line['precision'] = precision
    kept = scores[runner.columns].clone()
    if baseline is not None:
        line['max_abs_dscore_vs_f32'] = float((kept - baseline).abs().max())
        line['words_compared'] = int(kept.numel())
    line['_scores'] = kept
    kernels, passes = runner.kernel_times(5)
    roof, committed = roofline(
        kernels, passes, line['ms_per_step'], 'transformer',
        precision=precision)
    line.update({
        'workload': '64 synthetic 10 s 16 kHz utterances, Transformer config '
                    '(6 post-LN layers, 2 heads, 80 channels; seeded '
                    'weights), two batches in flight, hipGraph replay '
                    '(BASELINE.json configs[2])',
        'steps': steps, 'utterances_per_s':
            UTTERANCES / line['ms_per_step'] * 1e3,
        'roofline': roof, 'from_profiles': committed,
        'kernels_us_per_step': {
            name: value[3] / passes * 1e6 for name, value in kernels.items()},
        'preroll': ramp})
    del runner
    torch.cuda.empty_cache()
    return line


def side_conv_split(device, audios, alignments, args, baseline):
    """BASELINE configs[1] with the opt-in precision='bf16x3' (the frame-rate
    convs on the bf16 matrix pipe, operands split into two bf16 pieces): same
    protocol as the headline, never the headline; `baseline`: the f32 run's
    scores."""
    runner = Runner(cfg.DEFAULT, None, device, audios, alignments, streams=2,
                    precision='bf16x3')
    steps = max(10, min(args.steps, 40))
    laps, ramp, scores = timed_regions(runner, steps, 5, 5, 1)
    line = summary(laps, steps)
    kept = scores[runner.columns]
    kernels, passes = runner.kernel_times(5)
    kernels.pop('launch_probe', None)
    line.update({
        'precision': 'bf16x3',
        'workload': '64 synthetic 10 s 16 kHz utterances, conv config, the seven '
                    'frame-rate layers as two launches of emph_conv1d_split '
                    '(bf16x3, direct form); two batches in flight, hipGraph '
                    'replay (BASELINE.json configs[1], opt-in precision)',
        'steps': steps,
        'utterances_per_s': UTTERANCES / line['ms_per_step'] * 1e3,
        'max_abs_dscore_vs_f32': float((kept - baseline).abs().max()),
        'words_compared': int(kept.numel()),
        'kernels_us_per_step': {
            name: value[3] / passes * 1e6 for name, value in kernels.items()},
        'preroll': ramp})
    del runner
    torch.cuda.empty_cache()
    return line


def single_utterance_api(audios, alignments, host=None, rounds=300):
    """BASELINE configs[0]'s counterpart (the reference's only mode,
    `emphases/core.py:223-265`): the latency of ONE `from_alignment_and_audio`
    call on a 10 s host tensor - planning, staging, H2D, kernels, D2H - as
    shipped (`default`: kernels chosen by configuration, scores bitwise those
    of any batch) and with `conv_tile='auto'` (lowest-latency kernels for a
    small batch); beside the oracle's one-core latency on this box and the
    survey's figure for the reference itself."""
    audio, alignment = torch.from_numpy(audios[0]), alignments[0]
    result = {'workload': (
        'one emphases_amd.from_alignment_and_audio call on one 10 s 16 kHz '
        'utterance, float32 host tensor in, scores on the host out '
        '(BASELINE.json configs[0] on the GPU)')}

    def clock(call):
        for _ in range(20):
            scores = call()
        torch.cuda.synchronize()
        laps = []
        for _ in range(rounds):
            start = time.perf_counter()
            scores = call()
            laps.append(time.perf_counter() - start)
        laps = np.asarray(laps) * 1e3
        return {'ms_p50': float(np.median(laps)),
                'ms_p90': float(np.percentile(laps, 90)),
                'ms_p99': float(np.percentile(laps, 99)),
                'ms_worst': float(laps.max()), 'laps': rounds,
                'scores': int(scores.numel()),
                'checksum': float(scores.double().sum())}
    result['default'] = clock(lambda: emphases_amd.from_alignment_and_audio(
        alignment, audio, 16000))
    result['conv_tile_auto'] = clock(
        lambda: emphases_amd.from_alignments_and_audios(
            [alignment], [audio], 16000, conv_tile='auto')[0])
    # the kernels alone: the same utterance resident, graph replay
    engine = emphases_amd.get_engine(None, 0)
    with engine.lock:
        seconds, *_ = replay_time(
            engine, audio.reshape(-1).to(engine.device), [alignment],
            [int(audio.shape[-1])], None, least=0.1)
    result['device_only_ms'] = seconds * 1e3
    one_thread = dig(host or {}, 'threads', '1')
    if one_thread:
        result['oracle_1_core_ms'] = 1e3 / one_thread
        result['oracle_1_core_is'] = (
            'oracle/prominence.py (fp32, B=1) on one core of this box: the '
            'cpu_baseline leg at 1 torch thread')
    result['reference_survey_ms_1_thread'] = 18.8
    result['reference_survey_is'] = (
        'the reference as shipped (bf16 autocast), survey container, 8 vCPU '
        'Xeon 2.1 GHz, 1 torch thread (BASELINE.md)')
    return result


def replay_time(engine, packed, alignments, lengths, batch_size, least=0.25,
                keep=None):
    """(seconds per pass, words, frames, checksum, laps) of one ragged batch
    with its audio resident, replayed as a graph.  `keep`: a list that receives
    the batch's scores (a copy)."""
    plan = batch.plan_batch(alignments, lengths, batch_size)
    meta = engine.upload(plan)
    replay, scores, _ = engine.capture(packed, plan, meta)
    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    laps = []
    begin = time.perf_counter()
    while len(laps) < 5 or time.perf_counter() - begin < least:
        start = time.perf_counter()
        replay()
        torch.cuda.synchronize()
        laps.append(time.perf_counter() - start)
        if len(laps) >= 200:
            break
    columns = torch.as_tensor(plan.word_columns(), device=scores.device)
    values = scores[columns]
    assert bool(torch.isfinite(values).all())
    if keep is not None:
        keep.append(values.clone())
    return (float(np.median(laps)), plan.total_words, plan.total_frames,
            float(values.double().sum()), len(laps))


def api_time(alignments, audios, batch_size, rounds=5):
    for _ in range(4):      # both lanes see the layout twice: graphs captured
        emphases_amd.from_alignments_and_audios(
            alignments, audios, 16000, batch_size=batch_size, gpu=0)
    laps = []
    for _ in range(rounds):
        start = time.perf_counter()
        scores = emphases_amd.from_alignments_and_audios(
            alignments, audios, 16000, batch_size=batch_size, gpu=0)
        laps.append(time.perf_counter() - start)
    return float(np.median(laps)), float(
        sum(float(s.double().sum()) for s in scores))


def rates(count, frames, words, seconds):
    frames = int(frames)
    return {
        'ms': seconds * 1e3, 'utterances_per_s': count / seconds,
        'frames_per_s': frames / seconds,
        'realtime_factor': frames / 100. / seconds,
        'mfma_frac': (frames * FLOPS_PER_FRAME + words * FLOPS_PER_WORD)
        / seconds / (PEAK_FP32_MFMA * 1e12),
        'hbm_frac_compulsory': frames * BYTES_PER_FRAME / seconds / PEAK_HBM}


def split_entry(device, packed, alignments, lengths, batch_size, count,
                baseline, precision='bf16x3'):
    """The same resident batch through an engine of the opt-in precision:
    rates, and the worst score difference to the f32 engine's scores."""
    engine = emphases_amd.engine.Engine(
        cfg.DEFAULT, None, device, precision=precision)
    kept = []
    seconds, words, total, checksum, laps = replay_time(
        engine, packed, alignments, lengths, batch_size, keep=kept)
    entry = rates(count, total, words, seconds)
    entry.update({
        'precision': precision, 'checksum': checksum, 'laps': laps,
        'max_abs_dscore_vs_f32': float((kept[0] - baseline).abs().max()),
        'words_compared': int(kept[0].numel()),
        'what': 'device-only graph replay, audio resident, opt-in precision'})
    del engine
    return entry


def side_corpus(device):
    """BASELINE configs[3]: 10 000 utterances of 2-30 s (SURVEY.md §8d:
    F_i ~ U{200..3000}; 16 M frames, 0.48 M words).  One GPU here: rank 0's
    share under the 8-rank LPT assignment of `dist.assign` (what one GPU of
    the node does) device-only and through the public API, and the WHOLE
    corpus as one ragged batch device-only.  Utterance i's audio is a prefix
    of one of 40 distinct 30 s signals (generating 2.6 G distinct samples
    would take minutes); all alignments are distinct."""
    from emphases_amd import dist
    pool = 40
    engine = emphases_amd.engine.Engine(cfg.DEFAULT, None, device)
    host = [torch.from_numpy(synth.audio(7000 + i, 3000)) for i in range(pool)]
    on_device = [a.reshape(-1).to(device) for a in host]
    frames = synth.corpus_frames(10000, 200, 3000)
    alignments = [emphases_amd.Alignment.from_frames(
        synth.word_frames(5000 + i, int(n))) for i, n in enumerate(frames)]
    picks = np.arange(len(frames)) % pool
    shards = dist.assign(dist.cost(frames), 8)
    loads = [int(frames[s].sum()) for s in shards]
    own = shards[0]
    result = {'workload': (
        '10 000 synthetic utterances of 2-30 s (16.0 M frames), conv config '
        '(BASELINE.json configs[3]) on ONE GPU: rank 0 of 8 = its LPT share '
        'of the corpus; whole = all 10 000 as one ragged batch'),
        'shard_frames_min_max': [min(loads), max(loads)]}

    def packed_of(indices):
        lengths = [int(frames[i]) * cfg.HOPSIZE for i in indices]
        return torch.cat([on_device[picks[i]][:n]
                          for i, n in zip(indices, lengths)]), lengths

    packed, lengths = packed_of(own)
    kept = []
    seconds, words, total, checksum, laps = replay_time(
        engine, packed, [alignments[i] for i in own], lengths, None, keep=kept)
    entry = rates(len(own), total, words, seconds)
    entry.update({'utterances': len(own), 'frames': total, 'words': words,
                  'checksum': checksum, 'laps': laps,
                  'what': 'device-only graph replay, audio resident'})
    result['rank0_of_8_device_only'] = entry
    result['rank0_of_8_device_only_bf16x3'] = guarded(
        split_entry, device, packed, [alignments[i] for i in own], lengths,
        None, len(own), kept[0])
    audios = [host[picks[i]][:, :int(frames[i]) * cfg.HOPSIZE] for i in own]
    seconds, api_sum = api_time([alignments[i] for i in own], audios, None)
    entry = rates(len(own), total, words, seconds)
    entry.update({'checksum': api_sum, 'what': (
        'public API: pageable float32 host tensors in, scores out'),
        'pcie_floor_ms_at_55GBps': total * 640 / 55e9 * 1e3})
    result['rank0_of_8_api_float32'] = entry
    del packed
    packed, lengths = packed_of(range(len(frames)))
    seconds, words, total, checksum, laps = replay_time(
        engine, packed, alignments, lengths, None, least=0.4)
    entry = rates(len(frames), total, words, seconds)
    entry.update({'utterances': len(frames), 'frames': total, 'words': words,
                  'checksum': checksum, 'laps': laps,
                  'memory_allocated_GB':
                      torch.cuda.max_memory_allocated() / 1e9,
                  'what': 'whole corpus as ONE ragged batch, device-only'})
    result['whole_corpus_device_only'] = entry
    del packed, engine, on_device
    torch.cuda.empty_cache()
    return result


def side_longform(device):
    """BASELINE configs[4]: 5-minute utterances (30 000 frames) chunked at
    batch_size = 3000 frames, 8 per GPU (64 per node)."""
    engine = emphases_amd.engine.Engine(cfg.DEFAULT, None, device)
    count = 8
    host = [torch.from_numpy(synth.audio(7100 + i, 30000))
            for i in range(count)]
    alignments = [emphases_amd.Alignment.from_frames(
        synth.word_frames(9000 + i, 30000)) for i in range(count)]
    lengths = [int(a.shape[1]) for a in host]
    packed = torch.cat([a.reshape(-1) for a in host]).to(device)
    kept = []
    seconds, words, total, checksum, laps = replay_time(
        engine, packed, alignments, lengths, 3000, keep=kept)
    result = {'workload': (
        '8 synthetic 5-minute utterances (one GPU\'s share of 64 per node), '
        'conv config, chunked at batch_size = 3000 frames '
        '(BASELINE.json configs[4])')}
    entry = rates(count, total, words, seconds)
    entry.update({'frames': total, 'words': words, 'checksum': checksum,
                  'laps': laps, 'chunks': len(batch.plan_batch(
                      alignments, lengths, 3000)),
                  'what': 'device-only graph replay, audio resident'})
    result['device_only'] = entry
    result['device_only_bf16x3'] = guarded(
        split_entry, device, packed, alignments, lengths, 3000, count, kept[0])
    seconds, api_sum = api_time(alignments, host, 3000)
    entry = rates(count, total, words, seconds)
    entry.update({'checksum': api_sum, 'what': (
        'public API: pageable float32 host tensors in, scores out'),
        'pcie_floor_ms_at_55GBps': total * 640 / 55e9 * 1e3})
    result['api_float32'] = entry
    pcm = [torch.from_numpy(np.rint(a.numpy() * 32768.).astype(np.int16))
           for a in host]
    seconds, api_sum = api_time(alignments, pcm, 3000)
    entry = rates(count, total, words, seconds)
    entry.update({'checksum': api_sum,
                  'what': 'public API, 16-bit PCM tensors in'})
    result['api_pcm16'] = entry
    del packed, engine
    torch.cuda.empty_cache()
    return result


def side_files_api(device, count=4096):
    """The drop-in surface itself (`emphases/core.py:115-179`): `count`
    synthetic 10 s utterances as 16-bit PCM .wav + .TextGrid files in
    /dev/shm, `emphases_amd.from_files_to_files` over all of them (read,
    parse, plan, stage, H2D, kernels, D2H, write `<prefix>.TextGrid` + `.pt`),
    files/s; and where a batch's time goes, each phase timed by itself on one
    batch of 256 files (they overlap in the call: two batches in flight)."""
    import shutil
    import tempfile
    from emphases_amd import files, load
    root = '/dev/shm' if os.path.isdir('/dev/shm') else None
    directory = tempfile.mkdtemp(prefix='emph_files_', dir=root)
    try:
        distinct, laps_wanted = 32, 5
        texts, waves, prefixes = [], [], []
        for index in range(count * (laps_wanted + 1)):
            wave = os.path.join(directory, f'a{index % distinct}.wav')
            if index < distinct:
                load.save_wav(wave, synth.audio(index, FRAMES))
            else:       # hard links: distinct files, the same 32 signals
                link = os.path.join(directory, f'a{index}.wav')
                os.link(wave, link)
                wave = link
            text = os.path.join(directory, f'u{index}.TextGrid')
            bounds = synth.word_frames(3000 + index, FRAMES)
            emphases_amd.Alignment.from_frames(
                bounds, synth.word_names(bounds.shape[1])).save(text)
            texts.append(text)
            waves.append(wave)
            prefixes.append(os.path.join(directory, f'out{index}'))
        # warm-up on files of their own, then every lap on files (alignments)
        # nobody has seen: no cached plan, no captured graph - what a corpus is
        # (eight batches of 256: every pinned buffer of the session exists
        # afterwards)
        warm = slice(laps_wanted * count,
                     laps_wanted * count + min(count, 2048))
        emphases_amd.from_files_to_files(
            texts[warm], waves[warm], prefixes[warm], gpu=device.index)
        # (the 24 000 files above are this function's own litter: collected now,
        # not by a full collection of the interpreter inside the first lap)
        import gc
        gc.collect()
        laps = []
        for lap in range(laps_wanted):
            part = slice(lap * count, (lap + 1) * count)
            start = time.perf_counter()
            emphases_amd.from_files_to_files(
                texts[part], waves[part], prefixes[part], gpu=device.index)
            laps.append(time.perf_counter() - start)
        seconds = float(np.median(laps))
        # ... and the last lap's files once more (plans and graphs cached)
        start = time.perf_counter()
        emphases_amd.from_files_to_files(
            texts[part], waves[part], prefixes[part], gpu=device.index)
        again = time.perf_counter() - start
        scores = torch.load(prefixes[count - 1] + '.pt')
        result = {
            'workload': (
                f'{count} synthetic 10 s utterances as 16-bit PCM .wav + '
                '.TextGrid files in /dev/shm through '
                'emphases_amd.from_files_to_files (emphases/core.py:115-179): '
                'read + parse + plan + stage + H2D + kernels + D2H + write '
                '.TextGrid and .pt per file; batches of 256 files, two in '
                'flight; every lap on alignments never seen before'),
            'files': count, 'seconds': seconds, 'laps_s': laps,
            'files_per_s': count / seconds,
            'files_per_s_best_lap': count / min(laps),
            'files_per_s_layouts_seen_before': count / again,
            'realtime_factor': count * 10. / seconds,
            'last_file_scores': int(scores.numel()),
            'reference_loop_files_per_s_python_readers': None}
        # phases of ONE batch of 256 files, each by itself
        some = slice(0, 256)

        def clock(function, rounds=5):
            function()
            times = []
            for _ in range(rounds):
                start = time.perf_counter()
                value = function()
                times.append(time.perf_counter() - start)
            return float(np.median(times)) * 1e3, value
        phases = {}
        phases['open_parse_and_headers_ms'], opened = clock(
            lambda: files.FileBatch(texts[some], waves[some]))
        alignments = [opened.alignment(i) for i in range(256)]
        lengths = [opened.audio(i)[0].shape[0] for i in range(256)]
        pinned = torch.empty(sum(lengths), dtype=torch.int16).pin_memory()
        where = np.concatenate([[0], np.cumsum(lengths)[:-1]]) * 2
        phases['read_samples_into_pinned_ms'], _ = clock(lambda: opened.read(
            list(range(256)), where, np.asarray(lengths) * 2,
            pinned.data_ptr()))
        phases['plan_ms'], plan = clock(
            lambda: batch.plan_batch(alignments, lengths, None))
        on_device = torch.empty_like(pinned, device=device)

        def h2d():
            on_device.copy_(pinned, non_blocking=True)
            torch.cuda.synchronize()
        phases['h2d_ms'], _ = clock(h2d)
        engine = emphases_amd.get_engine(None, device.index)
        with engine.lock:
            meta = engine.upload(plan)
            replay, kept, _ = engine.capture(on_device, plan, meta)

            def kernels():
                replay()
                torch.cuda.synchronize()
            phases['kernels_ms'], _ = clock(kernels)
        rows = [torch.rand(1, len(a)) for a in alignments]
        phases['write_textgrid_and_pt_ms'], _ = clock(lambda: opened.write(
            list(range(256)), prefixes[some], rows))
        # the same batch through the Python readers / writers, one file at a
        # time (what the package did before csrc/files.hip)
        from emphases_amd import alignment as alignment_module

        def python_files():
            for index in range(64):
                item = alignment_module.Alignment(texts[index])
                load.wav(waves[index], raw=True)
                item.save(prefixes[index] + '.TextGrid')
                torch.save(rows[index], prefixes[index] + '.pt')
        per_64, _ = clock(python_files, rounds=3)
        phases['python_readers_and_writers_ms_per_256_files'] = per_64 * 4
        result['reference_loop_files_per_s_python_readers'] = \
            64 / (per_64 * 1e-3)
        result['phases_of_one_256_file_batch'] = phases
        result['host'] = {'cgroup_cpu_quota': cpu_quota(),
                          'threads': files.THREADS}
        return result
    finally:
        shutil.rmtree(directory, ignore_errors=True)


def guarded(function, *args):
    """(the side measurements must never cost the line its headline)"""
    try:
        start = time.perf_counter()
        result = function(*args)
        result['seconds_spent'] = time.perf_counter() - start
        return result
    except Exception as error:       # noqa: BLE001
        return {'error': repr(error)}


###############################################################################
# Strong scaling: BASELINE configs[3] / configs[4] sharded over the ranks
###############################################################################


def job_inputs(workload, args):
    """(frame counts of ALL utterances, batch_size, audio pool size, seeds,
    description) of a whole job - what every rank can compute from nothing."""
    if workload == 'corpus':
        frames = synth.corpus_frames(args.corpus_utterances, 200, 3000)
        return frames, None, 40, (7000, 5000), (
            f'{len(frames)} synthetic utterances of 2-30 s '
            f'({int(frames.sum()) / 1e6:.1f} M frames), conv config '
            '(BASELINE.json configs[3])')
    frames = np.full(args.longform_utterances, 30000, dtype=np.int64)
    return frames, 3000, 8, (7100, 9000), (
        f'{len(frames)} synthetic 5-minute utterances, conv config, chunked '
        'at batch_size = 3000 frames (BASELINE.json configs[4])')


def sharded_job(workload, args, rank, world, device, steps, warmup, regions,
                kernel_passes=3):
    """One step = one WHOLE job, strong-scaled (`/root/reference`'s loop at
    `emphases/core.py:169-179` over all files, here over the ranks):

        dist.assign (LPT by frames; planned once, outside the clock)
        dist.exchange_counts     collective 1, from the plan alone
        the shard's forward      hipGraph replay, audio resident in HBM
        dist.exchange_scores     collective 2: all scores, input order, on
                                 every rank (flat tensor + sizes)

    Utterance i's audio is a prefix of one of a small pool of distinct signals
    (2.6 G distinct samples would take minutes to generate); all alignments
    are distinct.  A rank that fails in its setup still joins collective 1
    with the failure sentinel, so every rank raises instead of hanging."""
    from emphases_amd import dist
    frames, batch_size, pool, (audio_seed, word_seed), description = \
        job_inputs(workload, args)
    shards = dist.assign(dist.cost(frames), world)
    own = [int(i) for i in shards[rank]]
    failure, runner, counts = None, None, []
    try:
        longest = int(frames.max())
        signals = [torch.from_numpy(synth.audio(audio_seed + i, longest))
                   .reshape(-1).to(device) for i in range(pool)]
        alignments = [emphases_amd.Alignment.from_frames(
            synth.word_frames(word_seed + i, int(frames[i]))) for i in own]
        lengths = [int(frames[i]) * cfg.HOPSIZE for i in own]
        packed = torch.cat([signals[i % pool][:n]
                            for i, n in zip(own, lengths)]) if own else \
            torch.zeros(0, device=device)
        del signals
        plan = batch.plan_batch(alignments, lengths, batch_size)
        counts = np.bincount(plan.utterance, weights=plan.words,
                             minlength=len(own)).astype(np.int64)
        if own:
            runner = Runner(cfg.DEFAULT, None, device, None, None, streams=1,
                            tile=args.tile, plan=plan, packed=packed)
    except Exception as error:       # noqa: BLE001
        failure = error
    all_counts = dist.exchange_counts(counts, shards, failure=failure)
    empty = torch.zeros(0, device=device)
    # small bookkeeping tensors travel on the backend's own device (the GPU
    # for nccl = RCCL, the host for a gloo rehearsal)
    wire = dist.collective_device()

    def step():
        table = dist.exchange_counts(counts, shards)
        if runner is None:
            return dist.exchange_scores(empty, table, shards, flat=True)
        scores = runner.step()
        return dist.exchange_scores(
            scores[runner.columns], table, shards, flat=True)

    def barrier():
        torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(warmup, 1)):
        flat, sizes = step()
    laps = []
    for _ in range(max(1, regions)):
        barrier()
        start = time.perf_counter()
        for _ in range(steps):
            flat, sizes = step()
        barrier()
        slowest = torch.tensor([time.perf_counter() - start],
                               dtype=torch.float64, device=wire)
        torch.distributed.all_reduce(
            slowest, op=torch.distributed.ReduceOp.MAX)
        laps.append(float(slowest.item()))
    # every rank holds every score: the checksums must agree across ranks
    # (summed on the host: one order whatever device the scores came back on)
    checksum = flat.cpu().double().sum().reshape(1).to(wire)
    lowest, highest = checksum.clone(), checksum.clone()
    torch.distributed.all_reduce(lowest, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(highest, op=torch.distributed.ReduceOp.MAX)
    assert float(lowest) == float(highest), 'ranks hold different scores'
    assert int(sizes.sum()) == int(np.asarray(all_counts).sum())
    # the compute alone (no collectives), this rank, same protocol
    compute_ms = None
    kernels, passes = {}, kernel_passes
    if runner is not None:
        torch.cuda.synchronize()
        start = time.perf_counter()
        for _ in range(steps):
            runner.step()
        torch.cuda.synchronize()
        compute_ms = (time.perf_counter() - start) / steps * 1e3
        kernels, passes = runner.kernel_times(kernel_passes)
        if 'conv_spans' in runner.meta:
            kernels['_stack_spans'] = runner.meta['conv_spans'][1] // 8
    compute = torch.tensor([compute_ms or 0.], dtype=torch.float64,
                           device=wire)
    every = torch.zeros(world, dtype=torch.float64, device=wire)
    torch.distributed.all_gather_into_tensor(every, compute)
    line = summary(laps, steps)
    seconds = line['timed_region_s'] / steps
    loads = [int(frames[s].sum()) for s in shards]
    total_frames, total_words = int(frames.sum()), int(sizes.sum())
    line.update(rates(len(frames), total_frames, total_words, seconds))
    line.update({
        'workload': f'{description}, sharded over {world} rank(s) by '
                    'dist.assign (LPT by frames); one step = the whole job: '
                    'exchange_counts + shard forward (hipGraph replay, audio '
                    'resident) + exchange_scores, both collectives timed',
        'steps': steps, 'n_gpus': world, 'scaling': 'strong',
        'utterances': len(frames), 'frames': total_frames,
        'scores': total_words, 'checksum': float(checksum.item()),
        'frames_per_rank': loads,
        'lpt_imbalance': max(loads) / (sum(loads) / len(loads)),
        'compute_only_ms_per_rank': [float(v) for v in every.tolist()],
        'collectives_and_reorder_ms':
            seconds * 1e3 - float(every.max().item()),
        'mfma_frac_per_gpu': line['mfma_frac'] / world,
        'memory_allocated_GB': torch.cuda.max_memory_allocated() / 1e9})
    line['mfma_frac'] = line.pop('mfma_frac_per_gpu')
    line['hbm_frac_compulsory'] /= world
    del runner
    torch.cuda.empty_cache()
    return line, kernels, passes


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_workers(args):
    """`--gpus N` outside a torchrun environment: start the N workers the
    way the driver does, before this process has touched the GPU (a child
    process, never an exec), and leave with their exit code."""
    visible = torch.cuda.device_count()      # (does not initialise the GPU)
    if args.backend == 'nccl' and visible < args.gpus:
        sys.exit(f'bench.py --gpus {args.gpus}: {visible} GPU(s) visible; '
                 "RCCL needs one GPU per rank (use --backend gloo to rehearse "
                 'N ranks on fewer GPUs)')
    command = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
               f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(free_port()), os.path.abspath(__file__)]
    sys.exit(subprocess.run(command + sys.argv[1:], cwd=ROOT).returncode)


###############################################################################
# Main
###############################################################################


LINE_OUT = None


LINE_LIMIT = 10000            # bytes; the driver parsed 12 KB, not 25 KB
LINE_KEYS = (
    'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
    'ms_per_step_min', 'ms_per_step_max', 'timed_region_s', 'regions',
    'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
    'frames_per_s_per_gpu', 'checksum')
ROOFLINE_KEYS = (
    'bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic',
    'avg_launch_us', 'avg_launch_us_source', 'avg_launch_us_events',
    'launches_per_step', 'share_of_step', 'executed_mfma_flops_per_launch',
    'algorithmic_flops_per_launch', 'useful_tflops', 'frac_algorithmic',
    'frac_pmc_pipe_busy', 'rocprof_avg_launch_us', 'rocprof_file',
    'frac_at_rocprof_avg', 'traffic_source')
JOB_KEYS = ('utterances', 'frames', 'scores', 'checksum', 'frames_per_rank',
            'lpt_imbalance', 'frames_per_s', 'compute_only_ms_per_rank',
            'collectives_and_reorder_ms', 'memory_allocated_GB')
CPU_KEYS = ('value', 'unit', 'cores', 'kind', 'sample', 'cpu',
            'cgroup_cpu_quota', 'physical_cores')


def dig(record, *path):
    """`record[path[0]][path[1]]...` or None (side measurements may have
    failed: `guarded` leaves {'error': ...} in their place)."""
    for key in path:
        if not isinstance(record, dict) or key not in record:
            return None
        record = record[key]
    return record


def rounded(value, digits=6):
    """Significant digits enough for a report; the side file keeps them all."""
    if isinstance(value, float):
        return float(f'{value:.{digits}g}')
    if isinstance(value, dict):
        return {key: rounded(item, digits) for key, item in value.items()}
    if isinstance(value, (list, tuple)):
        return [rounded(item, digits) for item in value]
    return value


def compact_line(result):
    """The contract's ONE line from the full record: the headline, `roofline`
    and `cpu_baseline` as objects, every side measurement as a scalar."""
    line = {key: result[key] for key in LINE_KEYS if key in result}
    if 'roofline' in result:
        line['roofline'] = rounded({
            key: result['roofline'][key] for key in ROOFLINE_KEYS
            if result['roofline'].get(key) is not None or key == 'traffic'})
    if 'cpu_baseline' in result:
        host = result['cpu_baseline']
        line['cpu_baseline'] = rounded({
            key: host[key] for key in CPU_KEYS if key in host}) \
            if 'error' not in host else host
    if 'job' in result:         # (--workload corpus | longform)
        line['job'] = {key: result['job'][key] for key in JOB_KEYS
                       if key in result['job']}
    scalars = {
        # opt-in precision on the headline workload (configs[1])
        'bf16x3_ms_per_step': ('configs_1_conv_bf16x3', 'ms_per_step'),
        'bf16x3_utterances_per_s':
            ('configs_1_conv_bf16x3', 'utterances_per_s'),
        'bf16x3_max_abs_dscore':
            ('configs_1_conv_bf16x3', 'max_abs_dscore_vs_f32'),
        # BASELINE configs[2]
        'configs_2_f32_ms_per_step': ('configs_2_transformer', 'ms_per_step'),
        'configs_2_f32_roofline_frac':
            ('configs_2_transformer', 'roofline', 'frac'),
        'configs_2_bf16x3_ms_per_step':
            ('configs_2_transformer_bf16x3', 'ms_per_step'),
        'configs_2_bf16x3_max_abs_dscore':
            ('configs_2_transformer_bf16x3', 'max_abs_dscore_vs_f32'),
        'configs_2_bf16x3_useful_tflops':
            ('configs_2_transformer_bf16x3', 'roofline', 'useful_tflops'),
        'configs_2_bf16x3_fast_ms_per_step':
            ('configs_2_transformer_bf16x3_fast', 'ms_per_step'),
        'configs_2_bf16x3_fast_max_abs_dscore':
            ('configs_2_transformer_bf16x3_fast', 'max_abs_dscore_vs_f32'),
        'configs_2_bf16x6_ms_per_step':
            ('configs_2_transformer_bf16x6', 'ms_per_step'),
        'configs_2_bf16x6_max_abs_dscore':
            ('configs_2_transformer_bf16x6', 'max_abs_dscore_vs_f32'),
        # BASELINE configs[3] / [4] on one GPU
        'configs_3_shard_utterances_per_s':
            ('configs_3_corpus', 'rank0_of_8_device_only', 'utterances_per_s'),
        'configs_3_shard_frames_per_s':
            ('configs_3_corpus', 'rank0_of_8_device_only', 'frames_per_s'),
        'configs_3_whole_corpus_utterances_per_s':
            ('configs_3_corpus', 'whole_corpus_device_only',
             'utterances_per_s'),
        'configs_3_shard_bf16x3_frames_per_s':
            ('configs_3_corpus', 'rank0_of_8_device_only_bf16x3',
             'frames_per_s'),
        'configs_3_shard_bf16x3_max_abs_dscore':
            ('configs_3_corpus', 'rank0_of_8_device_only_bf16x3',
             'max_abs_dscore_vs_f32'),
        'configs_4_frames_per_s':
            ('configs_4_longform', 'device_only', 'frames_per_s'),
        'configs_4_bf16x3_frames_per_s':
            ('configs_4_longform', 'device_only_bf16x3', 'frames_per_s'),
        'configs_4_bf16x3_max_abs_dscore':
            ('configs_4_longform', 'device_only_bf16x3',
             'max_abs_dscore_vs_f32'),
        'configs_4_api_pcm16_frames_per_s':
            ('configs_4_longform', 'api_pcm16', 'frames_per_s'),
        # ... strong-scaled over the ranks (N > 1)
        'configs_3_sharded_utterances_per_s':
            ('configs_3_corpus_sharded', 'utterances_per_s'),
        'configs_3_sharded_ms_per_job':
            ('configs_3_corpus_sharded', 'ms_per_step'),
        'configs_4_sharded_utterances_per_s':
            ('configs_4_longform_sharded', 'utterances_per_s'),
        'configs_4_sharded_ms_per_job':
            ('configs_4_longform_sharded', 'ms_per_step'),
        # the public API, PCIe included (never `value`)
        'api_float32_utterances_per_s':
            ('end_to_end_api', 'float32', 'utterances_per_s_pipelined'),
        'api_pcm16_utterances_per_s':
            ('end_to_end_api', 'pcm16', 'utterances_per_s_pipelined'),
        'files_api_files_per_s': ('files_api', 'files_per_s'),
        # BASELINE configs[0]'s counterpart: ONE from_alignment_and_audio call
        'single_utterance_ms_p50':
            ('single_utterance_api', 'default', 'ms_p50'),
        'single_utterance_ms_p99':
            ('single_utterance_api', 'default', 'ms_p99'),
        'single_utterance_tile_auto_ms_p50':
            ('single_utterance_api', 'conv_tile_auto', 'ms_p50'),
        'single_utterance_oracle_1_core_ms':
            ('single_utterance_api', 'oracle_1_core_ms'),
        'end_to_end_mfma_frac': ('end_to_end', 'mfma_frac'),
        'end_to_end_hbm_frac_compulsory':
            ('end_to_end', 'hbm_frac_compulsory')}
    side = {}
    for name, path in scalars.items():
        value = dig(result, *path)
        if value is not None:
            side[name] = value
    failed = sorted(key for key, value in result.items()
                    if isinstance(value, dict) and 'error' in value)
    if failed:
        side['failed'] = failed
    if side:
        line['side'] = rounded(side)
    return line


def side_path(args):
    if args.side_records:
        return args.side_records
    parts = [part for part, default in (
        (args.workload, 'batch'), (args.config, 'conv'),
        (args.precision, 'f32')) if part != default]
    if args.gpus > 1:
        parts.append(f'{args.gpus}gpus')
    name = '_'.join(['bench_side'] + parts) + '.json'
    return os.path.join(ROOT, name)


def emit(result, args):
    """The full record to the side-records file; the run's ONE line - under
    `LINE_LIMIT` bytes whatever the side measurements held - on the process's
    original stdout."""
    import hashlib
    result = dict(result, notes={'roofline': ROOFLINE_NOTES,
                                 'line': 'stdout carries LINE_KEYS, roofline, '
                                         'cpu_baseline and one scalar per '
                                         'side measurement of this record'})
    line = compact_line(result)
    path = side_path(args)
    text = json.dumps(result, indent=1, default=repr)
    try:
        with open(path, 'w') as file:
            file.write(text + '\n')
        line['side_records'] = {
            'file': os.path.relpath(path, ROOT), 'bytes': len(text) + 1,
            'sha256': hashlib.sha256((text + '\n').encode()).hexdigest()}
    except OSError as error:
        line['side_records'] = {'error': repr(error)}
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:       # (cannot happen with the keys above)
        line.pop('side', None)
        text = json.dumps(line)
    assert len(text) < LINE_LIMIT, len(text)
    out = LINE_OUT or sys.stdout
    out.write(text + '\n')
    out.flush()


def main():
    args = parse_args()
    if args.cpu_worker is not None:
        cpu_worker(args.cpu_worker)
        return
    if 'RANK' not in os.environ and args.gpus > 1:
        spawn_workers(args)             # (does not return)
    # The contract is ONE JSON line on stdout.  Libraries write there too (gloo
    # announces its peers on stdout, RCCL can be told to log): keep the real
    # stdout for the line and send everything else, from any layer, to stderr.
    global LINE_OUT
    LINE_OUT = os.fdopen(os.dup(1), 'w')
    sys.stdout.flush()
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        sys.exit(f'bench.py --gpus {args.gpus} inside a process group of '
                 f'{world} rank(s): launch one process per GPU '
                 '(python -m torch.distributed.run --nproc-per-node N ... '
                 'bench.py --gpus N), or leave the torchrun environment out '
                 'and bench.py starts the workers itself')
    sharded = args.workload != 'batch'

    # The host baseline first: its child processes are started before this
    # process has touched the GPU, and nothing else competes for the cores.
    # Rank 0 runs it at EVERY world size and workload (an N > 1 or
    # strong-scaling line without it reads as unmeasured); the other ranks wait
    # for rank 0 inside init_process_group, idle.
    host = None
    if rank == 0 and not args.no_cpu_baseline:
        host = guarded(cpu_baseline)

    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1 or sharded:
        import datetime
        if 'MASTER_PORT' not in os.environ:
            os.environ['MASTER_PORT'] = str(free_port())
        torch.distributed.init_process_group(
            args.backend, rank=rank, world_size=world,
            timeout=datetime.timedelta(minutes=10),
            device_id=device if args.backend == 'nccl' else None)
    try:
        if sharded:
            run_sharded(args, rank, world, device, host)
        else:
            run_batch(args, rank, world, device, host)
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


def run_sharded(args, rank, world, device, host):
    """`--workload corpus | longform`: one line for the whole strong-scaled
    job."""
    line, kernels, passes = sharded_job(
        args.workload, args, rank, world, device, args.steps, args.warmup,
        args.regions)
    if rank != 0:
        return
    roof, committed = roofline(
        kernels, passes, None, 'conv', spans=kernels.pop('_stack_spans', None),
        layers=1 + cfg.DEFAULT.layers)
    unit = 'utterances/s'
    result = {
        'metric': 'utterances/s (mixed 2-30 s @16 kHz) whole-node'
        if args.workload == 'corpus' else
        'utterances/s (5 min @16 kHz, batch_size 3000) whole-node',
        'value': line['utterances_per_s'], 'unit': unit,
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': line['ms_per_step'],
        'ms_per_step_min': line['ms_per_step_min'],
        'ms_per_step_max': line['ms_per_step_max'],
        'timed_region_s': line['timed_region_s'], 'regions': line['regions'],
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': line['workload'],
                   'parallelism': f'utterance-sharded x{world} (LPT)',
                   'exchange': f'two {args.backend} all_gathers per job '
                               '(score counts, padded scores), both inside '
                               'the timed region'},
        'frames_per_s_per_gpu': line['frames_per_s'] / world,
        'roofline': roof, 'from_profiles': committed,
        'job': line}
    if host is not None:
        result['cpu_baseline'] = host
    emit(result, args)


def run_batch(args, rank, world, device, host):
    """The headline: BASELINE configs[1] per GPU (weak scaling)."""
    config = cfg.DEFAULT if args.config == 'conv' else \
        cfg.Config(architecture='transformer')
    state = None if args.config == 'conv' else \
        emphases_amd.weights.random_state(config, seed=0)
    audios, alignments, bounds = workload(rank)
    runner = Runner(config, state, device, audios, alignments,
                    streams=args.streams, tile=args.tile,
                    winograd=not args.no_winograd, graph=not args.no_graph,
                    precision=args.precision)
    plan, columns = runner.plan, runner.columns

    # The one exchange of the path (north_star: "RCCL gather of per-word scores
    # only at the end"): every step leaves its dense per-word
    # scores in a row of `send_all`, and ONE all_gather of all rows closes the
    # timed region.  No collective sits between the steps.
    exchange = None
    if world > 1:
        # ranks hold different numbers of words: pad to the largest shard
        most = torch.tensor([plan.total_words], device=device)
        torch.distributed.all_reduce(most, op=torch.distributed.ReduceOp.MAX)
        most_words = int(most.item())
        rows = max(args.steps, 1)
        send_all = torch.zeros(
            rows, most_words, dtype=torch.float32, device=device)
        gathered_all = torch.empty(
            world * rows * most_words, dtype=torch.float32, device=device)

        def keep(scores, index):
            send_all[index % rows, :plan.total_words] = scores[columns]
        runner.after = keep

        def exchange():
            """All ranks' scores of all steps on every rank: one all_gather."""
            for stream, *_ in runner.lanes:
                torch.cuda.current_stream().wait_stream(stream)
            torch.distributed.all_gather_into_tensor(
                gathered_all, send_all.view(-1))

    laps, ramp, scores = timed_regions(
        runner, args.steps, args.warmup, args.regions, world, exchange,
        skip_preroll=args.no_preroll)
    if world > 1:
        # every rank now holds every rank's scores: its own rows came back intact
        mine = gathered_all.view(world, rows, most_words)[rank]
        assert torch.equal(mine, send_all), 'all_gather returned other scores'
    line = summary(laps, args.steps)
    elapsed = line['timed_region_s']

    # Dominant kernel, timed live with HIP events on the launch stream
    runner.after = None
    kernels, passes = runner.kernel_times(20)
    checksum = float(scores[columns].sum().item())
    headline_scores = scores[columns].clone()
    meta_tile, lanes = runner.meta['tile'], len(runner.lanes)
    stack_spans = runner.meta['conv_spans'][1] // 8 \
        if 'conv_spans' in runner.meta else None
    frame_layers = 1 + config.layers        # input layer + encoder layers
    del runner
    torch.cuda.empty_cache()

    # With N > 1 ranks the same line carries BASELINE configs[3] / configs[4]
    # strong-scaled through dist.assign + the two collectives (every rank
    # takes part; a failure on one rank raises on all of them, see dist.py).
    sharded = {}
    if world > 1 and args.config == 'conv' and not args.no_side:
        for name, workload_name in (('configs_3_corpus_sharded', 'corpus'),
                                    ('configs_4_longform_sharded', 'longform')):
            try:
                start = time.perf_counter()
                sharded[name] = sharded_job(
                    workload_name, args, rank, world, device, 5, 2, 3, 1)[0]
                sharded[name]['seconds_spent'] = time.perf_counter() - start
            except Exception as error:       # noqa: BLE001
                sharded[name] = {'error': repr(error)}

    if rank == 0:
        roof, committed = roofline(
            kernels, passes, line['ms_per_step'], args.config,
            spans=stack_spans, layers=frame_layers, precision=args.precision)
        total_utterances = UTTERANCES * world
        result = {
            'metric': 'utterances/s (10 s @16 kHz) whole-node',
            'value': total_utterances * args.steps / elapsed,
            'unit': 'utterances/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': line['ms_per_step'],
            'ms_per_step_min': line['ms_per_step_min'],
            'ms_per_step_max': line['ms_per_step_max'],
            'timed_region_s': elapsed, 'regions': line['regions'],
            'timing': (
                f'{args.steps} steps timed {line["regions"]} times after a '
                'steady-state pre-roll and the warm-up; the median region is '
                'reported (barrier + synchronize on both sides, max over '
                'ranks)'),
            'preroll': ramp,
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'f32' else
            f'f32 operands, matrix products as {args.precision} '
            '(bf16 pieces, fp32 accumulation; opt-in, not the headline)',
            'data': 'synthetic',
            'config': {
                'workload': (
                    f'{UTTERANCES} synthetic 10 s 16 kHz utterances per GPU, '
                    f'{args.config} config, random word alignments, ragged '
                    'batch with per-utterance (B=1) semantics '
                    '(BASELINE.json configs[1])'),
                'utterances_per_gpu': UTTERANCES, 'frames_per_gpu':
                    plan.total_frames, 'words_per_gpu': plan.total_words,
                'conv_tile': meta_tile,
                'launch': 'eager' if args.no_graph else 'hipGraph replay',
                'batches_in_flight': lanes,
                'parallelism': f'utterance-sharded x{world}',
                'exchange': 'none (one rank)' if world == 1 else
                f'one {args.backend} all_gather of the {args.steps} steps\' '
                'per-word scores at the end of each timed region'},
            'frames_per_s_per_gpu': plan.total_frames * args.steps / elapsed,
            'roofline': roof,
            'from_profiles': committed,
            # the whole path against both ceilings (per GPU).
            # The conv path is MFMA-bound: its compulsory HBM traffic is only
            # 641 B per frame.
            'end_to_end': {
                'mfma_frac': (plan.total_frames * FLOPS_PER_FRAME +
                              plan.total_words * FLOPS_PER_WORD) * args.steps /
                             elapsed / (PEAK_FP32_MFMA * 1e12),
                'hbm_frac_compulsory': plan.total_frames * BYTES_PER_FRAME *
                                       args.steps / elapsed / PEAK_HBM,
                'binds': 'mfma'} if args.config == 'conv' else None,
            # NOT a decomposition of ms_per_step: these are HIP-event times of
            # eager launches on ONE stream (each pair brackets its dispatch
            # gap), while the step replays a graph with two batches in
            # flight, whose kernels overlap - the sum is larger than the step.
            'kernels_us_per_step': {
                name: value[3] / passes * 1e6
                for name, value in kernels.items()},
            'kernels_us_per_step_events': {
                name: value[1] / passes * 1e6
                for name, value in kernels.items()},
            'kernels_us_per_step_is': (
                'eager launches, one stream: kernel-exact (emph_launch_timer) '
                'and between recorded HIP events (incl. dispatch gap); the '
                'step itself is a two-lane graph replay with overlapping '
                'kernels'),
            'kernels_us_rocprof': rocprof_kernels(args.config),
        }
        result['checksum'] = checksum
        result.update(sharded)
        if host is not None:
            result['cpu_baseline'] = host
        if world == 1 and args.config == 'conv':
            if not args.no_api:
                result['end_to_end_api'] = guarded(
                    end_to_end_api, audios, alignments)
                result['single_utterance_api'] = guarded(
                    single_utterance_api, audios, alignments, host)
            if not args.no_side:
                result['configs_1_conv_bf16x3'] = guarded(
                    side_conv_split, device, audios, alignments, args,
                    headline_scores)
                result['files_api'] = guarded(side_files_api, device)
                plain = guarded(
                    side_transformer, device, audios, alignments, args)
                reference_scores = plain.pop('_scores', None)
                result['configs_2_transformer'] = plain
                for precision in ('bf16x3', 'bf16x3_fast', 'bf16x6'):
                    entry = guarded(
                        side_transformer, device, audios, alignments, args,
                        precision, reference_scores)
                    entry.pop('_scores', None)
                    result[f'configs_2_transformer_{precision}'] = entry
                result['configs_4_longform'] = guarded(side_longform, device)
                result['configs_3_corpus'] = guarded(side_corpus, device)
        emit(result, args)


if __name__ == '__main__':
    main()
