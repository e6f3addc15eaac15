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
end of the timed region).

Prints ONE JSON line on rank 0 with BASELINE.json's metric (utterances/s,
whole job), the roofline of the dominant kernel (fp32-MFMA conv1d, measured
live with HIP events on the launch stream) and the CPU oracle timed on the
host cores beside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import emphases_amd  # noqa: E402
from emphases_amd import batch, config as cfg, runtime, synth  # noqa: E402

UTTERANCES = 64
FRAMES = 1000                 # 10 s at 100 frames/s
PEAK_FP32_MFMA = 157.3        # TFLOP/s, MI355X_MICROARCH.md (dense, f32 in)
PEAK_HBM = 8.0e12             # B/s, same guide
# compulsory traffic and algorithmic flops of the conv path
BYTES_PER_FRAME = 641.
FLOPS_PER_FRAME, FLOPS_PER_WORD = 0.2968e6, 0.2309e6


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=200)
    parser.add_argument('--warmup', type=int, default=20)
    parser.add_argument('--config', default='conv',
                        choices=['conv', 'transformer'])
    parser.add_argument('--tile', type=int, default=None)
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-api', action='store_true',
                        help='skip the end-to-end public-API measurement')
    parser.add_argument('--streams', type=int, default=2,
                        help='batches in flight: consecutive steps alternate '
                             'between this many HIP streams (each with its '
                             'own workspace), so that the VALU-bound front-end '
                             'of one batch overlaps the MFMA-bound encoder of '
                             'the previous one')
    parser.add_argument('--no-winograd', action='store_true',
                        help='direct (3-tap) form of the frame-rate convs '
                             'instead of Winograd F(2,3)')
    parser.add_argument('--backend', default='nccl',
                        help="torch.distributed backend ('nccl' = RCCL; "
                             "'gloo' only to rehearse the N > 1 path on a box "
                             'with fewer GPUs than ranks)')
    parser.add_argument('--no-graph', action='store_true',
                        help='launch kernel by kernel instead of replaying '
                             'the captured HIP graph')
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
    """Physical cores of this box (/proc/cpuinfo: distinct (physical id,
    core id) pairs), falling back to the logical count."""
    cores = set()
    try:
        physical = core = None
        with open('/proc/cpuinfo') as file:
            for line in file:
                if line.startswith('physical id'):
                    physical = line.split(':')[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':')[1].strip()
                elif not line.strip():
                    if physical is not None and core is not None:
                        cores.add((physical, core))
                    physical = core = None
    except OSError:
        pass
    return len(cores) or (os.cpu_count() or 1)


def cpu_baseline(audios, bounds, seconds=5.0):
    """The CPU oracle (port of the reference's op sequence, B=1 loop exactly
    like `emphases/core.py:169-179`) on this box's host cores, at 1 thread, at
    8 threads (what the survey measured the reference itself with) and at all
    physical cores."""
    from oracle import prominence as oracle
    from emphases_amd import weights
    state = {k: torch.from_numpy(v) for k, v in weights.load().items()}
    words = [[(int(s) / 100., int(e) / 100.) for s, e in b.T] for b in bounds]
    tensors = [torch.from_numpy(a) for a in audios]
    physical = physical_cores()
    runs = []
    for threads in sorted({1, min(8, physical), physical}):
        torch.set_num_threads(threads)
        oracle.from_alignment_and_audio(words[0], tensors[0], state)  # warm up
        done = 0
        start = time.perf_counter()
        while time.perf_counter() - start < seconds:
            index = done % len(tensors)
            oracle.from_alignment_and_audio(
                words[index], tensors[index], state)
            done += 1
        elapsed = time.perf_counter() - start
        runs.append({
            'threads': threads, 'value': done / elapsed,
            'utterances': done, 'seconds': elapsed})
    best = max(runs, key=lambda run: run['value'])
    return {
        'value': best['value'], 'unit': 'utterances/s',
        'cores': best['threads'], 'kind': 'port',
        'threads': {str(run['threads']): run['value'] for run in runs},
        'physical_cores': physical, 'logical_cores': os.cpu_count(),
        'cpu': cpu_model(),
        'sample': (
            f"{best['utterances']} x 10 s utterances in "
            f"{best['seconds']:.1f} s at {best['threads']} torch threads "
            f"(the fastest of {[run['threads'] for run in runs]} threads, "
            f"{seconds:.0f} s each), one at a time (B=1) through "
            'oracle/prominence.py, torch CPU fp32'),
        # BASELINE.md / SURVEY.md §6: the reference itself (bf16 autocast as
        # shipped), measured in the survey container (8 vCPU Xeon @2.1 GHz)
        'reference_survey': {
            'ms_per_utterance_1_thread': 18.8,
            'ms_per_utterance_8_threads': 10.4,
            'utterances_per_s_1_thread': 1000. / 18.8,
            'utterances_per_s_8_threads': 1000. / 10.4,
            'where': 'survey container, 8 vCPU Xeon 2.1 GHz, reference '
                     'from_alignment_and_audio as shipped'}}


def end_to_end_api(audios, alignments, rounds=40):
    """SURVEY.md §8(d) protocol (ii): the public batch API on the same 64
    utterances, host tensors in, scores out - chunk planning, staging, H2D,
    kernels, D2H and the split into per-utterance tensors all inside the
    clock.  `call` = one synchronous `from_alignments_and_audios` after the
    other; `pipelined` = `Session.submit` with two batches in flight (what
    `from_files_to_files` does).  float32 = the reference's input type
    (pageable CPU tensors); pcm16 = the same audio as 16-bit PCM tensors."""
    floats = [torch.from_numpy(a) for a in audios]
    pcm = [torch.from_numpy(np.rint(a * 32768.).astype(np.int16))
           for a in audios]
    session = emphases_amd.get_session(None, 0)
    result = {}
    reference = None
    for name, tensors in (('float32', floats), ('pcm16', pcm)):
        for _ in range(8):          # both lanes: buffers, layout cache, graph
            scores = emphases_amd.from_alignments_and_audios(
                alignments, tensors, 16000)
        torch.cuda.synchronize()
        laps = []
        for _ in range(rounds):
            start = time.perf_counter()
            scores = emphases_amd.from_alignments_and_audios(
                alignments, tensors, 16000)
            laps.append(time.perf_counter() - start)
        call = float(np.median(laps))
        # pipelined: three runs of `rounds` submissions, the median run counts
        runs = []
        for _ in range(3):
            start = time.perf_counter()
            previous = None
            for _ in range(rounds):
                pending = session.submit(alignments, tensors, 16000)
                if previous is not None:
                    previous.result()
                previous = pending
            scores = previous.result()
            runs.append((time.perf_counter() - start) / rounds)
        piped = float(np.median(runs))
        flat = torch.cat([s.reshape(-1) for s in scores])
        if reference is None:
            reference = flat
        result[name] = {
            'ms_per_call': call * 1e3,
            'ms_per_call_mean': float(np.mean(laps)) * 1e3,
            'ms_per_call_worst': float(np.max(laps)) * 1e3,
            'utterances_per_s': len(audios) / call,
            'ms_per_call_pipelined': piped * 1e3,
            'utterances_per_s_pipelined': len(audios) / piped,
            'bit_identical_to_float32': bool(torch.equal(flat, reference))}
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
    result['float32_new_layout_every_call'] = {
        'ms_per_call': float(np.median(laps)) * 1e3,
        'ms_per_call_pipelined': piped * 1e3,
        'utterances_per_s_pipelined': len(audios) / piped}
    result['what'] = (
        'emphases_amd.from_alignments_and_audios on 64 x 10 s host tensors: '
        'planning + staging + H2D + kernels + D2H; pipelined = 2 batches in '
        'flight (session.Session); ms_per_call = median of the laps')
    return result


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        torch.distributed.init_process_group(
            args.backend, rank=rank, world_size=world,
            device_id=device if args.backend == 'nccl' else None)

    config = cfg.DEFAULT if args.config == 'conv' else \
        cfg.Config(architecture='transformer')
    state = None if args.config == 'conv' else \
        emphases_amd.weights.random_state(config, seed=0)
    engine = emphases_amd.engine.Engine(
        config, state, device, conv_tile=args.tile,
        winograd=not args.no_winograd)

    audios, alignments, bounds = workload(rank)
    plan = build_plan(audios, alignments)
    packed = torch.cat(
        [torch.from_numpy(a).reshape(-1) for a in audios]).to(device)
    meta = engine.upload(plan)
    columns = torch.from_numpy(plan.word_columns()).to(device)
    most_words = plan.total_words
    if world > 1:
        # ranks hold different numbers of words: pad to the largest shard
        most = torch.tensor([plan.total_words], device=device)
        torch.distributed.all_reduce(most, op=torch.distributed.ReduceOp.MAX)
        most_words = int(most.item())

    # The one exchange of the path (north_star: "RCCL gather of per-word scores
    # only at the end"): every step leaves its dense per-word
    # scores in a row of `send_all`, and ONE all_gather of all rows closes the
    # timed region.  No collective sits between the steps.
    rows = max(args.steps, 1)
    if world > 1:
        send_all = torch.zeros(rows, most_words, dtype=torch.float32, device=device)
        gathered_all = torch.empty(
            world * rows * most_words, dtype=torch.float32, device=device)

    # One lane per stream: its own engine workspace (weights are shared
    # read-only through the same state), its own captured graph.
    lanes = []
    for index in range(max(1, args.streams)):
        lane_engine = engine if index == 0 else emphases_amd.engine.Engine(
            config, state, device, conv_tile=args.tile,
            winograd=not args.no_winograd)
        stream = torch.cuda.Stream(device=device) if args.streams > 1 \
            else torch.cuda.current_stream()
        with torch.cuda.stream(stream):
            if args.no_graph:
                lanes.append((stream, lane_engine, None, None))
            else:
                replay, buffer, _ = lane_engine.capture(packed, plan, meta)
                lanes.append((stream, lane_engine, replay, buffer))
    torch.cuda.synchronize()
    counter = [0]

    def step():
        stream, lane_engine, replay, buffer = lanes[counter[0] % len(lanes)]
        row = counter[0] % rows
        counter[0] += 1
        with torch.cuda.stream(stream):
            if replay is None:
                scores = lane_engine.forward(packed, plan, meta)[0]
            else:
                replay()
                scores = buffer
            if world > 1:
                send_all[row, :plan.total_words] = scores[columns]
        return scores

    def exchange():
        """All ranks' scores of all steps on every rank: one RCCL all_gather."""
        if world == 1:
            return
        for stream, *_ in lanes:
            torch.cuda.current_stream().wait_stream(stream)
        torch.distributed.all_gather_into_tensor(
            gathered_all, send_all.view(-1))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    exchange()
    counter[0] = 0
    barrier()
    start = time.perf_counter()
    for _ in range(args.steps):
        scores = step()
    exchange()
    barrier()
    elapsed = time.perf_counter() - start
    if world > 1:
        # every rank now holds every rank's scores: its own rows came back intact
        mine = gathered_all.view(world, rows, most_words)[rank]
        assert torch.equal(mine, send_all), 'all_gather returned other scores'
        slowest = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(
            slowest, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(slowest.item())

    # Dominant kernel, timed live with HIP events on the launch stream
    engine.timers = []
    for _ in range(min(args.steps, 10)):
        engine.forward(packed, plan, meta)
    torch.cuda.synchronize()
    kernels = {}
    for name, flops, begin, end in engine.timers:
        entry = kernels.setdefault(name, [0, 0., 0.])
        entry[0] += 1
        entry[1] += begin.elapsed_time(end) * 1e-3
        entry[2] += flops
    engine.timers = None
    dominant = max(kernels, key=lambda name: kernels[name][1])
    launches, seconds, flops = kernels[dominant]
    achieved = flops / seconds / 1e12
    # HBM bytes per launch of that kernel from the committed PMC passes
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs)
    traffic = None
    for tag in ('r2', 'r1'):
        name = f'{tag}_pmc_summary.json' if args.config == 'conv' else \
            f'{tag}_transformer_pmc_summary.json'
        summary = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(summary):
            with open(summary) as file:
                traffic = json.load(file).get(dominant, {}).get(
                    'traffic_bytes')
            break

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_utterances = UTTERANCES * world
        result = {
            'metric': 'utterances/s (10 s @16 kHz) whole-node',
            'value': total_utterances * args.steps / elapsed,
            'unit': 'utterances/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': (
                    f'{UTTERANCES} synthetic 10 s 16 kHz utterances per GPU, '
                    f'{args.config} config, random word alignments, ragged '
                    'batch with per-utterance (B=1) semantics '
                    '(BASELINE.json configs[1])'),
                'utterances_per_gpu': UTTERANCES, 'frames_per_gpu':
                    plan.total_frames, 'words_per_gpu': plan.total_words,
                'conv_tile': meta['tile'],
                'launch': 'eager' if args.no_graph else 'hipGraph replay',
                'batches_in_flight': len(lanes),
                'parallelism': f'utterance-sharded x{world}',
                'exchange': 'none (one rank)' if world == 1 else
                f'one {args.backend} all_gather of the {args.steps} steps\' '
                'per-word scores at the end of the timed region'},
            'frames_per_s_per_gpu': plan.total_frames * args.steps / elapsed,
            'roofline': {
                'bound': 'mfma', 'kernel': dominant,
                'achieved': achieved, 'peak': PEAK_FP32_MFMA,
                'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_MFMA,
                'avg_launch_us': seconds / launches * 1e6,
                'share_of_step': seconds / min(args.steps, 10) /
                    (elapsed / args.steps),
                'traffic': traffic,
                'algorithmic_flops_per_launch': flops / launches,
                # Winograd F(2,3) executes 2/3, F(4,3) 1/2 of the direct form's
                # MFMA work
                # SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) of this
                # kernel, profiles/r2_pmc_utilisation.md
                'mfma_pipe_busy_pmc': 0.49 if 'winograd4' in dominant else (
                    0.74 if dominant == 'attention_frames' else None),
                'executed_mfma_flops_per_launch': flops / launches * (
                    .5 if 'winograd4' in dominant else
                    2. / 3. if 'winograd' in dominant else 1.)},
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
            'kernels_us_per_step': {
                name: value[1] / min(args.steps, 10) * 1e6
                for name, value in kernels.items()},
        }
        check = float(scores[columns].sum().item())
        result['checksum'] = check
        # (the side measurements must never cost the line its headline)
        if world == 1 and args.config == 'conv' and not args.no_api:
            try:
                result['end_to_end_api'] = end_to_end_api(audios, alignments)
            except Exception as error:       # noqa: BLE001
                result['end_to_end_api'] = {'error': repr(error)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                result['cpu_baseline'] = cpu_baseline(audios, bounds)
            except Exception as error:       # noqa: BLE001
                result['cpu_baseline'] = {'error': repr(error)}
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
