"""The HIP engine under N > 1 ranks (BASELINE configs[3]): fresh
child processes, one per rank, running `dist.from_alignments_and_audios` with
its DEFAULT compute.  The GPU box has one MI355X, so the two gloo ranks share
`cuda:0` (LOCAL_RANK modulo the device count); a world_size-1 `nccl` run takes
RCCL initialisation and the device-tensor all_gather through once.  The
scores must be bitwise those of a single-process run: no arithmetic crosses a
rank boundary."""
import os
import socket
import subprocess
import sys

import pytest
import torch

import emphases_amd
from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = os.path.join(ROOT, 'tests', 'dist_worker.py')
COUNT, LOW, HIGH = 200, 200, 3000        # 2-30 s utterances


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _launch(backend, world, tmp_path, count=COUNT, files=None,
            precision='f32'):
    port = _free_port()
    children = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0',
                   EMPHASES_TEST_PRECISION=precision)
        out = tmp_path / f'{backend}_{world}_{rank}_{precision}.pt'
        arguments = [backend, str(count), str(LOW), str(HIGH), str(out)] \
            if files is None else \
            [backend, 'files', str(files), str(count), str(out)]
        children.append((out, subprocess.Popen(
            [sys.executable, WORKER] + arguments, env=env, cwd=ROOT)))
    results = []
    for out, child in children:
        assert child.wait(timeout=600) == 0
        results.append(torch.load(out))
    return results


@pytest.fixture(scope='module')
def single_process():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dist_worker
    aligns, audios = dist_worker.corpus(COUNT, LOW, HIGH)
    scores = emphases_amd.from_alignments_and_audios(aligns, audios, gpu=0)
    return [s.cpu() for s in scores], aligns


@pytest.mark.timeout(900)
def test_two_ranks_share_one_gpu(tmp_path, single_process):
    want, aligns = single_process
    results = _launch('gloo', 2, tmp_path)
    assert sum(r['shard'] for r in results) == COUNT
    assert min(r['shard'] for r in results) >= COUNT // 2 - 20
    for result in results:
        assert result['device'] == 'cpu'          # gloo gathers on the host
        assert len(result['scores']) == COUNT
        for got, expect, words in zip(result['scores'], want, aligns):
            assert got.shape == (1, len(words))
            assert torch.equal(got, expect)       # bitwise


@pytest.mark.timeout(900)
def test_one_rank_rccl(tmp_path, single_process):
    want, _ = single_process
    (result,) = _launch('nccl', 1, tmp_path)
    assert result['device'].startswith('cuda')    # RCCL gathers device tensors
    for got, expect in zip(result['scores'], want):
        assert torch.equal(got, expect)


@pytest.mark.timeout(900)
def test_more_ranks_than_utterances(tmp_path):
    """Three ranks, two utterances: a rank with an empty shard still joins the
    collectives on the right device."""
    results = _launch('gloo', 3, tmp_path, count=2)
    assert sorted(r['shard'] for r in results) == [0, 1, 1]
    for result in results[1:]:
        for a, b in zip(result['scores'], results[0]['scores']):
            assert torch.equal(a, b)


@pytest.mark.timeout(900)
def test_bench_with_two_ranks(tmp_path):
    """`bench.py --gpus 2` the way the driver launches it (one process per
    rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) with
    the gloo backend, both ranks on the one GPU of the box: rank 0 prints ONE
    JSON line whose value is the whole job's rate (128 utterances per step),
    the closing all_gather returns every rank's own rows intact (asserted
    inside bench.py), and the other rank prints nothing."""
    import json
    port = _free_port()
    children = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        children.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
             '--steps', '6', '--warmup', '2', '--backend', 'gloo',
             '--no-cpu-baseline', '--no-api'],
            env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True))
    outputs = [child.communicate(timeout=600)[0] for child in children]
    assert all(child.returncode == 0 for child in children)
    lines = [l for l in outputs[0].splitlines() if l.startswith('{')]
    assert len(lines) == 1
    assert not [l for l in outputs[1].splitlines() if l.startswith('{')]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 6
    assert line['scaling'] == 'weak' and line['unit'] == 'utterances/s'
    assert line['config']['utterances_per_gpu'] == 64
    assert 'all_gather' in line['config']['exchange']
    assert abs(line['value'] - 128 / (line['ms_per_step'] * 1e-3)) \
        < 1e-6 * line['value']
    assert line['roofline']['frac'] > 0 and 'cpu_baseline' not in line


def _bench(arguments, world, backend=None, timeout=900):
    """bench.py the way the driver launches it: one process per rank."""
    port = _free_port()
    children = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        children.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus',
             str(world)] + arguments +
            (['--backend', backend] if backend else []),
            env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True))
    outputs = [child.communicate(timeout=timeout)[0] for child in children]
    assert all(child.returncode == 0 for child in children)
    return [[l for l in out.splitlines() if l.startswith('{')]
            for out in outputs]


@pytest.mark.timeout(1200)
def test_bench_strong_scaling_modes(tmp_path):
    """`bench.py --workload corpus | longform` (BASELINE configs[3] / [4]
    strong-scaled): the whole job sharded by dist.assign, both collectives
    inside the timed region.  One rank under nccl (= RCCL) and two gloo ranks
    sharing the box's one GPU must report the SAME job - same utterances,
    frames, scores and checksum (all scores on every rank, input order) -
    and the two-rank line names its shards."""
    import json
    common = ['--steps', '3', '--warmup', '1', '--regions', '2',
              '--no-cpu-baseline']
    for workload, extra, utterances in (
            ('corpus', ['--corpus-utterances', '600'], 600),
            ('longform', ['--longform-utterances', '4'], 4)):
        arguments = common + ['--workload', workload] + extra
        (one,) = _bench(arguments, 1)
        assert len(one) == 1
        one = json.loads(one[0])
        first, second = _bench(arguments, 2, 'gloo')
        assert len(first) == 1 and not second
        two = json.loads(first[0])
        for line, world in ((one, 1), (two, 2)):
            job = line['job']
            assert line['n_gpus'] == world and line['scaling'] == 'strong'
            assert line['unit'] == 'utterances/s' and line['steps'] == 3
            assert job['utterances'] == utterances
            assert len(job['frames_per_rank']) == world
            assert sum(job['frames_per_rank']) == job['frames']
            assert job['lpt_imbalance'] < 1.05
            assert abs(line['value'] - utterances /
                       (line['ms_per_step'] * 1e-3)) < 1e-6 * line['value']
            assert line['roofline']['frac'] > 0
            assert 'timed' in line['config']['exchange']
        assert two['job']['scores'] == one['job']['scores']
        assert two['job']['checksum'] == one['job']['checksum']   # bitwise


def test_bench_refuses_a_mismatched_world():
    """`--gpus` must be the size of the process group bench.py runs in."""
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    child = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
         '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True,
        text=True, timeout=300)
    assert child.returncode != 0
    assert 'process group of 1' in child.stderr


@pytest.mark.timeout(900)
def test_sharded_file_api(tmp_path):
    """dist.from_files_to_files (core.py:115-179 over a process group): 200
    WAV + TextGrid files (some at 8 and 22.05 kHz), two gloo ranks sharing the
    GPU and three ranks: every rank reads the samples of its own shard only,
    writes its own outputs, and both the files and the gathered scores are
    BITWISE those of one process (`emphases_amd.from_files_to_files` with the
    same pinned conv tile) - shards of 100 and of 67 files alike."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dist_worker
    from emphases_amd import dist as edist, load
    directory = tmp_path / 'corpus'
    directory.mkdir()
    dist_worker.write_files(str(directory), COUNT)
    texts, audios, prefixes = dist_worker.file_lists(
        str(directory), COUNT, 'single')
    emphases_amd.from_files_to_files(
        texts, audios, prefixes, gpu=0, conv_tile=edist.CONV_TILE)
    want = [torch.load(f'{prefix}.pt') for prefix in prefixes]
    assert all(w.dtype == torch.float32 and not w.is_cuda for w in want)
    headers = [load.wav_info(file) for file in audios]
    frames = [edist.frames_at_16k(samples, rate) for rate, _, samples in headers]
    for world in (2, 3):
        results = _launch('gloo', world, tmp_path, files=directory)
        shards = edist.assign(edist.cost(frames), world)
        for result, shard in zip(results, shards):
            assert result['read'] == sorted(f'u{i}.wav' for i in shard)
            assert len(result['scores']) == COUNT
            for got, expect in zip(result['scores'], want):
                assert torch.equal(got, expect)
        for index in range(COUNT):
            saved = torch.load(directory / f'w{world}_{index}.pt')
            assert torch.equal(saved, want[index])
            assert (directory / f'w{world}_{index}.TextGrid').exists()
    # the command line under two ranks (what `torchrun -m emphases_amd` starts):
    # every rank writes its own shard of the outputs
    few = 24
    port = _free_port()
    children = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0',
                   EMPHASES_DIST_BACKEND='gloo')
        children.append(subprocess.Popen(
            [sys.executable, '-m', 'emphases_amd', '--text_files', *texts[:few],
             '--audio_files', *audios[:few], '--output_prefixes',
             *[str(directory / f'cli_{i}') for i in range(few)]],
            env=env, cwd=ROOT))
    assert all(child.wait(timeout=600) == 0 for child in children)
    for index in range(few):
        assert torch.equal(torch.load(directory / f'cli_{index}.pt'), want[index])


@pytest.mark.timeout(1500)
def test_opt_in_precision_reaches_every_entry_point(tmp_path):
    """`precision='bf16x3'` through the reference's own entry points
    (`emphases/core.py:23-179`, `__main__.py:12-46`): the file API's scores are
    BITWISE the tensor API's at the same precision; two gloo ranks (tensors
    and files) are bitwise one process; the command line takes `--precision`;
    and on a slice of BASELINE configs[3] (2-30 s utterances) and a configs[4]
    long-form utterance (5 minutes, batch_size 3000) the scores stay within
    1e-5 of the f32 engine's."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dist_worker
    from emphases_amd import synth
    precision, count = 'bf16x3', 60
    aligns, audios = dist_worker.corpus(count, LOW, HIGH)
    plain = emphases_amd.from_alignments_and_audios(aligns, audios, gpu=0)
    want = emphases_amd.from_alignments_and_audios(
        aligns, audios, gpu=0, precision=precision)
    worst = max(float((a - b).abs().max()) for a, b in zip(plain, want))
    assert 0. < worst < 1e-5, worst
    # one utterance through from_alignment_and_audio: the same bits
    alone = emphases_amd.from_alignment_and_audio(
        aligns[7], audios[7], 16000, gpu=0, precision=precision)
    assert torch.equal(alone, want[7])
    # configs[4]: 5 minutes chunked at 3000 frames
    long_audio = torch.from_numpy(synth.audio(7100, 30000))
    long_words = emphases_amd.Alignment.from_frames(
        synth.word_frames(9000, 30000))
    a = emphases_amd.from_alignment_and_audio(
        long_words, long_audio, 16000, batch_size=3000, gpu=0)
    b = emphases_amd.from_alignment_and_audio(
        long_words, long_audio, 16000, batch_size=3000, gpu=0,
        precision=precision)
    assert a.shape == b.shape and 0. < float((a - b).abs().max()) < 1e-5
    # two ranks, tensors in memory
    results = _launch('gloo', 2, tmp_path, count=count, precision=precision)
    for result in results:
        for got, expect in zip(result['scores'], want):
            assert torch.equal(got, expect.cpu())
    # files: one process == from_file per file == two ranks == the command line
    directory = tmp_path / 'corpus'
    directory.mkdir()
    dist_worker.write_files(str(directory), count)
    texts, waves, prefixes = dist_worker.file_lists(
        str(directory), count, 'single')
    emphases_amd.from_files_to_files(
        texts, waves, prefixes, gpu=0, precision=precision)
    saved = [torch.load(f'{prefix}.pt') for prefix in prefixes]
    for index in (0, 3, 5, 16, count - 1):      # 16 kHz, 8 kHz, 22.05 kHz files
        one = emphases_amd.from_file(
            texts[index], waves[index], gpu=0, precision=precision)
        assert torch.equal(one.cpu(), saved[index])
    emphases_amd.from_file_to_file(
        texts[1], waves[1], str(directory / 'one'), gpu=0, precision=precision)
    assert torch.equal(torch.load(directory / 'one.pt'), saved[1])
    f32_prefixes = [str(directory / f'f32_{i}') for i in range(count)]
    emphases_amd.from_files_to_files(texts, waves, f32_prefixes, gpu=0)
    differences = [float((torch.load(f'{p}.pt') - s).abs().max())
                   for p, s in zip(f32_prefixes, saved)]
    assert 0. < max(differences) < 1e-5
    results = _launch('gloo', 2, tmp_path, count=count, files=directory,
                      precision=precision)
    for result in results:
        for got, expect in zip(result['scores'], saved):
            assert torch.equal(got, expect)
    for index in range(count):
        assert torch.equal(
            torch.load(directory / f'w2_{precision}_{index}.pt'), saved[index])
    few = 12
    child = subprocess.run(
        [sys.executable, '-m', 'emphases_amd', '--text_files', *texts[:few],
         '--audio_files', *waves[:few], '--output_prefixes',
         *[str(directory / f'cli_{i}') for i in range(few)], '--gpu', '0',
         '--precision', precision], cwd=ROOT, timeout=600)
    assert child.returncode == 0
    for index in range(few):
        assert torch.equal(
            torch.load(directory / f'cli_{index}.pt'), saved[index])
