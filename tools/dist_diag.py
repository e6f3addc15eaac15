"""Diagnostic: two gloo ranks on one GPU - local scores before the gather and after."""
import os, subprocess, sys, socket, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))

def worker(out, mode):
    import dist_worker
    from emphases_amd import dist as edist, core
    rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    aligns, audios = dist_worker.corpus(200, 200, 3000)
    shards = edist.assign(edist.cost([a.shape[-1] // 160 for a in audios]), world)
    mine = shards[rank]
    result = {'mine': [int(i) for i in mine]}
    if mode == 'full_first':
        result['full'] = [s.cpu() for s in edist.from_alignments_and_audios(aligns, audios)]
    gpu = edist.local_device()
    local_dev = core.from_alignments_and_audios([aligns[i] for i in mine], [audios[i] for i in mine], 16000, None, None, gpu, None, conv_tile=64)
    if mode != 'nosnap':
        result['local'] = [s.cpu() for s in local_dev]
    gathered = edist.gather_scores([s.reshape(-1) for s in local_dev], shards)
    result['gathered'] = [s.cpu() for s in gathered]
    result['local_after'] = [s.cpu() for s in local_dev]
    result['full2'] = [s.cpu() for s in edist.from_alignments_and_audios(aligns, audios)]
    torch.distributed.destroy_process_group()
    torch.save(result, out)

if len(sys.argv) > 1:
    worker(sys.argv[1], sys.argv[2])
    sys.exit(0)

import emphases_amd
import dist_worker
def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); return s.getsockname()[1]
aligns, audios = dist_worker.corpus(200, 200, 3000)
whole = [s.cpu() for s in emphases_amd.from_alignments_and_audios(aligns, audios, gpu=0, conv_tile=64)]
def compare(a, b, tag):
    bad = [(i, float((x.reshape(-1) - y.reshape(-1)).abs().max())) for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x.reshape(-1), y.reshape(-1))]
    print(tag, 'mismatching', len(bad), 'of', len(a), bad[:4], flush=True)
for mode, world in (('snap', 2), ('nosnap', 2), ('full_first', 2)):
    tmp = tempfile.mkdtemp(); port = free_port(); children = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        out = os.path.join(tmp, f'{rank}.pt')
        children.append((rank, out, subprocess.Popen([sys.executable, os.path.abspath(__file__), out, mode], env=env, cwd=ROOT)))
    for rank, out, child in children:
        assert child.wait(timeout=600) == 0
        r = torch.load(out)
        for key in ('local', 'local_after'):
            if key in r:
                compare([whole[i] for i in r['mine']], r[key], f'{mode} rank {rank} {key}')
        for key in ('gathered', 'full', 'full2'):
            if key in r:
                compare(whole, r[key], f'{mode} rank {rank} {key}')
