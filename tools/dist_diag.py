"""Diagnostic: where do two gloo ranks sharing one GPU differ from one process?"""
import os, subprocess, sys, socket, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import emphases_amd
import dist_worker

def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); return s.getsockname()[1]

aligns, audios = dist_worker.corpus(200, 200, 3000)
runs = []
for tile in (None, 64, None):
    scores = emphases_amd.from_alignments_and_audios(aligns, audios, gpu=0, conv_tile=tile)
    runs.append([s.cpu() for s in scores])
def compare(a, b, tag):
    bad = [(i, float((x - y).abs().max())) for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]
    print(tag, 'mismatching utterances', len(bad), bad[:8], flush=True)
compare(runs[0], runs[1], 'single auto vs 64')
compare(runs[0], runs[2], 'single auto vs auto again')
# one utterance at a time
singles = [emphases_amd.from_alignments_and_audios([a], [x], gpu=0, conv_tile=64)[0].cpu() for a, x in list(zip(aligns, audios))[:40]]
compare(runs[1][:40], singles, 'batch(64) vs singles(64)')
for attempt in range(2):
    tmp = tempfile.mkdtemp()
    port = free_port()
    children = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        out = os.path.join(tmp, f'{rank}.pt')
        children.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'),
                                                'gloo', '200', '200', '3000', out], env=env, cwd=ROOT)))
    for out, child in children:
        assert child.wait(timeout=600) == 0
        result = torch.load(out)
        compare(runs[0], result['scores'], f'attempt {attempt} rank result vs single')
