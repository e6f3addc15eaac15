import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
import bench
from emphases_amd import config as cfg
import emphases_amd
device=torch.device('cuda',0)
audios, alignments, bounds = bench.workload(0)
config = cfg.Config(architecture='transformer')
state = emphases_amd.weights.random_state(config, seed=0)
def run(precision, streams, graph):
    r = bench.Runner(config, state, device, audios, alignments, streams=streams, graph=graph, precision=precision)
    outs=[]
    for _ in range(4):
        s = r.step()
        torch.cuda.synchronize()
        outs.append(s[r.columns].clone())
    return outs
base = run('f32',1,False)
print('f32 eager repeat', max(float((o-base[0]).abs().max()) for o in base))
for precision in ('f32','bf16x3','bf16x6'):
    for streams in (1,2):
        for graph in (False,True):
            outs = run(precision, streams, graph)
            print(precision, 'streams', streams, 'graph', graph, ['%.2e'%float((o-base[0]).abs().max()) for o in outs])
