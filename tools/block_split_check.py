"""configs[2] (Transformer) at f32 and the opt-in precisions: ms per step, per-kernel us
(eager, one stream) and the worst score difference - the comparison the split
projection / block kernels (csrc/block_split.hip) are judged on.
usage (GPU box): python tools/block_split_check.py [f32 bf16x3 bf16x3_fast bf16x6]"""
import argparse
import json
import sys

sys.path.insert(0, '/root/repo')
import torch  # noqa: E402

import bench  # noqa: E402

device = torch.device('cuda', 0)
audios, alignments, bounds = bench.workload(0)
args = argparse.Namespace(steps=40)
precisions = sys.argv[1:] or ['f32', 'bf16x3']
baseline = None
for precision in precisions:
    line = bench.side_transformer(
        device, audios, alignments, args, precision=precision,
        baseline=baseline)
    if baseline is None:
        baseline = line['_scores']
    print(precision, 'ms_per_step %.4f' % line['ms_per_step'],
          'dscore', line.get('max_abs_dscore_vs_f32'))
    print(json.dumps({k: round(v, 1) for k, v in
                      line['kernels_us_per_step'].items()}, indent=0))
