"""End-to-end rate of the public batch API on the bench workload (host planning,
H2D, kernels, D2H included): python tools/api_throughput.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import emphases_amd  # noqa: E402

audios, alignments, _ = bench.workload(0)
tensors = [torch.from_numpy(a) for a in audios]
for _ in range(3):
    emphases_amd.from_alignments_and_audios(alignments, tensors, 16000, gpu=0)
torch.cuda.synchronize()
start = time.perf_counter()
rounds = 10
for _ in range(rounds):
    scores = emphases_amd.from_alignments_and_audios(
        alignments, tensors, 16000, gpu=0)
torch.cuda.synchronize()
elapsed = (time.perf_counter() - start) / rounds
print(f'{elapsed * 1e3:.2f} ms per 64-utterance call = '
      f'{64 / elapsed:.0f} utterances/s through the public API')
profile = cProfile.Profile()
profile.enable()
emphases_amd.from_alignments_and_audios(alignments, tensors, 16000, gpu=0)
profile.disable()
pstats.Stats(profile).sort_stats('cumulative').print_stats(18)
