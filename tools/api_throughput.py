"""End-to-end rate of the public batch API on the bench workload (host planning,
H2D, kernels, D2H included): python tools/api_throughput.py [--profile]"""
import cProfile
import json
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import emphases_amd  # noqa: E402

audios, alignments, _ = bench.workload(0)
print(json.dumps(bench.end_to_end_api(audios, alignments), indent=1))
if '--profile' in sys.argv:
    import torch
    tensors = [torch.from_numpy(a) for a in audios]
    profile = cProfile.Profile()
    profile.enable()
    for _ in range(5):
        emphases_amd.from_alignments_and_audios(
            alignments, tensors, 16000, gpu=0)
    profile.disable()
    pstats.Stats(profile).sort_stats('tottime').print_stats(25)
