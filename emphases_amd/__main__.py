"""Command line of the reference (`emphases/__main__.py:12-50`): same flags.

Under `torchrun` (WORLD_SIZE > 1: one process per GPU) the files are sharded
over the ranks by `dist.from_files_to_files`; every rank writes its own
outputs and `--gpu` is ignored (the rank's GPU is `LOCAL_RANK`):

    torchrun --standalone --nproc-per-node 8 -m emphases_amd \
        --text_files *.TextGrid --audio_files *.wav
"""
import argparse
import os
from pathlib import Path

import emphases_amd


def parse_args():
    parser = argparse.ArgumentParser(
        description='Determine which words in a speech file are emphasized')
    parser.add_argument(
        '--text_files', type=Path, nargs='+', required=True,
        help='The alignment (.TextGrid / .json) files')
    parser.add_argument(
        '--audio_files', type=Path, nargs='+', required=True,
        help='The corresponding speech audio files (.wav)')
    parser.add_argument(
        '--output_prefixes', type=Path, nargs='+', required=False,
        help='The output files. Defaults to text file stems.')
    parser.add_argument(
        '--checkpoint', type=Path,
        help='The model checkpoint to use for inference')
    parser.add_argument(
        '--batch_size', type=int,
        help='The maximum number of frames per batch')
    parser.add_argument(
        '--gpu', type=int, help='The index of the gpu to run inference on')
    # (an addition to the reference's flags)
    parser.add_argument(
        '--precision', default='f32',
        choices=sorted(emphases_amd.engine.PRECISIONS),
        help='f32 (default), or an opt-in precision: fp32 operands split '
             'into bf16 pieces on the bf16 matrix pipe, fp32 accumulation '
             '(scores within 1e-5 of f32)')
    return parser.parse_args()


def main():
    arguments = vars(parse_args())
    if int(os.environ.get('WORLD_SIZE', 1)) <= 1:
        emphases_amd.from_files_to_files(**arguments)
        return
    import torch
    from emphases_amd import dist
    backend = os.environ.get('EMPHASES_DIST_BACKEND', 'nccl')
    device = None
    if backend == 'nccl':
        device = torch.device('cuda', dist.local_device())
    torch.distributed.init_process_group(backend, device_id=device)
    try:
        arguments.pop('gpu')
        dist.from_files_to_files(**arguments, gather=False)
    finally:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
