"""Command line of the reference (`emphases/__main__.py:12-50`): same flags."""
import argparse
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
    return parser.parse_args()


if __name__ == '__main__':
    emphases_amd.from_files_to_files(**vars(parse_args()))
