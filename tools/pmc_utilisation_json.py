"""profiles/<tag>_pmc_utilisation.txt (tools/pmc_round.sh) -> <tag>_pmc_utilisation.json,
the per-kernel figures bench.py quotes: matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES /
(4 x SQ_BUSY_CU_CYCLES) and the MFMA instructions per launch (SQ_INSTS_MFMA).

usage: python tools/pmc_utilisation_json.py profiles r5
"""
import json
import os
import re
import sys

NAMES = {'conv1d_stack_kernel': 'conv1d_stack_frames_80x80_k3',
         'attention_group_kernel': 'attention_frames',
         'frontend_kernel': 'frontend_logmel',
         'word_decoder_kernel': 'word_decoder',
         # the opt-in precision's kernels (<tag>_split_pmc.txt, tools/split_pmc.sh):
         # what bench.py looks up as '<timer name>@bf16x3'
         'attention_split_kernel': 'attention_frames@bf16x3',
         'conv1d_split_kernel': 'conv1d_split_frames_80x80_k3@bf16x3',
         'position_wise16_kernel': 'transformer_block_qkv_split_frames@bf16x3'}


def main(directory, tag):
    source = os.path.join(directory, f'{tag}_pmc_utilisation.txt')
    sections, current = {}, None
    for path in (source, os.path.join(directory, f'{tag}_split_pmc.txt')):
        if not os.path.exists(path):
            continue
        with open(path) as file:
            for line in file:
                header = re.match(r'== (\w+): (.*)', line)
                if header:
                    current = sections.setdefault(header.group(1), {})
                    continue
                row = re.match(r'(\w+)\s+([\d.]+)\s+\(x(\d+)\)', line)
                if row and current is not None:
                    current[row.group(1)] = float(row.group(2))
                    current['_launches_' + row.group(1)] = int(row.group(3))
    result = {'_comment': (
        f'per launch, mean over the launches of the pass, from {os.path.basename(source)} '
        f'(tools/pmc_round.sh {tag}); mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES '
        '/ (4 x SQ_BUSY_CU_CYCLES); the conv figures are means over the three '
        'launches of a step (3 + 2 + 2 layers)')}
    for kernel, counters in sections.items():
        entry = {}
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in counters and counters.get('SQ_BUSY_CU_CYCLES'):
            entry['mfma_pipe_busy'] = round(
                counters['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * counters['SQ_BUSY_CU_CYCLES']), 4)
        if 'SQ_INSTS_MFMA' in counters:
            entry['sq_insts_mfma_per_launch'] = counters['SQ_INSTS_MFMA']
        if 'FETCH_SIZE' in counters and 'WRITE_SIZE' in counters:
            # (KiB units; FETCH doubled on gfx950 per MI355X_MICROARCH.md)
            entry['traffic_bytes'] = int(
                (2 * counters['FETCH_SIZE'] + counters['WRITE_SIZE']) * 1024)
            entry['traffic_bytes_raw'] = int(
                (counters['FETCH_SIZE'] + counters['WRITE_SIZE']) * 1024)
        if entry:
            result[NAMES.get(kernel, kernel)] = entry
    target = os.path.join(directory, f'{tag}_pmc_utilisation.json')
    with open(target, 'w') as file:
        json.dump(result, file, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:3])
