"""Per-launch HBM traffic of the bench kernels from the two rocprofv3 --pmc
passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch) -> profiles/<tag>_pmc_summary.json

    python tools/pmc_summary.py profiles r1
"""
import collections
import csv
import json
import sys

FRAMES, WORDS, CHANNELS = 64000, 1882, 80
KERNELS = {
    # kernel function substring -> (bench.py name, algorithmic bytes per launch)
    # the frame-rate layers as three launches per step (3 + 2 + 2 layers): each
    # reads its input once; the first two (<false>) write their output, the last
    # one (<true>) leaves the running sums of the per-word sum (about two rows of
    # 80 floats per word); plus the packs of the launch's layers
    'conv1d_stack_kernel<false>': (
        'conv1d_stack_frames_80x80_k3',
        2 * CHANNELS * FRAMES * 4 + 5 * 153600 // 2),
    'conv1d_stack_kernel<true>': (
        'conv1d_stack_word_sums_80x80_k3',
        CHANNELS * FRAMES * 4 + 2 * WORDS * CHANNELS * 4 + 2 * 153600),
    'word_sums_kernel': (
        'word_sums', 3 * WORDS * CHANNELS * 4 + CHANNELS * WORDS * 4),
    'conv1d_winograd4_kernel': (
        'conv1d_winograd4_frames_80x80_k3',
        2 * CHANNELS * FRAMES * 4 + 153600),
    'conv1d_winograd_kernel': (
        'conv1d_winograd_frames_80x80_k3',
        2 * CHANNELS * FRAMES * 4 + 102400),
    'conv1d_kernel': (
        'conv1d_frames_80x80_k3', 2 * CHANNELS * FRAMES * 4 + 76800),
    'frontend_kernel': (
        'frontend_logmel', FRAMES * 160 * 4 + CHANNELS * FRAMES * 4),
    'segment_reduce_kernel': (
        'segment_reduce', CHANNELS * FRAMES * 4 + CHANNELS * WORDS * 4),
    'word_decoder_kernel': (
        'word_decoder', CHANNELS * WORDS * 4 + 6 * 76800 + 2 * WORDS * 4),
}
# --config transformer (BASELINE configs[2]): per launch of the frame-rate
# kernels (the word-rate launches of the same kernels are averaged in by
# rocprofv3, so the summary keeps the larger group only: see `select`)
TRANSFORMER = {
    'attention_group_kernel': (
        'attention_frames', 4 * CHANNELS * FRAMES * 4),      # Q, K, V read + O written
    'qkv_kernel': ('qkv_projection_frames', 4 * CHANNELS * FRAMES * 4 + 3 * 25600),
    # out_proj .. LayerNorm2 fused with the NEXT layer's Q / K / V projections
    # (layers 0-4): reads x and the attended values, writes x, Q, K and V
    'transformer_block_kernel<5, 2, true>': (
        'transformer_block_qkv_frames', 6 * CHANNELS * FRAMES * 4 + 6 * 25600),
    # the last layer's block: reads x and the attended values, writes x
    'transformer_block_kernel<5, 2, false>': (
        'transformer_block_frames', 3 * CHANNELS * FRAMES * 4 + 3 * 25600),
}


def means(path, kernels=None):
    """Mean counter value per dispatch; for kernels launched at two very
    different sizes (frame axis and word axis) the mean over the larger half."""
    kernels = KERNELS if kernels is None else kernels
    values = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        for key in kernels:
            if key in row['Kernel_Name']:
                values[key].append(float(row['Counter_Value']))
                break
    result = {}
    for key, series in values.items():
        series.sort()
        if series[-1] > 4 * max(series[0], 1e-9):
            series = series[len(series) // 2:]
        result[key] = sum(series) / len(series)
    return result


def main():
    directory, tag = sys.argv[1], sys.argv[2]
    if len(sys.argv) > 3 and sys.argv[3] == 'transformer':
        fetch = means(f'{directory}/{tag}_transformer_pmc_fetch_size.csv',
                      TRANSFORMER)
        write = means(f'{directory}/{tag}_transformer_pmc_write_size.csv',
                      TRANSFORMER)
        summary = {'_comment': (
            'HBM traffic per frame-rate launch on BASELINE configs[2] (64 x '
            '10 s, Transformer config) from rocprofv3 --pmc FETCH_SIZE / '
            'WRITE_SIZE (separate passes, KiB); traffic_bytes doubles '
            'FETCH_SIZE (gfx950 tallies 128-byte requests at 64 bytes).')}
        for key, (name, algorithmic) in TRANSFORMER.items():
            if key not in fetch:
                continue
            summary[name] = {
                'fetch_size_kib': round(fetch[key], 1),
                'write_size_kib': round(write.get(key, 0.), 1),
                'traffic_bytes_raw': int(
                    (fetch[key] + write.get(key, 0.)) * 1024),
                'traffic_bytes': int(
                    (2 * fetch[key] + write.get(key, 0.)) * 1024),
                'algorithmic_bytes': algorithmic}
        with open(f'{directory}/{tag}_transformer_pmc_summary.json', 'w') as f:
            json.dump(summary, f, indent=2)
        print(json.dumps(summary, indent=2))
        return
    fetch = means(f'{directory}/{tag}_bench_pmc_fetch_size.csv')
    write = means(f'{directory}/{tag}_bench_pmc_write_size.csv')
    summary = {'_comment': (
        'HBM traffic per launch on BASELINE configs[1] (64 x 10 s, conv '
        'config) from rocprofv3 --pmc, one counter per pass '
        f'(profiles/{tag}_bench_pmc_fetch_size.csv, _write_size.csv; bench.py '
        '--streams 1). FETCH_SIZE/WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE '
        'tallies 128-byte requests at 64 bytes for wide streaming reads '
        '(MI355X_MICROARCH.md, HBM); the conv kernel reads 16 bytes per lane '
        'at 8-byte granularity, for which the factor is uncalibrated, so both '
        'the raw and the doubled figure are kept and `traffic_bytes` uses the '
        'doubled (upper) one.')}
    for key, (name, algorithmic) in KERNELS.items():
        if key not in fetch:
            continue
        raw = (fetch[key] + write.get(key, 0.)) * 1024
        summary[name] = {
            'fetch_size_kib': round(fetch[key], 1),
            'write_size_kib': round(write.get(key, 0.), 1),
            'traffic_bytes_raw': int(raw),
            'traffic_bytes': int((2 * fetch[key] + write.get(key, 0.)) * 1024),
            'algorithmic_bytes': algorithmic}
    with open(f'{directory}/{tag}_pmc_summary.json', 'w') as file:
        json.dump(summary, file, indent=2)
    print(json.dumps(summary, indent=2))


if __name__ == '__main__':
    main()
