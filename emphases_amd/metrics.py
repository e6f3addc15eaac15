"""Word-level evaluation metrics on the device.

Same classes and call pattern as the reference's `emphases/evaluate/
metrics.py:12-110` — `Metrics(predicted_stats, target_stats)`, `.update(logits,
targets, word_lengths)`, `()` -> `{'pearson_correlation', 'bce', 'mse'}`,
`.reset()`, and `Statistics` — with the masked reductions done by
`emph_word_metrics` (csrc/metrics.hip) over the packed word axis, accumulated
in float64 on the device; one tiny D2H copy when the values are asked for.

The reference builds on `torchutil.metrics.{Average, MeanStd,
PearsonCorrelation}` (third-party, not under /root/reference, parity-unpinned),
restated here from their published definitions:
    Average             total / count
    MeanStd             mean and the (count - 1)-normalised standard deviation
    PearsonCorrelation  sum((p - mean_p)(t - mean_t)) / count / (std_p std_t)
"""
import math

import numpy as np
import torch

from . import config as cfg
from . import runtime


def _packed(values, lengths, device):
    """[B, 1, W] / [B, W] padded values -> (packed float32 [n], mask int32 [n])
    on the device: the batch rows back to back, padding columns masked out
    (`mask_from_lengths`, `model/core.py:146-149`)."""
    values = torch.as_tensor(values)
    if values.dim() == 3:
        values = values[:, 0]
    if values.dim() == 1:
        values = values[None]
    lengths = torch.as_tensor(lengths).reshape(-1).to(torch.int64)
    width = values.shape[-1]
    mask = (torch.arange(width)[None] < lengths.cpu()[:, None])
    mask = torch.where(mask, 0, -1).to(torch.int32)
    return (values.to(device, torch.float32).reshape(-1).contiguous(),
            mask.to(device).reshape(-1).contiguous())


class _Accumulator:
    """float64 [METRIC_FIELDS] sums on the device."""

    def __init__(self, gpu=None, loss=None):
        self.device = runtime.require_gpu(gpu)
        self.lib = runtime.library()
        self.post = runtime.POSTPROCESS[loss or cfg.DEFAULT.loss]
        self.predicted_mean = 0.
        self.target_mean = 0.
        self.reset()

    def reset(self):
        self.sums = torch.zeros(
            runtime.METRIC_FIELDS, dtype=torch.float64, device=self.device)

    def add(self, logits, targets, mask):
        """`logits`, `targets` float32 [n], `mask` int32 [n] (>= 0 = a word),
        all on the device - e.g. an engine's packed `logits` and the plan's
        `word_segment` table, without any gather."""
        with torch.cuda.device(self.device):
            runtime.check(self.lib.emph_word_metrics(
                logits.data_ptr(), targets.data_ptr(), mask.data_ptr(),
                logits.numel(), self.post, self.predicted_mean,
                self.target_mean, self.sums.data_ptr(), runtime.stream()),
                'emph_word_metrics')

    def values(self):
        return self.sums.cpu().numpy()


class Metrics:
    """`emphases/evaluate/metrics.py:12-51`"""

    def __init__(self, predicted_stats, target_stats, gpu=None, loss=None):
        self._sums = _Accumulator(gpu, loss)
        predicted_mean, self.predicted_std = predicted_stats()
        target_mean, self.target_std = target_stats()
        self._sums.predicted_mean = float(predicted_mean)
        self._sums.target_mean = float(target_mean)

    def __call__(self):
        sums = self._sums.values()
        count = sums[runtime.METRIC_COUNT]
        if not count:
            nan = float('nan')
            return {'pearson_correlation': nan, 'bce': nan, 'mse': nan}
        return {
            'pearson_correlation': float(
                sums[runtime.METRIC_COVARIANCE] / count /
                (self.predicted_std * self.target_std)),
            'bce': float(sums[runtime.METRIC_BCE] / count),
            'mse': float(sums[runtime.METRIC_SQUARED_ERROR] / count)}

    def update(self, logits, targets, word_lengths):
        """logits, targets [B, 1, W] (padded), word_lengths [B]"""
        packed, mask = _packed(logits, word_lengths, self._sums.device)
        target, _ = _packed(targets, word_lengths, self._sums.device)
        self._sums.add(packed, target, mask)

    def update_packed(self, logits, targets, word_segment):
        """The engine's own layout: packed device rows + the plan's
        `word_segment` table (no padding/unpadding round trip)."""
        self._sums.add(logits, targets, word_segment)

    def reset(self):
        self._sums.reset()


class Statistics:
    """`metrics.py:101-110` over `torchutil.metrics.MeanStd`: mean and
    standard deviation of the (masked) values; `()` -> (mean, std)."""

    def __init__(self, gpu=None):
        self._sums = _Accumulator(gpu, None)
        self._sums.post = runtime.POSTPROCESS[None]

    def update(self, values, lengths):
        packed, mask = _packed(values, lengths, self._sums.device)
        # the values ride in the `logits` slot (identity postprocess)
        self._sums.add(packed, packed, mask)

    def __call__(self):
        sums = self._sums.values()
        count = sums[runtime.METRIC_COUNT]
        if count < 1:
            return float('nan'), float('nan')
        mean = sums[runtime.METRIC_SUM_PREDICTED] / count
        if count < 2:
            return float(mean), float('nan')
        m2 = sums[runtime.METRIC_SUMSQ_PREDICTED] - count * mean * mean
        return float(mean), float(math.sqrt(max(m2, 0.) / (count - 1)))

    def reset(self):
        self._sums.reset()
