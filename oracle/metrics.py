"""ORACLE — CPU restatement of the reference's evaluation metrics
(`emphases/evaluate/metrics.py:12-110`).  TEST INFRASTRUCTURE ONLY.

The reference's classes derive from `torchutil.metrics.{Average, MeanStd,
PearsonCorrelation}` (third-party, absent: PARITY UNPINNED there); their
published definitions are restated below.  First-party arithmetic (the mask,
`binary_cross_entropy_with_logits` vs the clamped log form, `mse_loss` of the
postprocessed logits) follows the reference line by line with the same ATen
ops and is PINNED: `tests/golden/generate.py` runs the reference's own
`Statistics` / `Metrics.update` on seeded ragged batches (with stand-in
running averages for the absent bases) and `tests/test_oracle.py` compares
per-word values bitwise and the results to 2e-6 (`tests/golden/metrics.npz`)."""
import math

import torch


def mask_from_lengths(lengths):
    """model/core.py:146-149"""
    x = torch.arange(lengths.max(), dtype=lengths.dtype)
    return (x.unsqueeze(0) < lengths.unsqueeze(1)).unsqueeze(1)


def postprocess(logits, loss='bce'):
    return torch.sigmoid(logits) if loss == 'bce' else \
        torch.clamp(logits, 0., 1.)


def word_values(logits, targets, word_lengths, loss='bce'):
    """(per-word BCE values, per-word squared errors) of the masked words:
    what `BinaryCrossEntropy.update` / `MeanSquaredError.update`
    (metrics.py:59-92) hand to their running averages."""
    mask = mask_from_lengths(word_lengths)                     # metrics.py:36
    logits, targets = logits[mask], targets[mask]
    if loss == 'bce':                                          # metrics.py:62-67
        values = torch.nn.functional.binary_cross_entropy_with_logits(
            logits, targets, reduction='none')
    else:                                                      # metrics.py:71-74
        x, y = torch.clamp(logits, 0., 1.), targets
        values = -(y * torch.log(x + 1e-6) +
                   (1 - y) * torch.log(1 - x + 1e-6))
    return values, torch.nn.functional.mse_loss(
        postprocess(logits, loss), targets, reduction='none')


class Metrics:
    """metrics.py:12-51 with torchutil's Average / PearsonCorrelation."""

    def __init__(self, predicted_stats, target_stats, loss='bce'):
        self.predicted_mean, self.predicted_std = predicted_stats
        self.target_mean, self.target_std = target_stats
        self.loss = loss
        self.reset()

    def reset(self):
        self.count = 0
        self.bce = 0.
        self.mse = 0.
        self.covariance = 0.

    def update(self, logits, targets, word_lengths):
        mask = mask_from_lengths(word_lengths)                 # metrics.py:36
        logits, targets = logits[mask], targets[mask]
        if self.loss == 'bce':                                 # metrics.py:62-67
            values = torch.nn.functional.binary_cross_entropy_with_logits(
                logits, targets, reduction='none')
        else:                                                  # metrics.py:71-74
            x, y = torch.clamp(logits, 0., 1.), targets
            values = -(y * torch.log(x + 1e-6) +
                       (1 - y) * torch.log(1 - x + 1e-6))
        scores = postprocess(logits, self.loss)
        self.bce += float(values.sum())
        self.mse += float(torch.nn.functional.mse_loss(
            scores, targets, reduction='none').sum())
        self.covariance += float((
            (scores - self.predicted_mean) *
            (targets - self.target_mean)).sum())
        self.count += logits.numel()

    def __call__(self):
        return {
            'pearson_correlation':
                self.covariance / self.count /
                (self.predicted_std * self.target_std),
            'bce': self.bce / self.count,
            'mse': self.mse / self.count}


def mean_std(values):
    """torchutil.metrics.MeanStd: mean, (n - 1)-normalised std."""
    values = [float(v) for v in values]
    count = len(values)
    mean = sum(values) / count
    return mean, math.sqrt(
        sum((v - mean) ** 2 for v in values) / (count - 1))
