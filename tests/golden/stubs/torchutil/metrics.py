"""Stand-ins for `torchutil.metrics.{Average, MeanStd, PearsonCorrelation}`
(third-party, absent; written for this repo from their published definitions,
PARITY UNPINNED) so that `generate.py` can run the reference's own
`emphases/evaluate/metrics.py` — the first-party mask, BCE-with-logits /
clamped-log and MSE-of-postprocessed-logits arithmetic — on seeded batches."""
import math

import torch


class Average:
    """Running total / count."""

    def __init__(self):
        self.reset()

    def __call__(self):
        return float(self.total / self.count)

    def update(self, values, count):
        self.total = self.total + values.sum().double()
        self.count += count

    def reset(self):
        self.total = torch.zeros((), dtype=torch.float64)
        self.count = 0


class MeanStd:
    """Running mean and (n - 1)-normalised standard deviation of a list."""

    def __init__(self):
        self.reset()

    def __call__(self):
        return self.mean, math.sqrt(self.m2 / (self.count - 1))

    def update(self, values):
        for value in values:
            self.count += 1
            delta = value - self.mean
            self.mean += delta / self.count
            self.m2 += delta * (value - self.mean)

    def reset(self):
        self.count = 0
        self.mean = 0.
        self.m2 = 0.


class PearsonCorrelation:
    """sum((p - mean_p)(t - mean_t)) / count / (std_p std_t) with the means and
    standard deviations given up front."""

    def __init__(self, predicted_mean, predicted_std, target_mean, target_std):
        self.predicted_mean, self.predicted_std = predicted_mean, predicted_std
        self.target_mean, self.target_std = target_mean, target_std
        self.reset()

    def __call__(self):
        return float(
            self.total / self.count / (self.predicted_std * self.target_std))

    def update(self, predicted, target):
        self.total = self.total + (
            (predicted - self.predicted_mean) *
            (target - self.target_mean)).sum().double()
        self.count += predicted.numel()

    def reset(self):
        self.total = torch.zeros((), dtype=torch.float64)
        self.count = 0
