"""Placeholder for import; resampling/loading are not exercised by goldens."""
from . import transforms


def load(file):
    raise NotImplementedError


def info(file):
    raise NotImplementedError
