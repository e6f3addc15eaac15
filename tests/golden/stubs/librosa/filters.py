"""`librosa.filters.mel` stand-in: the oracle's restatement of the algorithm."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), *['..'] * 4))
from oracle.librosa_mel import mel  # noqa: E402,F401
