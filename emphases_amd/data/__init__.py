"""`emphases.data` of the reference, as far as the inference hot path reaches:
`emphases.data.preprocess` (`/root/reference/emphases/data/__init__.py`).
Datasets, loaders, samplers and downloads belong to training and are out of
scope."""
from . import preprocess  # noqa: F401
