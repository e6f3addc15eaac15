"""`emphases.data.preprocess` (`/root/reference/emphases/data/preprocess/
__init__.py`): `from_audio` and the `mels` / `loudness` modules, on the HIP
front-end.  The dataset-level drivers (`datasets`, `from_files_to_files`,
which cache feature files for training) are out of scope."""
from .core import from_audio  # noqa: F401
from . import mels  # noqa: F401
from . import loudness  # noqa: F401
