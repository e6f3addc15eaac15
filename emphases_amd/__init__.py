"""emphases_amd — MI355X-native prominence inference.

The drop-in surface mirrors `import emphases` of the reference for its
inference hot path: `from_alignment_and_audio`, `from_file(s)_to_file(s)`,
the step functions `preprocess` / `infer` / `postprocess` / `downsample` and
`Model`; the hot path itself runs in libemphases_hip.so (include/emphases_hip.h).
"""
from .config import *  # noqa: F401,F403  (SAMPLE_RATE, HOPSIZE, Config, ...)
from .config import Config, DEFAULT  # noqa: F401
from . import alignment  # noqa: F401
from . import batch  # noqa: F401
from . import convert  # noqa: F401
from . import load  # noqa: F401
from . import melbasis  # noqa: F401
from . import runtime  # noqa: F401
from . import synth  # noqa: F401
from . import weights  # noqa: F401
from . import engine  # noqa: F401
from . import session  # noqa: F401
from . import metrics  # noqa: F401
from .alignment import Alignment, Word  # noqa: F401
from .core import (  # noqa: F401
    Model, active_config, configure, downsample, from_alignment_and_audio,
    from_alignments_and_audios, from_file, from_file_to_file,
    from_files_to_files, from_text_and_audio, get_engine, get_session, infer,
    inference_context, postprocess, preprocess, resample, segment)
# the torch.library operator seams (torch.ops.emphases_amd.*): registration
# only, nothing runs at import
from . import ops  # noqa: F401,E402
# emphases.data.preprocess.{from_audio, mels.from_audio, loudness.from_audio}
from . import data  # noqa: F401,E402
