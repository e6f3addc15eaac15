"""Immutable configuration of the prominence-inference path.

Names mirror the module-level globals of the reference
(`emphases/config/defaults.py:47-233`, derived values `config/static.py:27-46`)
so that experiment overrides written for the reference (`config/**/*.py`) map
one-to-one onto keyword arguments of :class:`Config`.  The reference mutates
module globals through `yapecs`; here the configuration is an explicit frozen
object handed to the operator layer.
"""
import dataclasses
import math

# Audio constants (defaults.py:47-74)
SAMPLE_RATE = 16000
HOPSIZE = 160
WINDOW_SIZE = 1024
NUM_FFT = 1024
NUM_MELS = 80
NUM_BINS = NUM_FFT // 2 + 1
MIN_DB = -100.
REF_DB = 20.
FMIN = 40.
FMAX = 550.
HOPSIZE_SECONDS = HOPSIZE / SAMPLE_RATE          # static.py:30
LOGFMIN = math.log2(FMIN)                        # static.py:36
LOGFMAX = math.log2(FMAX)                        # static.py:33

# Padding applied by `preprocess` and again by `mels.from_audio`
# (core.py:357, mels.py:31): int((1024 - 160) / 2)
PADDING = (WINDOW_SIZE - HOPSIZE) // 2

# Transformer positional-encoding table length (transformer.py:40)
MAX_POSITIONS = 5000

ACTIVATIONS = ('relu', 'gelu', 'silu', 'leaky_relu')
ARCHITECTURES = ('convolution', 'transformer')
DOWNSAMPLE_LOCATIONS = ('input', 'intermediate', 'inference', 'loss')
DOWNSAMPLE_METHODS = ('sum', 'average', 'max', 'center')
LOSSES = ('bce', 'mse')


@dataclasses.dataclass(frozen=True)
class Config:
    """Model/feature switches (defaults = `emphases/config/defaults.py`)."""
    # Features (defaults.py:89-113)
    mel_feature: bool = True
    pitch_feature: bool = False
    periodicity_feature: bool = False
    loudness_feature: bool = False
    normalize: bool = False
    # Model (defaults.py:181-215)
    architecture: str = 'convolution'
    activation: str = 'relu'
    channels: int = 80
    layers: int = 6
    encoder_kernel_size: int = 3
    decoder_kernel_size: int = 3
    downsample_location: str = 'intermediate'
    downsample_method: str = 'sum'
    # Postprocess switch (defaults.py:227)
    loss: str = 'bce'
    # Transformer constants (transformer.py:18-23)
    heads: int = 2
    layer_norm_eps: float = 1e-5

    def __post_init__(self):
        if self.architecture not in ARCHITECTURES:
            # layers/__init__.py:13-14
            raise ValueError(
                f'Network layer {self.architecture} is not defined')
        if self.activation not in ACTIVATIONS:
            raise ValueError(f'Activation {self.activation} is not defined')
        if self.downsample_location not in DOWNSAMPLE_LOCATIONS:
            # model/core.py:127-130
            raise ValueError(
                f'Downsample location {self.downsample_location} '
                'not recognized')
        if self.downsample_method not in DOWNSAMPLE_METHODS:
            # core.py:468-469
            raise ValueError(
                f'Interpolation method {self.downsample_method} '
                'is not defined')
        if self.loss not in LOSSES:
            raise ValueError(f'Loss {self.loss} is not defined')
        for k in (self.encoder_kernel_size, self.decoder_kernel_size):
            if k % 2 != 1 or not 1 <= k <= 7:
                raise ValueError('kernel sizes must be odd and in 1..7')
        if self.architecture == 'transformer' and \
                self.channels % self.heads != 0:
            raise ValueError('channels must be divisible by heads')

    @property
    def num_features(self):
        """static.py:42-46"""
        return (
            int(self.mel_feature) * NUM_MELS + int(self.pitch_feature) +
            int(self.periodicity_feature) + int(self.loudness_feature))

    @property
    def has_decoder(self):
        """model/core.py:28-30"""
        return self.downsample_location in ('input', 'intermediate')


DEFAULT = Config()
