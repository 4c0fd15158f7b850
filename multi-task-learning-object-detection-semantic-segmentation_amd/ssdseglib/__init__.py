"""ssdseglib -- MI355X-native drop-in for the hot path of matteo-stat's `ssdseglib` (same module and symbol
names as the reference package, reference __init__.py:1-9).  Importing the package never touches the GPU; the
HIP library (libssdseg_hip.so) is loaded on first use and its absence is an error, not a fallback."""
from . import boxes
from . import blocks
from . import layers
from . import models
from . import losses
from . import metrics
from . import datacoder
from . import evaluators
from . import optimizers
