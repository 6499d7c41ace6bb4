"""MI355X-native DualVGR reasoning unit (hot path of NJUPT-MCC/DualVGR-VideoQA).

`DualVGRUnit_multiple` is the drop-in module (reference: model/models.py:86-173); the arithmetic is
hand-written HIP for gfx950 behind the C ABI of include/dvgr.h (libdvgr_hip.so)."""
from . import synth  # noqa: F401
from . import dropout  # noqa: F401
from . import _lib  # noqa: F401
from . import config  # noqa: F401
from .unit import DualVGRUnit_multiple, param_table  # noqa: F401
from .head import AnswerHead  # noqa: F401
from .clip import ClipEncoder  # noqa: F401
from .inputs import HipLinear, WordFeatures  # noqa: F401
from .store import FeatureStore, VideoQAFlatLoader, write_flat  # noqa: F401
from .model import DualVGR, init_modules  # noqa: F401
from .trainer import ClipAdam, TrainStep  # noqa: F401

__version__ = "0.1.0"
