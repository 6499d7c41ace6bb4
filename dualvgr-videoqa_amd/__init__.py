"""MI355X-native DualVGR reasoning unit (hot path of NJUPT-MCC/DualVGR-VideoQA)."""
from . import synth  # noqa: F401

__version__ = "0.1.0"
