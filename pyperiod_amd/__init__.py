"""pyperiod_amd -- MI355X-native periodicity transforms with the pyPeriod class surface.

    from pyperiod_amd import Periods, QOPeriods, RamanujanPeriods     # drop-in (reference __init__.py:1-3)
    from pyperiod_amd import PeriodEngine                            # batched (W, N) API

Importing the package does not touch the GPU; the first call creates the context on
cuda:LOCAL_RANK and fails loudly if libperiod_hip.so or a GPU is missing (no CPU fallback).
"""

from .Periods import Periods
from .QOPeriods import QOPeriods
from .RamanujanPeriods import RamanujanPeriods
from .engine import PeriodEngine, default_engine

__all__ = ["Periods", "QOPeriods", "RamanujanPeriods", "PeriodEngine", "default_engine"]
__version__ = "0.1.0"
