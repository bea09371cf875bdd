"""sparkfm_amd — MI355X-native Factorization Machines trainer behind SparkFM's API surface.

The arithmetic runs in hand-written HIP kernels (csrc/) behind a plain C ABI
(include/fmhip.h); this package is the thin host mirror of the reference's classes
(FM, FMModel, FMLearn, DataSet) over ctypes.  There is no CPU fallback.
"""
from .dataset import DataSet, Features  # noqa: F401
from .model import FMModel, Model  # noqa: F401
from .learn import FMLearn, HipALS, HipSGD  # noqa: F401
from .fm import FM, FactorizationMachines, Task  # noqa: F401
from .datacollection import DataCollection  # noqa: F401
from .feature_order import FeatureOrder  # noqa: F401
from . import fmutils as FMUtils  # noqa: F401

__all__ = ["DataSet", "Features", "FMModel", "Model", "FMLearn", "HipSGD", "HipALS", "FM", "FactorizationMachines", "Task",
           "DataCollection", "FMUtils", "FeatureOrder"]
