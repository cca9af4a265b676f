"""diffpir_amd -- MI355X-native DiffPIR sampling engine.

Host side (Python, mirroring the reference's operator / denoiser plug surface) over a C-ABI HIP
library (csrc/libdiffpir_hip.so, include/diffpir_engine.h).  See DESIGN.md and INTEGRATION.md.
"""
from .engine import Engine, DeviceArray, EngineError, EngineRangeError, default_engine  # noqa: F401
from ._lib import EngineLibraryError  # noqa: F401

__all__ = ["Engine", "DeviceArray", "EngineError", "EngineRangeError", "EngineLibraryError", "default_engine"]
