"""lorenzcycletoolkit_amd -- MI355X-native Lorenz Energy Cycle engine.

Host side (Python, PyTorch-ROCm tensors) of a C-ABI HIP library (``liblec_hip.so``, gfx950) that
replaces the xarray/MetPy numerics of daniloceano/LorenzCycleToolkit's ``src/analysis`` and
``src/utils/box_data.py``.  See DESIGN.md and include/lec_hip.h.
"""
from .constants import CP_D, G, RD, RE  # noqa: F401

__all__ = ["G", "RE", "RD", "CP_D"]
__version__ = "0.1.0"
