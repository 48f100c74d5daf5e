"""MI355X-native sampling-MPC trajectory optimiser for nav2_sortham_controller.

The product is the C-ABI library built from csrc/ (include/smpc.h); this
package is its Python face: ctypes bindings (optimizer.Smpc), per-tick input
structs (tick), synthetic benchmark inputs (synthetic) and the batch-sharded
driver over torch.distributed (sharded).
"""
from . import _abi  # noqa: F401
from .tick import Tick, default_config, default_critics  # noqa: F401

__all__ = ["Tick", "default_config", "default_critics"]
