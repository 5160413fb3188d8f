"""taichi_mpm_amd — MI355X-native MLS-MPM time-stepping core behind the taichi_mpm MPM surface.

The compute path is libmpmhip.so (hand-written HIP for gfx950, C ABI in include/mpmhip.h); this package
is the thin host-side mirror of the reference's Python-facing interface (`tc.dynamics.MPM`,
scripts/async/async_mpm.py:17-300 for its shape).  There is NO CPU fallback: importing the simulation
classes without the built library, or creating a simulation without a GPU, fails loudly.
"""
from ._lib import build, lib_path, load  # noqa: F401
from .materials import MATERIAL_IDS, group_params, initial_aux  # noqa: F401
from .mpm import MPM, AsyncMPM, MPMError, Simulation3D, create_simulation2, create_simulation3  # noqa: F401
from .mpm2d import Simulation2D  # noqa: F401
from .mpm88 import MPM88  # noqa: F401

__all__ = ["MPM", "AsyncMPM", "MPM88", "MPMError", "Simulation2D", "Simulation3D", "create_simulation2", "create_simulation3", "build", "load", "lib_path",
           "group_params", "initial_aux", "MATERIAL_IDS"]
