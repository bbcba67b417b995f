"""adaflo_amd -- MI355X-native matrix-free operator engine for adaflo's hot path.

Only what the path needs: csrc/ (HIP kernels + C ABI), the ctypes binding and a
host-side mirror of the reference's operator interface (NavierStokesMatrix,
TimeStepping, FlowParameters).  There is no CPU fallback: without the HIP
library or without a GPU every operator raises.
"""
from .parameters import FlowParameters, flow_parameters_from_prm  # noqa: F401
from .time_stepping import TimeStepping  # noqa: F401
from .vectors import BlockVector, DeviceVector  # noqa: F401
from .navier_stokes_matrix import BrickMesh, NavierStokesMatrix  # noqa: F401
from .indexed_mesh import IndexedMesh, RefinedMesh  # noqa: F401
