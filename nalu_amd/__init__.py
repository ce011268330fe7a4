"""nalu_amd -- MI355X-native drop-in for gantech/Nalu's edge-based CVFEM assembly and
Krylov solve (continuity / scalar transport systems).  The product is the HIP library
`libnalu_gpu.so` (csrc/, C ABI in include/nalu_gpu.h); this package is its thin loader and
the Python-side mirror of the reference's LinearSystem interface used by tests and bench."""
from ._lib import LIB_PATH, NaluGpuError, load  # noqa: F401
from .box import BoxSlab, box_edges, split_elems  # noqa: F401
from .linsys import GpuLinearSystem, LinearSolverConfig, suggest_edge_order  # noqa: F401
from . import eqsys  # noqa: F401
