"""cpu_tsdf_amd -- MI355X (gfx950) TSDF fusion behind the cpu_tsdf API.

The compute path is hand-written HIP in ``cpu_tsdf_amd/csrc`` reached through the C ABI declared in
``include/tsdf_hip.h`` (``cpu_tsdf_amd/lib/libtsdf_hip.so``).  This package is the Python-side host
mirror of the reference interface (``TSDFVolumeOctree`` / ``MarchingCubesTSDFOctree``), the Z-slab
multi-GPU layer (``torch.distributed`` over RCCL) and the synthetic scenes used by tests and
``bench.py``.  There is no CPU fallback: importing :mod:`cpu_tsdf_amd.capi` raises if the HIP
library has not been built (``python -c 'import __graft_entry__ as g; g.build()'``).
"""
__version__ = "0.1.0"
