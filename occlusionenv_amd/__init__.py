"""occlusionenv_amd -- MI355X-native OcclusionEnv step() hot path.

Public surface mirrors the reference modules (``environment``, ``SubProcVecEnv``, ``baseVecEnv``);
the arithmetic lives in hand-written HIP kernels behind the C ABI in include/occlusionenv_amd.h.
"""
from .meshes import MeshPool, SyntheticShapeNet, load_obj  # noqa: F401

__all__ = ["MeshPool", "SyntheticShapeNet", "load_obj"]
