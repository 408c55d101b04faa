"""Backend selector -- mirror of raynet/ray_marching/ray_marching.py:84-90."""
from .ray_tracing_hip import perform_ray_marching as perform_ray_marching_hip


def get_voxel_traversal_backend(name):
    if name == "hip":
        return perform_ray_marching_hip
    raise NotImplementedError(
        "backend %r: raynet_amd provides the 'hip' backend only (no CPU fallback)" % (name,))
