"""Backend selector -- mirror of raynet/planes_voxels_mapping/depth_to_voxels.py:4-39."""
from .planes_voxels_mapping_hip import depth_to_voxels as depth_to_voxels_hip


def get_depth_to_voxels_backend(name, ray_voxel_count, ray_voxel_indices, rays_idxs, voxel_grid,
                                points, S, S_new=None, single_ray_depth_to_voxels=None,
                                gamma=None):
    """Like the reference this RUNS the mapping with the chosen backend and returns S_new."""
    if name == "hip":
        return depth_to_voxels_hip(ray_voxel_count, ray_voxel_indices, rays_idxs, voxel_grid,
                                   points, S, S_new)
    raise NotImplementedError(
        "backend %r: raynet_amd provides the 'hip' backend only (no CPU fallback)" % (name,))
