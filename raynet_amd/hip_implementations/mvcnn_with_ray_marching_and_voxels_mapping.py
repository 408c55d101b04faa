"""HIP twins of cuda_implementations/mvcnn_with_ray_marching_and_voxels_mapping.py (K11, K12)."""
import numpy as np
import torch

from . import get_context


def _factory(M, D, N, F, H, W, padding, bbox, grid_shape, sampling_scheme, with_depth):
    if sampling_scheme != "sample_in_bbox":
        raise NotImplementedError(sampling_scheme)
    ctx = get_context(M, D, N, F, H, W, padding, bbox, grid_shape)

    def fp(ray_idxs, features, P, P_inv, camera_center, voxel_grid, ray_voxel_indices,
           ray_voxel_count, S_new, depth_map=None, threads=2048):
        d = ctx.dev
        ray_idxs, features = d(ray_idxs, torch.int32), d(features, torch.float32)
        P = d(np.asarray(P, dtype=np.float32) if not isinstance(P, torch.Tensor) else P,
              torch.float32)
        P_inv, camera_center = d(P_inv, torch.float32), d(camera_center, torch.float32)
        rvi, rvc, S_new = d(ray_voxel_indices), d(ray_voxel_count), d(S_new)
        # mvcnn_with_ray_marching_and_voxels_mapping.py:150-158
        assert S_new.shape[1] == M
        assert tuple(rvi.shape[1:]) == (M, 3)
        assert len(rvc.shape) == 1
        assert len(rvc) == len(S_new) == len(rvi)
        assert torch.float32 == S_new.dtype
        assert torch.int32 == rvi.dtype and torch.int32 == rvc.dtype
        if not ctx._grid_set or getattr(ctx, "_grid_src", None) is not voxel_grid:
            ctx.set_voxel_grid(voxel_grid)
            ctx._grid_src = voxel_grid
        if with_depth:
            depth_map = d(depth_map)
            assert depth_map.dtype == torch.float32 and len(depth_map) >= len(ray_idxs)
        ctx.mvcnn_voxel_space(ray_idxs, features, P, P_inv, camera_center, rvi, rvc, S_new,
                              depth_map if with_depth else None)
        return S_new

    fp.context = ctx
    return fp


def batch_mvcnn_voxel_traversal_with_ray_marching(M, D, N, F, H, W, padding, bbox, grid_shape,
                                                  sampling_scheme):
    """:11-174 (K11)."""
    return _factory(M, D, N, F, H, W, padding, bbox, grid_shape, sampling_scheme, False)


def batch_mvcnn_voxel_traversal_with_ray_marching_with_depth_estimation(
        M, D, N, F, H, W, padding, bbox, grid_shape, sampling_scheme):
    """:177-378 (K12); the closure takes the extra depth_map argument."""
    return _factory(M, D, N, F, H, W, padding, bbox, grid_shape, sampling_scheme, True)
