"""perform_raynet_fp -- HIP twin of cuda_implementations/raynet_fp.py:10-378."""
import numpy as np
import torch

from . import get_context, to_device


def perform_raynet_fp(M, D, N, F, H, W, padding, bbox, grid_shape, sampling_scheme):
    """Returns (raynet_fp, raynet_de): the K1 / K2 closures of the reference
    (raynet_fp.py:274-326 and :328-376), same argument order and assertions.

    Arrays may be NumPy arrays (uploaded, like `all_arrays_to_gpu`) or CUDA
    tensors (used in place).  `sampling_scheme` must be "sample_in_bbox"
    (the only scheme the reference's kernels implement).
    """
    if sampling_scheme != "sample_in_bbox":
        raise NotImplementedError(sampling_scheme)
    ctx = get_context(M, D, N, F, H, W, padding, bbox, grid_shape)
    grid_shape = tuple(int(g) for g in grid_shape)

    def _common(ray_idxs, features, P, P_inv, camera_center, voxel_grid, ray_voxel_indices,
                ray_voxel_count, S_voxel_space, acc, msgs):
        d = ctx.dev
        ray_idxs = d(ray_idxs, torch.int32)
        features = d(features, torch.float32)
        P = d(np.asarray(P, dtype=np.float32) if not isinstance(P, torch.Tensor) else P,
              torch.float32)
        P_inv, camera_center = d(P_inv, torch.float32), d(camera_center, torch.float32)
        rvi, rvc = d(ray_voxel_indices), d(ray_voxel_count)
        S_voxel_space, acc, msgs = d(S_voxel_space), d(acc), d(msgs)
        # raynet_fp.py:291-301
        assert S_voxel_space.shape[1] == M
        assert tuple(rvi.shape[1:]) == (M, 3)
        assert len(rvc.shape) == 1
        assert len(rvc) == len(S_voxel_space) == len(rvi)
        assert S_voxel_space.shape[1] == msgs.shape[1]
        assert tuple(acc.shape) == grid_shape
        assert torch.float32 == S_voxel_space.dtype
        assert torch.float32 == msgs.dtype
        assert torch.int32 == rvi.dtype
        assert torch.int32 == rvc.dtype
        assert features.numel() == int(np.prod(ctx.feature_shape))
        n = len(ray_idxs)
        assert n <= len(S_voxel_space) and n <= len(msgs)
        if not ctx._grid_set or getattr(ctx, "_grid_src", None) is not voxel_grid:
            ctx.set_voxel_grid(voxel_grid)
            ctx._grid_src = voxel_grid
        return ray_idxs, features, P, P_inv, camera_center, rvi, rvc, S_voxel_space, acc, msgs

    def raynet_fp(ray_idxs, features, P, P_inv, camera_center, voxel_grid, ray_voxel_indices,
                  ray_voxel_count, S_voxel_space, ray_to_occupancy_accumulated_pon,
                  ray_to_occupancy_messages_pon, ray_to_occupancy_accumulated_out_pon,
                  threads=2048):
        (ray_idxs, features, P, P_inv, camera_center, rvi, rvc, Sv, acc_in, msgs) = _common(
            ray_idxs, features, P, P_inv, camera_center, voxel_grid, ray_voxel_indices,
            ray_voxel_count, S_voxel_space, ray_to_occupancy_accumulated_pon,
            ray_to_occupancy_messages_pon)
        acc_out = ctx.dev(ray_to_occupancy_accumulated_out_pon)
        assert tuple(acc_out.shape) == grid_shape
        # the same buffer is msgs_in and msgs_out (raynet_fp.py:321-323)
        ctx.fused_bp_sweep(ray_idxs, features, P, P_inv, camera_center, rvi, rvc, Sv, acc_in, msgs,
                           acc_out, msgs)
        return msgs

    def raynet_de(ray_idxs, features, P, P_inv, camera_center, voxel_grid, ray_voxel_indices,
                  ray_voxel_count, S_voxel_space, ray_to_occupancy_accumulated_pon,
                  ray_to_occupancy_messages_pon, depth_map, threads=2048):
        (ray_idxs, features, P, P_inv, camera_center, rvi, rvc, Sv, acc, msgs) = _common(
            ray_idxs, features, P, P_inv, camera_center, voxel_grid, ray_voxel_indices,
            ray_voxel_count, S_voxel_space, ray_to_occupancy_accumulated_pon,
            ray_to_occupancy_messages_pon)
        depth_map = ctx.dev(depth_map)
        assert torch.float32 == depth_map.dtype and len(depth_map) >= len(ray_idxs)
        ctx.fused_depth(ray_idxs, features, P, P_inv, camera_center, rvi, rvc, Sv, acc, msgs,
                        depth_map)

    raynet_fp.context = ctx
    raynet_de.context = ctx
    return raynet_fp, raynet_de
