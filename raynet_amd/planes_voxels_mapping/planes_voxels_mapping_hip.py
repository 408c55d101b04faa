"""HIP twin of raynet/planes_voxels_mapping/planes_voxels_mapping_cuda.py (K6)."""
import numpy as np
import torch

from ..hip_implementations import get_context


def batch_depth_to_voxels_mapping(M, D, grid_shape, bbox=(0, 0, 0, 1, 1, 1)):
    """planes_voxels_mapping_cuda.py:11-67 -> pvm(voxel_grid, rvi, rvc, ray_start, ray_end, S, S_new)."""
    ctx = get_context(M=M, D=D, bbox=bbox, grid_shape=grid_shape)

    def pvm(voxel_grid, ray_voxel_indices, ray_voxel_count, ray_start, ray_end, S, S_new,
            threads=2048):
        d = ctx.dev
        rvi, rvc = d(ray_voxel_indices), d(ray_voxel_count)
        rs, re, S, S_new = d(ray_start), d(ray_end), d(S), d(S_new)
        # planes_voxels_mapping_cuda.py:38-47
        assert S.shape[1] == D
        assert S_new.shape[1] == M
        assert tuple(rvi.shape[1:]) == (M, 3)
        assert len(rvc) == len(S) == len(S_new) == len(rvi)
        assert torch.float32 == S.dtype and torch.float32 == S_new.dtype
        assert torch.int32 == rvi.dtype and torch.int32 == rvc.dtype
        if not ctx._grid_set or getattr(ctx, "_grid_src", None) is not voxel_grid:
            ctx.set_voxel_grid(voxel_grid)
            ctx._grid_src = voxel_grid
        ctx.planes_to_voxels(rvi, rvc, rs, re, S, S_new)
        return S_new

    pvm.context = ctx
    return pvm


def depth_to_voxels(ray_voxel_count, ray_voxel_indices, rays_idxs, voxel_grid, points, S, S_new,
                    batch_size=20000):
    """planes_voxels_mapping_cuda.py:70-124, same arguments:
        voxel_grid [3, gx, gy, gz]; points [4, N, D]; S [N, D]; S_new [N, M] (zero-filled
        here, rows of `rays_idxs` written).  Ray start/end are points[:, r, 0] / [:, r, -1]."""
    N, M, _ = ray_voxel_indices.shape
    _, _, D = points.shape
    S_new.fill(0)
    rays_idxs = np.asarray(rays_idxs)
    pvm = batch_depth_to_voxels_mapping(M, D, np.array(voxel_grid.shape[1:]))
    ctx = pvm.context
    vg = ctx.dev(np.ascontiguousarray(voxel_grid.transpose(1, 2, 3, 0), dtype=np.float32))
    rs = ctx.dev(np.ascontiguousarray(points[:-1, rays_idxs, 0].T, dtype=np.float32))
    re = ctx.dev(np.ascontiguousarray(points[:-1, rays_idxs, -1].T, dtype=np.float32))
    rvc = ctx.dev(np.ascontiguousarray(ray_voxel_count[rays_idxs], dtype=np.int32))
    for i in range(0, len(rays_idxs), batch_size):
        sel = rays_idxs[i:i + batch_size]
        out = torch.zeros((len(sel), M), dtype=torch.float32, device=ctx.device)
        pvm(vg, np.ascontiguousarray(ray_voxel_indices[sel]), rvc[i:i + batch_size],
            rs[i:i + batch_size], re[i:i + batch_size], np.ascontiguousarray(S[sel]), out)
        S_new[sel] = out.cpu().numpy()
    return S_new
