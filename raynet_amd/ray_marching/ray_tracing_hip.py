"""HIP twin of raynet/ray_marching/ray_tracing_cuda.py (K5)."""
import numpy as np
import torch

from ..hip_implementations import get_context


def batch_voxel_traversal(M, bbox, grid_shape):
    """ray_tracing_cuda.py:12-63 -> vtr(points_start, points_end, rvi, rvc)."""
    ctx = get_context(M=M, bbox=bbox, grid_shape=grid_shape)

    def vtr(points_start, points_end, ray_voxel_indices, ray_voxel_count, threads=1024):
        d = ctx.dev
        ps, pe = d(points_start), d(points_end)
        rvi, rvc = d(ray_voxel_indices), d(ray_voxel_count)
        assert rvi.shape[1] == M
        assert torch.float32 == ps.dtype and torch.float32 == pe.dtype
        assert torch.int32 == rvi.dtype and torch.int32 == rvc.dtype
        assert len(ps) == len(pe) == len(rvc) and len(rvi) >= len(rvc)
        ctx.voxel_traversal(ps, pe, rvi, rvc)
        return rvi, rvc

    vtr.context = ctx
    return vtr


def voxel_traversal(bbox, grid_shape, ray_voxel_indices, ray_start, ray_end):
    """Single-ray call with the signature of ray_tracing.pyx:64 / ray_tracing_cuda.py:66-89:
    fills ray_voxel_indices [M, 3] in place, returns the count."""
    M = ray_voxel_indices.shape[0]
    vtr = batch_voxel_traversal(M, np.asarray(bbox, np.float32).ravel(), grid_shape)
    ctx = vtr.context
    rvi = ctx.dev(np.ascontiguousarray(ray_voxel_indices, dtype=np.int32).reshape(1, M, 3))
    rvc = torch.zeros((1,), dtype=torch.int32, device=ctx.device)
    vtr(np.asarray(ray_start, np.float32).reshape(1, 3),
        np.asarray(ray_end, np.float32).reshape(1, 3), rvi, rvc)
    ray_voxel_indices[:, :] = rvi.cpu().numpy()[0]
    return int(rvc.cpu().numpy()[0])


def perform_ray_marching(scene, img_idx, M, rays_idxs, grid_shape, batch_size=40000):
    """ray_tracing_cuda.py:92-143.  The entry/exit points come from sample_in_bbox on
    the device (the reference builds them with a TF graph, ray_marching.py:53-56).
    Like the Cython path (ray_marching.py:41-42) a ray that fills all M slots raises."""
    from ..hip_implementations.context import to_device
    H, W = scene.image_shape
    cam = scene.get_image(img_idx).camera
    ctx = get_context(M=M, H=H, W=W, bbox=scene.bbox.ravel(), grid_shape=grid_shape)
    n = len(rays_idxs)
    ridx = to_device(np.asarray(rays_idxs).astype(np.int32), device=ctx.device)
    starts = torch.zeros((n, 3), dtype=torch.float32, device=ctx.device)
    ends = torch.zeros((n, 3), dtype=torch.float32, device=ctx.device)
    ctx.sample_rays(ridx, ctx.dev(np.asarray(cam.P_pinv, np.float32)),
                    ctx.dev(np.asarray(cam.center, np.float32).ravel()), starts, ends)
    rvi = torch.zeros((n, M, 3), dtype=torch.int32, device=ctx.device)
    rvc = torch.zeros((n,), dtype=torch.int32, device=ctx.device)
    for r in range(0, n, batch_size):
        ctx.voxel_traversal(starts[r:r + batch_size], ends[r:r + batch_size],
                            rvi[r:r + batch_size], rvc[r:r + batch_size])
    rvc_h = rvc.cpu().numpy()
    if np.any(rvc_h >= M):
        raise ValueError("Nr=%d > M=%d" % (int(rvc_h.max()), M))
    return rvi.cpu().numpy(), rvc_h
