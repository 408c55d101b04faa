"""HIP twin of cuda_implementations/sample_points.py (K8)."""
import numpy as np
import torch

from . import get_context


def batch_sample_points(D, H, W, bbox, sampling_scheme):
    """sample_points.py:12-54 -> sp(ray_idxs, P_inv, camera_center, points)."""
    if sampling_scheme != "sample_in_bbox":
        raise NotImplementedError(sampling_scheme)
    ctx = get_context(1, D, 2, 1, H, W, 0, bbox, (1, 1, 1))

    def sp(ray_idxs, P_inv, camera_center, points, threads=2048):
        d = ctx.dev
        ray_idxs = d(ray_idxs, torch.int32)
        P_inv, camera_center, points = d(P_inv, torch.float32), d(camera_center, torch.float32), \
            d(points)
        assert points.dtype == torch.float32 and points.numel() >= len(ray_idxs) * D * 4
        ctx.sample_points(ray_idxs, P_inv, camera_center, points)
        return points

    sp.context = ctx
    return sp


def sample_points(ray_idxs, P_inv, camera_center, H, W, D, bbox, batch_size=100000,
                  sampling_scheme="sample_in_bbox"):
    """sample_points.py:57-91: returns points (4, n, D) float32 on the host."""
    sp = batch_sample_points(D, H, W, bbox, sampling_scheme)
    ctx = sp.context
    ray_idxs_gpu = ctx.dev(np.asarray(ray_idxs).astype(np.int32))
    n = len(ray_idxs_gpu)
    pts = torch.zeros((n, D, 4), dtype=torch.float32, device=ctx.device)
    sp(ray_idxs_gpu, np.asarray(P_inv, np.float32), np.asarray(camera_center, np.float32), pts)
    return pts.cpu().numpy().transpose(2, 0, 1)
