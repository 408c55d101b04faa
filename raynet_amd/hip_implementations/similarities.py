"""HIP twins of cuda_implementations/similarities.py (K9, K10)."""
import numpy as np
import torch

from . import get_context


def perform_multi_view_cnn_forward_pass(D, N, F, H, W, padding, bbox, sampling_scheme):
    """similarities.py:11-130 -> closure mvcnnfp(ray_idxs, features, P, P_inv, camera_center, S)."""
    if sampling_scheme != "sample_in_bbox":
        raise NotImplementedError(sampling_scheme)
    ctx = get_context(1, D, N, F, H, W, padding, bbox, (1, 1, 1))

    def mvcnnfp(ray_idxs, features, P, P_inv, camera_center, S, threads=2048):
        d = ctx.dev
        ray_idxs, features = d(ray_idxs, torch.int32), d(features, torch.float32)
        P = d(np.asarray(P, dtype=np.float32) if not isinstance(P, torch.Tensor) else P,
              torch.float32)
        P_inv, camera_center, S = d(P_inv, torch.float32), d(camera_center, torch.float32), d(S)
        assert S.shape[1] == D and S.dtype == torch.float32       # similarities.py:112-113
        assert len(ray_idxs) <= len(S)
        ctx.mvcnn_similarities(ray_idxs, features, P, P_inv, camera_center, S)
        return S

    mvcnnfp.context = ctx
    return mvcnnfp


def perform_multi_view_cnn_forward_pass_with_depth_estimation(D, N, F, H, W, padding, bbox,
                                                              sampling_scheme):
    """similarities.py:133-285 -> closure(ray_idxs, features, P, P_inv, camera_center, S,
    points, depth_map)."""
    if sampling_scheme != "sample_in_bbox":
        raise NotImplementedError(sampling_scheme)
    ctx = get_context(1, D, N, F, H, W, padding, bbox, (1, 1, 1))

    def mvcnnfp(ray_idxs, features, P, P_inv, camera_center, S, points, depth_map, threads=2048):
        d = ctx.dev
        ray_idxs, features = d(ray_idxs, torch.int32), d(features, torch.float32)
        P = d(np.asarray(P, dtype=np.float32) if not isinstance(P, torch.Tensor) else P,
              torch.float32)
        P_inv, camera_center = d(P_inv, torch.float32), d(camera_center, torch.float32)
        S, points, depth_map = d(S), d(points), d(depth_map)
        assert S.shape[1] == D and S.dtype == torch.float32
        assert points.dtype == torch.float32 and depth_map.dtype == torch.float32
        n = len(ray_idxs)
        assert n <= len(S) and points.numel() >= n * D * 4 and len(depth_map) >= n
        ctx.mvcnn_depth(ray_idxs, features, P, P_inv, camera_center, S, points, depth_map)
        return depth_map

    mvcnnfp.context = ctx
    return mvcnnfp


def multi_view_cnn_fp(ray_idxs, features, P, P_inv, camera_center, bbox, S, padding,
                      batch_size=80000, sampling_scheme="sample_in_bbox"):
    """similarities.py:288-341: host arrays in, S [n, D] (NumPy) out."""
    _, D = S.shape
    N, Fh, Fw, F = features.shape
    H, W = Fh - padding - 1, Fw - padding - 1
    assert len(P) == N
    sim = perform_multi_view_cnn_forward_pass(D, N, F, H, W, padding, bbox, sampling_scheme)
    ctx = sim.context
    features_gpu = ctx.dev(features, torch.float32)
    ray_idxs_gpu = ctx.dev(np.asarray(ray_idxs).astype(np.int32))
    P_gpu = ctx.dev(np.array(P, dtype=np.float32))
    P_inv_gpu = ctx.dev(np.asarray(P_inv, dtype=np.float32))
    cc_gpu = ctx.dev(np.asarray(camera_center, dtype=np.float32))
    s_gpu = torch.zeros((batch_size, D), dtype=torch.float32, device=ctx.device)
    for i in range(0, len(ray_idxs_gpu), batch_size):
        chunk = ray_idxs_gpu[i:i + batch_size]
        sim(chunk, features_gpu, P_gpu, P_inv_gpu, cc_gpu, s_gpu)
        S[i:i + batch_size] = s_gpu[:len(chunk)].cpu().numpy()
    return S


def multi_view_cnn_fp_with_depth_estimation(ray_idxs, features, P, P_inv, camera_center, bbox,
                                            D, padding, H, W, batch_size=80000,
                                            sampling_scheme="sample_in_bbox"):
    """similarities.py:344-406: returns the (H, W) depth map (reshape(W, H).T)."""
    N, Fh, Fw, F = features.shape
    assert len(P) == N
    sim = perform_multi_view_cnn_forward_pass_with_depth_estimation(
        D, N, F, H, W, padding, bbox, sampling_scheme)
    ctx = sim.context
    features_gpu = ctx.dev(features, torch.float32)
    ray_idxs_gpu = ctx.dev(np.asarray(ray_idxs).astype(np.int32))
    P_gpu = ctx.dev(np.array(P, dtype=np.float32))
    P_inv_gpu = ctx.dev(np.asarray(P_inv, dtype=np.float32))
    cc_gpu = ctx.dev(np.asarray(camera_center, dtype=np.float32))
    s_gpu = torch.zeros((batch_size, D), dtype=torch.float32, device=ctx.device)
    points_gpu = torch.zeros((batch_size, D, 4), dtype=torch.float32, device=ctx.device)
    depth_map = torch.zeros((H * W,), dtype=torch.float32, device=ctx.device)
    for i in range(0, len(ray_idxs_gpu), batch_size):
        chunk = ray_idxs_gpu[i:i + batch_size]
        sim(chunk, features_gpu, P_gpu, P_inv_gpu, cc_gpu, s_gpu, points_gpu,
            depth_map[i:i + batch_size])
    return depth_map.cpu().numpy().reshape(W, H).T
