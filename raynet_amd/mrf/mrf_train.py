"""Differentiable MRF block for training on MI355X.

The reference builds this block out of TensorFlow ops and differentiates it with autodiff
(raynet/mrf/mrf_tf.py:60-271 `belief_propagation`, `depth_estimate`, `clip_and_renorm`;
raynet/tf_implementations/forward_backward_pass.py:194-230).  Here the forward runs the same
HIP sweeps as inference and the backward is an analytic reverse sweep (one wavefront per ray,
`k_ray_bwd` in csrc/raynet_train.inl); the two dense, data-parallel steps in front of it
(planes->voxels interpolation and clip + renormalise) stay in torch so that autograd carries
the gradient on to the similarity scores and the MV-CNN.

    S [n, D] --planes_to_voxels--> x [n, M] --clip_and_renorm--> Sr [n, M]
      --MRFDepthDistribution (HIP fwd / HIP bwd)--> depth distribution [n, M]
"""
import numpy as np
import torch

from ..hip_implementations import get_context


def plane_weights(hip, ray_voxel_indices, ray_voxel_count, ray_start, ray_end):
    """(left [n,M] int64, c1 [n,M], c2 [n,M]) of planes_voxels_mapping.cu:48-84, zero
    beyond each ray's count."""
    n, M = ray_voxel_indices.shape[:2]
    dev = ray_voxel_indices.device
    left = torch.zeros((n, M), dtype=torch.int32, device=dev)
    c1 = torch.zeros((n, M), dtype=torch.float32, device=dev)
    c2 = torch.zeros((n, M), dtype=torch.float32, device=dev)
    hip.plane_weights(ray_voxel_indices, ray_voxel_count, ray_start.contiguous(),
                      ray_end.contiguous(), left, c1, c2)
    return left.long(), c1, c2


def planes_to_voxels(S, left, c1, c2, ray_voxel_count):
    """Differentiable statement of planes_voxels_mapping.cu:81-91 (and of
    single_ray_depth_to_voxels_map_li, forward_backward_pass.py:76-125)."""
    M = left.shape[1]
    mask = torch.arange(M, device=S.device)[None, :] < ray_voxel_count[:, None]
    z = (c1 * S.gather(1, left) + c2 * S.gather(1, left + 1)) * mask
    tot = z.sum(1, keepdim=True)
    return z / torch.where(tot > 0, tot, torch.ones_like(tot)), mask


def clip_and_renorm(x, mask, eps=1e-5):
    """mrf_tf.py clip_and_renorm / mrf_bp.cu:103-111 over the first `count` entries."""
    y = torch.clamp(x, float(np.float32(eps)), float(np.float32(1 - eps))) * mask
    tot = y.sum(1, keepdim=True)
    return y / torch.where(tot > 0, tot, torch.ones_like(tot))


class MRFDepthDistribution(torch.autograd.Function):
    """`bp_iterations` BP sweeps over the given rays + the per-ray depth distribution.

    forward(Sr, prior, rvi, rvc, hip, iters): Sr [n,M] clipped + renormalised; prior is a
    0-d tensor log(gamma) - log(1 - gamma) (differentiable, the reference can train gamma:
    forward_backward_pass.py:240-242)."""

    @staticmethod
    def forward(ctx, Sr, prior, rvi, rvc, hip, iters):
        Sr = Sr.contiguous()
        n, M = Sr.shape
        G = hip.G
        dev = Sr.device
        accs = [torch.full((G,), float(prior), dtype=torch.float32, device=dev)]
        msgs = [torch.zeros((n, M), dtype=torch.float32, device=dev)]
        for _ in range(iters):
            acc_out = torch.full((G,), float(prior), dtype=torch.float32, device=dev)
            m_out = torch.zeros((n, M), dtype=torch.float32, device=dev)
            hip.train_bp_sweep(Sr, rvi, rvc, accs[-1], msgs[-1], acc_out, m_out)
            accs.append(acc_out)
            msgs.append(m_out)
        out = torch.zeros((n, M), dtype=torch.float32, device=dev)
        hip.train_depth(Sr, rvi, rvc, accs[-1], msgs[-1], out)
        ctx.hip, ctx.iters, ctx.rvi, ctx.rvc = hip, iters, rvi, rvc
        ctx.accs, ctx.msgs = accs, msgs
        ctx.save_for_backward(Sr)
        return out

    @staticmethod
    def backward(ctx, g_out):
        Sr, = ctx.saved_tensors
        hip, rvi, rvc, accs, msgs = ctx.hip, ctx.rvi, ctx.rvc, ctx.accs, ctx.msgs
        n, M = Sr.shape
        dev = Sr.device
        g_out = g_out.contiguous().float()
        g_Sr = torch.zeros_like(Sr)
        g_acc = torch.zeros((hip.G,), dtype=torch.float32, device=dev)
        g_m = torch.zeros((n, M), dtype=torch.float32, device=dev)
        hip.train_depth_bwd(Sr, rvi, rvc, accs[-1], msgs[-1], g_out, g_Sr, g_acc, g_m)
        g_prior = g_acc.sum()
        for it in range(ctx.iters - 1, -1, -1):
            g_acc_prev = torch.zeros_like(g_acc)
            g_m_prev = torch.zeros_like(g_m)
            hip.train_bp_sweep_bwd(Sr, rvi, rvc, accs[it], msgs[it], g_m, g_acc, g_Sr,
                                   g_acc_prev, g_m_prev)
            g_acc, g_m = g_acc_prev, g_m_prev
            g_prior = g_prior + g_acc.sum()
        return g_Sr, g_prior, None, None, None, None


def mrf_depth_distribution(S, ray_voxel_indices, ray_voxel_count, ray_start, ray_end, gamma,
                           bp_iterations, hip):
    """S [n, D] (softmax output, requires_grad) -> depth distribution over the traversed
    voxels [n, M]; differentiable w.r.t. S and gamma."""
    assert S.is_cuda and S.dtype == torch.float32
    assert ray_voxel_indices.dtype == torch.int32 and ray_voxel_count.dtype == torch.int32
    left, c1, c2 = plane_weights(hip, ray_voxel_indices, ray_voxel_count, ray_start, ray_end)
    x, mask = planes_to_voxels(S, left, c1, c2, ray_voxel_count)
    Sr = clip_and_renorm(x, mask)
    if not torch.is_tensor(gamma):
        gamma = torch.tensor(float(gamma), dtype=torch.float32, device=S.device)
    prior = torch.log(gamma) - torch.log(1 - gamma)
    return MRFDepthDistribution.apply(Sr, prior, ray_voxel_indices, ray_voxel_count, hip,
                                      int(bp_iterations))


def training_context(M, D, bbox, grid_shape, voxel_grid):
    """HipContext for the training block: only M, D, the grid and the bbox matter here
    (no feature maps are gathered -- training works on per-ray patches,
    forward_backward_pass.py:176-183)."""
    hip = get_context(M, D, 2, 1, 1, 1, 0, bbox, grid_shape)
    hip.set_voxel_grid(voxel_grid)
    return hip
