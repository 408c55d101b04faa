"""HIP twin of raynet/mrf/mrf_cuda.py: K3/K4 closures and the batched drivers."""
import numpy as np
import torch

from ..hip_implementations import get_context


def batch_ray_belief_propagation(M, grid_shape):
    """mrf_cuda.py:12-124 -> (bp, de) closures with the reference's argument order."""
    ctx = get_context(M=M, grid_shape=grid_shape)
    grid_shape = tuple(int(g) for g in grid_shape)

    def _check(S, rvi, rvc, acc, msgs):
        # mrf_cuda.py:47-58
        assert S.shape[1] == M
        assert tuple(rvi.shape[1:]) == (M, 3)
        assert len(rvc.shape) == 1
        assert len(rvc) == len(S) == len(rvi)
        assert len(rvc) == len(msgs)
        assert S.shape[1] == msgs.shape[1]
        assert tuple(acc.shape) == grid_shape
        assert torch.float32 == S.dtype
        assert torch.float32 == msgs.dtype
        assert torch.int32 == rvi.dtype
        assert torch.int32 == rvc.dtype

    def bp(S, ray_voxel_indices, ray_voxel_count, ray_to_occupancy_accumulated_pon,
           ray_to_occupancy_messages_pon, ray_to_occupancy_accumulated_out_pon, threads=1024):
        d = ctx.dev
        S, rvi, rvc = d(S), d(ray_voxel_indices), d(ray_voxel_count)
        acc_in, msgs = d(ray_to_occupancy_accumulated_pon), d(ray_to_occupancy_messages_pon)
        acc_out = d(ray_to_occupancy_accumulated_out_pon)
        _check(S, rvi, rvc, acc_in, msgs)
        assert tuple(acc_out.shape) == grid_shape
        ctx.bp_sweep(S, rvi, rvc, acc_in, msgs, acc_out, msgs)   # msgs aliased, mrf_cuda.py:73-75
        return acc_out, msgs

    def de(S, ray_voxel_indices, ray_voxel_count, ray_to_occupancy_accumulated_pon,
           ray_to_occupancy_messages_pon, S_new, threads=1024):
        d = ctx.dev
        S, rvi, rvc = d(S), d(ray_voxel_indices), d(ray_voxel_count)
        acc, msgs, S_new = d(ray_to_occupancy_accumulated_pon), d(ray_to_occupancy_messages_pon), \
            d(S_new)
        _check(S, rvi, rvc, acc, msgs)
        assert S_new.shape[1] == M and torch.float32 == S_new.dtype
        ctx.depth_estimation(S, rvi, rvc, acc, msgs, S_new)
        return S_new

    bp.context = ctx
    de.context = ctx
    return bp, de


def belief_propagation(S, ray_voxel_indices, ray_voxel_count, ray_to_occupancy_messages_pon,
                       grid_shape, gamma=0.05, bp_iterations=3, batch_size=50000):
    """mrf_cuda.py:127-197.  Host arrays in; returns (accumulator ndarray, messages).
    Messages are zero-filled first, accumulators start at log(gamma/(1-gamma)),
    are swapped and the new `out` refilled with the prior after every iteration."""
    N, M = S.shape
    ray_to_occupancy_messages_pon.fill(0)
    bp, _ = batch_ray_belief_propagation(M, grid_shape)
    ctx = bp.context
    prior = float(np.float32(np.log(gamma) - np.log(1 - gamma)))
    shape = tuple(int(g) for g in grid_shape)
    acc = torch.full(shape, prior, dtype=torch.float32, device=ctx.device)
    acc_out = torch.full(shape, prior, dtype=torch.float32, device=ctx.device)
    S_d = ctx.dev(S)
    rvi_d = ctx.dev(ray_voxel_indices)
    rvc_d = ctx.dev(ray_voxel_count)
    msgs_d = ctx.dev(ray_to_occupancy_messages_pon)
    for it in range(bp_iterations):
        for i in range(0, N, batch_size):
            bp(S_d[i:i + batch_size], rvi_d[i:i + batch_size], rvc_d[i:i + batch_size], acc,
               msgs_d[i:i + batch_size], acc_out)
        acc_out, acc = acc, acc_out
        ctx.fill_f32(acc_out, prior)
    ray_to_occupancy_messages_pon[...] = msgs_d.cpu().numpy()
    return acc.cpu().numpy(), ray_to_occupancy_messages_pon


def compute_depth_distribution(S, ray_voxel_indices, ray_voxel_count,
                               ray_to_occupancy_messages_pon, ray_to_occupancy_accumulated_pon,
                               S_new, grid_shape, batch_size=50000):
    """mrf_cuda.py:200-251."""
    N, M = S.shape
    S_new.fill(0)
    _, de = batch_ray_belief_propagation(M, grid_shape)
    ctx = de.context
    S_d, rvi_d, rvc_d = ctx.dev(S), ctx.dev(ray_voxel_indices), ctx.dev(ray_voxel_count)
    msgs_d = ctx.dev(ray_to_occupancy_messages_pon)
    acc_d = ctx.dev(np.asarray(ray_to_occupancy_accumulated_pon, dtype=np.float32))
    out_d = ctx.dev(S_new)
    for i in range(0, N, batch_size):
        de(S_d[i:i + batch_size], rvi_d[i:i + batch_size], rvc_d[i:i + batch_size], acc_d,
           msgs_d[i:i + batch_size], out_d[i:i + batch_size])
    S_new[...] = out_d.cpu().numpy()
    return S_new
