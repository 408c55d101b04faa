"""A SECOND, independent derivative of the differentiable MRF block: a float64 torch restatement of
the forward, op for op the graph the reference builds with TensorFlow ops and differentiates by
autodiff (raynet/mrf/mrf_tf.py:17-58 occupancy extraction, :60-142 one ray's sum-product sweep with
EXCLUSIVE cumprod / cumsum and a reversed exclusive cumsum, :145-172 depth estimate, :175-246 the
unrolled iterations with sparse adds into the accumulator; tf_implementations/
forward_backward_pass.py:194-230 mapping -> clip_and_renorm -> BP -> depth), differentiated by
torch.autograd.

TEST INFRASTRUCTURE ONLY (never imported by raynet_amd).  oracle/mrf_backward.py is a hand-derived
reverse pass checked against finite differences of its own forward; this file shares no line of
derivative code with it -- autograd differentiates the forward below -- so the two agreeing on every
entry of dL/dS and on dL/dgamma is an independent check of both (tests/test_mrf_backward.py), and
the HIP backward is held against them (tests/test_mrf_backward_gpu.py).  TensorFlow itself is
absent from the image: the reference's own graph cannot run.

Conventions kept from SURVEY.md section 9 (as in mrf_backward.py): the NumPy / CUDA clip_and_renorm
over the ray's own voxels (mrf_bp.cu:103-111; the TF variant's eps bookkeeping over the padded row,
mrf_tf.py:5-14, is the same quantity), rays with <= 1 voxels contribute nothing (mrf_np.py:300).
"""
import numpy as np
import torch

F64 = torch.float64


def _occupancy_to_ray(acc_on_ray, msgs):
    # mrf_tf.py:38-58: exp(-max) / exp(mu - max) trick, normalise, clip to [1e-4, 1 - 1e-4]
    mu = acc_on_ray - msgs
    mx = torch.clamp(mu, min=0.0)
    t1 = torch.exp(0.0 - mx)
    t2 = torch.exp(mu - mx)
    lo, hi = float(np.float32(1e-4)), float(np.float32(1 - 1e-4))
    return torch.clamp(t2 / (t1 + t2), lo, hi)


def _exclusive_cumprod(x):
    return torch.cat([torch.ones(1, dtype=x.dtype), torch.cumprod(x, 0)[:-1]])


def _exclusive_cumsum(x):
    return torch.cat([torch.zeros(1, dtype=x.dtype), torch.cumsum(x, 0)[:-1]])


def _ray_sweep(s, o):
    # mrf_tf.py:92-137
    neg_cumprod = _exclusive_cumprod(1.0 - o)
    common = neg_cumprod * s
    new_common = _exclusive_cumsum(o * common)
    positive = common + new_common
    t1 = torch.flip(_exclusive_cumsum(torch.flip(o * common, [0])), [0])
    negative = new_common + t1 / (1.0 - o)
    pos = positive / (positive + negative)
    return torch.log(pos) - torch.log(1.0 - pos)


def forward(S, voxel_centers, rvi, rvc, starts, ends, grid_shape, gamma, iters=3, planes=None):
    """S: [n, D] float64 tensor (may require grad); gamma: 0-d float64 tensor (may require grad).
    -> [n, M] depth distributions (zero beyond a ray's count and for rays with count <= 1)."""
    n, D = S.shape
    M = rvi.shape[1]
    G = int(np.prod(grid_shape))
    lo, hi = float(np.float32(1e-5)), float(np.float32(1 - 1e-5))
    step = 1.0 / (D - 1)
    rays = []
    for r in range(n):
        c = int(rvc[r])
        if c <= 1:
            rays.append(None)
            continue
        idx = np.asarray(rvi[r, :c], np.int64)
        flat = torch.from_numpy((idx[:, 0] * grid_shape[1] + idx[:, 1]) * grid_shape[2] + idx[:, 2])
        cen = torch.from_numpy(voxel_centers[tuple(idx.T)].astype(np.float64))
        st = torch.from_numpy(starts[r].astype(np.float64))
        ray = torch.from_numpy(ends[r].astype(np.float64)) - st
        t = torch.clamp(((cen - st) * ray).sum(1) / (ray * ray).sum(), 1e-4, 1 - 1e-4)
        if planes is None:
            left = torch.clamp(torch.ceil(t / step).to(torch.int64) - 1, 0, D - 2)
            left = torch.cummax(left, 0).values
        else:
            left = torch.from_numpy(np.asarray(planes[r, :c], np.int64))
        ld = torch.abs(t - left.to(F64) * step)
        rd = torch.abs(t - (left + 1).to(F64) * step)
        z = (1.0 - ld / (ld + rd)) * S[r, left] + (1.0 - rd / (ld + rd)) * S[r, left + 1]
        x = z / z.sum()
        y = torch.clamp(x, lo, hi)
        rays.append((flat, y / y.sum(), c))
    prior = torch.log(gamma) - torch.log(1.0 - gamma)
    acc = prior.expand(G)
    msgs = [torch.zeros(c, dtype=F64) if ray is not None else None for ray, c in
            ((ray, ray[2] if ray is not None else 0) for ray in rays)]
    for _ in range(iters):
        new_msgs, add = [], torch.zeros(G, dtype=F64)
        for r, ray in enumerate(rays):
            if ray is None:
                new_msgs.append(None)
                continue
            flat, s, c = ray
            m = _ray_sweep(s, _occupancy_to_ray(acc[flat], msgs[r]))
            new_msgs.append(m)
            add = add.index_add(0, flat, m)
        acc = add + prior          # mrf_tf.py:230-246: the messages' sum plus the prior
        msgs = new_msgs
    rows = []
    for r, ray in enumerate(rays):
        if ray is None:
            rows.append(torch.zeros(M, dtype=F64))
            continue
        flat, s, c = ray
        o = _occupancy_to_ray(acc[flat], msgs[r])
        P = o * _exclusive_cumprod(1.0 - o) * s                # mrf_tf.py:158-163
        rows.append(torch.cat([P / P.sum(), torch.zeros(M - c, dtype=F64)]))
    return torch.stack(rows)


def gradients(G, S, voxel_centers, rvi, rvc, starts, ends, grid_shape, gamma=0.05, iters=3,
              planes=None):
    """-> (out [n, M], dL/dS [n, D], dL/dgamma) for L = sum(G * out), by autograd (NumPy in / out)."""
    St = torch.tensor(np.asarray(S, np.float64), dtype=F64, requires_grad=True)
    gt = torch.tensor(float(gamma), dtype=F64, requires_grad=True)
    out = forward(St, voxel_centers, rvi, rvc, starts, ends, tuple(int(g) for g in grid_shape), gt,
                  iters, planes)
    (out * torch.from_numpy(np.asarray(G, np.float64))).sum().backward()
    return out.detach().numpy(), St.grad.numpy(), float(gt.grad)
