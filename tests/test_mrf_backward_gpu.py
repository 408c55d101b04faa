"""SURVEY.md 8(f) row 2 on the GPU: the HIP training forward and the analytic HIP backward
(csrc/raynet_train.inl through raynet_amd/mrf/mrf_train.py) against the float64 statement
oracle/mrf_backward.py, which tests/test_mrf_backward.py pins to finite differences."""
import numpy as np
import pytest

from test_mrf_backward import make_problem

pytestmark = pytest.mark.gpu


def _setup(oracle_mod, **kw):
    import torch
    from raynet_amd.mrf import mrf_train
    o, vg, starts, ends, rvi, rvc, S, planes = make_problem(oracle_mod, **kw)
    bbox = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    hip = mrf_train.training_context(rvi.shape[1], S.shape[1], bbox, o.grid_shape, vg)
    dev = dict(rvi=torch.from_numpy(rvi).cuda(), rvc=torch.from_numpy(rvc).cuda(),
               starts=torch.from_numpy(starts).cuda(), ends=torch.from_numpy(ends).cuda())
    return o, vg, starts, ends, rvi, rvc, S, planes, hip, dev


def test_plane_weights_match_oracle(oracle_mod):
    import torch
    from raynet_amd.mrf import mrf_train
    from oracle import mrf_backward as mb
    o, vg, starts, ends, rvi, rvc, S, planes, hip, dev = _setup(oracle_mod)
    left, c1, c2 = mrf_train.plane_weights(hip, dev["rvi"], dev["rvc"], dev["starts"], dev["ends"])
    left, c1, c2 = left.cpu().numpy(), c1.cpu().numpy(), c2.cpu().numpy()
    for r in range(len(rvc)):
        c = int(rvc[r])
        assert np.array_equal(left[r, :c], planes[r, :c])          # bit-exact index work
        assert not left[r, c:].any() and not c1[r, c:].any()
        if c == 0:
            continue
        _, e1, e2 = mb._interp_weights(vg[tuple(rvi[r, :c].T)].astype(np.float64),
                                       starts[r].astype(np.float64), ends[r].astype(np.float64),
                                       S.shape[1], planes[r, :c])
        assert np.abs(c1[r, :c] - e1).max() < 1e-4 and np.abs(c2[r, :c] - e2).max() < 1e-4
    # the differentiable mapping equals the inference kernel K6
    St = torch.from_numpy(S.astype(np.float32)).cuda()
    x, _ = mrf_train.planes_to_voxels(St, torch.from_numpy(left).cuda(),
                                      torch.from_numpy(c1).cuda(), torch.from_numpy(c2).cuda(),
                                      dev["rvc"])
    Sv = o.planes_to_voxels(vg, rvi, rvc, starts, ends, S.astype(np.float32))
    assert np.abs(x.cpu().numpy() - Sv).max() < 1e-6


@pytest.mark.parametrize("seed,iters", [(0, 3), (1, 3), (2, 1), (4, 0)])
def test_forward_and_backward_match_float64(oracle_mod, seed, iters):
    import torch
    from raynet_amd.mrf import mrf_train
    from oracle import mrf_backward as mb
    o, vg, starts, ends, rvi, rvc, S, planes, hip, dev = _setup(oracle_mod, seed=seed)
    rng = np.random.default_rng(50 + seed)
    G = rng.standard_normal((len(S), rvi.shape[1]))
    gamma = 0.05
    St = torch.from_numpy(S.astype(np.float32)).cuda().requires_grad_(True)
    gt = torch.tensor(gamma, dtype=torch.float32, device="cuda", requires_grad=True)
    out = mrf_train.mrf_depth_distribution(St, dev["rvi"], dev["rvc"], dev["starts"], dev["ends"],
                                           gt, iters, hip)
    (out * torch.from_numpy(G).float().cuda()).sum().backward()
    args = (vg, rvi, rvc, starts, ends, o.grid_shape)
    ref = mb.forward(S, *args, gamma=gamma, iters=iters, planes=planes)
    dS, prior_bar = mb.backward(G, S, *args, gamma=gamma, iters=iters, planes=planes,
                                with_prior=True)
    assert np.abs(out.detach().cpu().numpy() - ref).max() < 5e-5
    got = St.grad.cpu().numpy()
    scale = np.abs(dS).max()
    assert np.all(got[rvc <= 1] == 0)
    # fp32 backward of a chain conditioned like the forward
    assert np.abs(got - dS).max() < 2e-3 * scale, (np.abs(got - dS).max(), scale)
    dgamma = prior_bar * (1 / gamma + 1 / (1 - gamma))
    assert abs(float(gt.grad) - dgamma) < 2e-3 * max(1.0, abs(dgamma)), (float(gt.grad), dgamma)


def test_training_forward_equals_inference_kernels(oracle_mod):
    """rn_train_* on a pre-normalised column == K3/K4 on the raw column (they only differ in
    where clip_and_renorm happens)."""
    import torch
    from raynet_amd.mrf import mrf_train
    o, vg, starts, ends, rvi, rvc, S, planes, hip, dev = _setup(oracle_mod, seed=5)
    St = torch.from_numpy(S.astype(np.float32)).cuda()
    out = mrf_train.mrf_depth_distribution(St, dev["rvi"], dev["rvc"], dev["starts"], dev["ends"],
                                           0.05, 3, hip).cpu().numpy()
    Sv = o.planes_to_voxels(vg, rvi, rvc, starts, ends, S.astype(np.float32))
    acc, msgs = o.belief_propagation(Sv, rvi, rvc, np.zeros_like(Sv), gamma=0.05, bp_iterations=3)
    ref = o.depth_distribution(Sv, rvi, rvc, acc, msgs)
    assert np.abs(out - ref).max() < 5e-5


def test_end_to_end_training_step_reduces_loss():
    """forward_backward_pass (forward_backward_pass.py:128-248) end to end: CNN on patches ->
    similarities -> softmax -> MRF block -> squared EMD; a few Adam steps lower the loss and
    every parameter (and gamma) receives a finite gradient."""
    import torch
    from raynet_amd.hip_implementations.forward_backward_pass import forward_backward_pass
    from raynet_amd.mrf import mrf_train
    from raynet_amd.ray_marching.ray_tracing_hip import batch_voxel_traversal  # noqa: F401
    from raynet_amd.hip_implementations import get_context
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    n, D, M, views, grid = 64, 16, 48, 3, (16, 16, 16)
    bbox = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    axes = [np.linspace(bbox[i], bbox[i + 3], grid[i] + 1)[:-1] + (bbox[i + 3] - bbox[i]) /
            grid[i] / 2 for i in range(3)]
    vg = np.stack(np.meshgrid(*axes, indexing="ij"), -1).astype(np.float32)
    hip = mrf_train.training_context(M, D, bbox, grid, vg)
    starts = np.c_[rng.random((n, 2)) * 1.2 - 0.6, -np.ones(n)].astype(np.float32)
    ends = np.c_[rng.random((n, 2)) * 0.6 - 0.3, np.ones(n)].astype(np.float32)
    st, en = torch.from_numpy(starts).cuda(), torch.from_numpy(ends).cuda()
    rvi = torch.zeros((n, M, 3), dtype=torch.int32, device="cuda")
    rvc = torch.zeros((n,), dtype=torch.int32, device="cuda")
    hip.voxel_traversal(st, en, rvi, rvc)
    assert int(rvc.min()) > 1
    t = torch.linspace(0, 1, D, device="cuda")[None, :, None]
    points = torch.cat([st[:, None] + t * (en - st)[:, None],
                        torch.ones((n, D, 1), device="cuda")], -1)
    target = torch.zeros((n, M), device="cuda")
    target[torch.arange(n), (rvc // 2).long()] = 1.0
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.ReLU(),
                                torch.nn.Conv2d(8, 8, 3), torch.nn.Flatten()).cuda()
    images = [torch.randn((n, D, 3, 5, 5), device="cuda") for _ in range(views)]
    gamma = torch.tensor(0.031, device="cuda", requires_grad=True)
    opt = torch.optim.Adam(list(model.parameters()) + [gamma], lr=1e-3)
    cams = torch.cat([st, torch.ones((n, 1), device="cuda")], 1)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = forward_backward_pass(model, images, torch.from_numpy(vg).cuda(), rvi, rvc, target,
                                     points, cams, hip, views=views, gamma=gamma, bp_iterations=3,
                                     loss="squared_emd")
        loss.backward()
        for p in list(model.parameters()) + [gamma]:
            assert p.grad is not None and torch.isfinite(p.grad).all()
        opt.step()
        with torch.no_grad():
            gamma.clamp_(1e-3, 0.5)
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
