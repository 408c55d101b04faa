"""the analytic backward of mapping -> clip/renorm -> BP -> depth.
CPU part: the float64 statement (oracle/mrf_backward.py) against central finite
differences of its own forward, and its forward against the fp32 C oracle."""
import numpy as np
import pytest


def make_problem(oracle_mod, n=24, D=12, M=40, grid=(12, 12, 12), seed=0):
    rng = np.random.default_rng(seed)
    bbox = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    o = oracle_mod.Oracle(M=M, D=D, N=2, F=4, H=4, W=4, padding=3, bbox=bbox, grid_shape=grid)
    vg = oracle_mod.voxel_grid_centers(bbox, grid)
    # rays converging on a region so that they share voxels (coupled through the accumulator)
    starts = np.zeros((n, 3), np.float32)
    ends = np.zeros((n, 3), np.float32)
    for r in range(n):
        a = rng.integers(0, 3)
        p = rng.random(3) * 1.2 - 0.6
        q = rng.random(3) * 0.6 - 0.3
        p[a], q[a] = -1.0, 1.0
        starts[r], ends[r] = p, q
    starts[0] = ends[0] = 5.0                # outside the box: count 0 (skipped ray)
    rvi, rvc = o.traversal(starts, ends)
    S = rng.random((n, D)) ** 3 + 0.02
    S /= S.sum(1, keepdims=True)
    planes = o.plane_indices(vg, rvi, rvc, starts, ends)
    return o, vg, starts, ends, rvi, rvc, S, planes


def test_float64_forward_matches_fp32_oracle(oracle_mod):
    from oracle import mrf_backward as mb
    o, vg, starts, ends, rvi, rvc, S, planes = make_problem(oracle_mod)
    out = mb.forward(S, vg, rvi, rvc, starts, ends, o.grid_shape, planes=planes)
    Sv = o.planes_to_voxels(vg, rvi, rvc, starts, ends, S.astype(np.float32))
    msgs = np.zeros_like(Sv)
    acc, msgs = o.belief_propagation(Sv, rvi, rvc, msgs, gamma=0.05, bp_iterations=3)
    ref = o.depth_distribution(Sv, rvi, rvc, acc, msgs)
    assert np.abs(out - ref).max() < 5e-5
    assert np.all(out[rvc <= 1] == 0)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_analytic_backward_matches_finite_differences(oracle_mod, seed):
    from oracle import mrf_backward as mb
    o, vg, starts, ends, rvi, rvc, S, planes = make_problem(oracle_mod, seed=seed)
    rng = np.random.default_rng(100 + seed)
    G = rng.standard_normal((len(S), rvi.shape[1]))
    args = (vg, rvi, rvc, starts, ends, o.grid_shape)
    dS = mb.backward(G, S, *args, planes=planes)
    assert np.all(dS[rvc <= 1] == 0)

    def loss(Sx):
        return (G * mb.forward(Sx, *args, planes=planes)).sum()

    h = 1e-6
    worst = 0.0
    checked = 0
    for r in range(len(S)):
        if rvc[r] <= 1:
            continue
        for k in rng.choice(S.shape[1], 4, replace=False):
            Sp, Sm = S.copy(), S.copy()
            Sp[r, k] += h
            Sm[r, k] -= h
            fd = (loss(Sp) - loss(Sm)) / (2 * h)
            err = abs(fd - dS[r, k]) / max(1e-6, abs(fd), abs(dS[r, k]))
            worst = max(worst, err)
            checked += 1
    assert checked > 50 and worst < 1e-4, worst


def test_prior_gradient_matches_finite_differences(oracle_mod):
    from oracle import mrf_backward as mb
    o, vg, starts, ends, rvi, rvc, S, planes = make_problem(oracle_mod, seed=3)
    rng = np.random.default_rng(7)
    G = rng.standard_normal((len(S), rvi.shape[1]))
    args = (vg, rvi, rvc, starts, ends, o.grid_shape)
    gamma = 0.05
    _, prior_bar = mb.backward(G, S, *args, gamma=gamma, planes=planes, with_prior=True)
    dgamma = prior_bar * (1.0 / gamma + 1.0 / (1.0 - gamma))
    h = 1e-7
    fd = ((G * mb.forward(S, *args, gamma=gamma + h, planes=planes)).sum() -
          (G * mb.forward(S, *args, gamma=gamma - h, planes=planes)).sum()) / (2 * h)
    assert abs(fd - dgamma) < 1e-5 * max(1.0, abs(fd)), (fd, dgamma)


@pytest.mark.parametrize("seed,iters", [(0, 3), (1, 3), (2, 1), (5, 2), (4, 0)])
def test_analytic_backward_matches_an_independent_autograd_derivative(oracle_mod, seed, iters):
    """The second derivative check (the reference differentiates its TF graph by autodiff,
    tf_implementations/forward_backward_pass.py:194-246, mrf/mrf_tf.py:60-271; TF is absent): a
    float64 torch restatement of that graph, differentiated by torch.autograd
    (oracle/mrf_autograd.py), against the hand-derived reverse pass -- EVERY entry of dL/dS and
    dL/dgamma, not a sample of finite differences."""
    from oracle import mrf_autograd as ma
    from oracle import mrf_backward as mb
    o, vg, starts, ends, rvi, rvc, S, planes = make_problem(oracle_mod, seed=seed)
    rng = np.random.default_rng(200 + seed)
    G = rng.standard_normal((len(S), rvi.shape[1]))
    gamma = 0.05
    args = (vg, rvi, rvc, starts, ends, o.grid_shape)
    out_a, dS_a, dgamma_a = ma.gradients(G, S, *args, gamma=gamma, iters=iters, planes=planes)
    out_b = mb.forward(S, *args, gamma=gamma, iters=iters, planes=planes)
    dS_b, prior_bar = mb.backward(G, S, *args, gamma=gamma, iters=iters, planes=planes, with_prior=True)
    assert np.abs(out_a - out_b).max() < 1e-12
    scale = np.abs(dS_b).max()
    assert scale > 1e-3
    assert np.abs(dS_a - dS_b).max() <= 1e-9 * scale
    dgamma_b = prior_bar * (1.0 / gamma + 1.0 / (1.0 - gamma))
    assert abs(dgamma_a - dgamma_b) <= 1e-9 * max(1.0, abs(dgamma_b))
    # ... and with the plane indices derived in float64 by each side on its own
    out_c, dS_c, _ = ma.gradients(G, S, *args, gamma=gamma, iters=iters)
    dS_d = mb.backward(G, S, *args, gamma=gamma, iters=iters)
    assert np.abs(out_c - mb.forward(S, *args, gamma=gamma, iters=iters)).max() < 1e-12
    assert np.abs(dS_c - dS_d).max() <= 1e-9 * max(scale, np.abs(dS_d).max())
