"""The cooperative plane sweep for D <= 32 (k_sweep_map_packed: two rays per wavefront for
D <= 32 -- the reference's own default, scripts/arguments.py:154 -- four for D <= 16) against
the one-ray-per-wavefront kernel it replaces there (rn_options.sweep_rays_per_wave = 1: the SAME
bits are required, the per-sample instruction sequence is the same) and against the oracle, through
every entry that sweeps: K7, K9 / K10, K11, the resident prepare (with and without the folded first
BP iteration) and the whole forward pass.  Odd ray counts (a wavefront's last ray slots empty),
D not a power of two (rounds of dead samples skipped), 2 ... 9 views."""
import numpy as np
import pytest

from conftest import load_cases

pytestmark = pytest.mark.gpu

CU = load_cases("crosscheck_cu_host.npz")


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (no CPU fallback exists)"
    from raynet_amd import _lib
    _lib.build()
    return torch


def _case(oracle_mod, D, N, M=96, n_rays=None):
    """The `wide` geometry (30 x 40 image, 32^3 grid, F = 32) with D planes and N views; views
    beyond the fixture's five are its cameras with slightly different matrices."""
    c = CU["wide"]
    rng = np.random.default_rng(1000 * D + N)
    P = np.array(c["P"], np.float32)
    while len(P) < N:
        P = np.concatenate([P, P[1:] * (1 + 0.01 * rng.standard_normal(P[1:].shape)).astype(np.float32)])
    P = np.ascontiguousarray(P[:N])
    o = oracle_mod.Oracle(M=M, D=D, N=N, F=32, H=int(c["H"]), W=int(c["W"]),
                          padding=int(c["padding"]), bbox=c["bbox"], grid_shape=c["grid"])
    feats = rng.standard_normal((N, o.H + o.padding + 1, o.W + o.padding + 1, 32),
                                dtype=np.float32) * np.float32(0.25)
    ridx = np.array(c["ray_idxs"], np.int32)
    if n_rays is not None:
        ridx = ridx[:n_rays]
    vg = oracle_mod.voxel_grid_centers(c["bbox"], c["grid"])
    return c, o, P, feats, ridx, vg


def _ctx(o, rays_per_wave):
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.hip_implementations.options import PathOptions
    ctx = get_context(o.M, o.D, o.N, o.F, o.H, o.W, o.padding, o.bbox, o.grid_shape)
    ctx.set_options(PathOptions(sweep_rays_per_wave=rays_per_wave))
    assert ctx.get_options()["sweep_rays_per_wave"] == rays_per_wave
    return ctx


@pytest.mark.parametrize("D,N", [(32, 5), (32, 2), (32, 9), (16, 5), (16, 3), (20, 5), (27, 4),
                                 (9, 5), (2, 5), (12, 7), (31, 6), (17, 8)])
def test_k7_columns_are_the_same_bits_and_meet_the_oracle(torch, oracle_mod, D, N):
    c, o, P, feats, ridx, _ = _case(oracle_mod, D, N)
    n = len(ridx) - 1 if len(ridx) % 4 == 0 else len(ridx)       # a wavefront with empty ray slots
    s, e = o.sample(ridx[:n], c["P_inv"], c["center"])
    out = {}
    for mode in (1, 0):
        ctx = _ctx(o, mode)
        S = torch.full((n, D), -7.0, device="cuda")
        ctx.compute_similarities(ctx.dev(feats), ctx.dev(P), ctx.dev(s), ctx.dev(e), S)
        out[mode] = S.cpu().numpy()
    assert np.array_equal(out[0], out[1])
    So = o.similarities(feats, P, s, e)
    assert np.abs(out[0] - So).max() <= 1e-5
    assert np.abs(out[0].sum(1) - 1).max() < 1e-5


@pytest.mark.parametrize("D,N", [(32, 5), (16, 5), (24, 3)])
def test_k10_points_depth_and_k11_columns(torch, oracle_mod, D, N):
    """K9 / K10 (similarities.py:168-230) and K11: the packed kernel's per-ray tails (arg-max
    plane, distance; planes -> voxels) take each ray's segment from ITS lanes."""
    c, o, P, feats, ridx, vg = _case(oracle_mod, D, N)
    n = len(ridx) - 3
    ridx = ridx[:n]
    res = {}
    for mode in (1, 0):
        ctx = _ctx(o, mode)
        ctx.set_voxel_grid(torch.from_numpy(vg).cuda())
        r_d, f_d, P_d = ctx.dev(ridx), ctx.dev(feats), ctx.dev(P)
        Pi_d, cc_d = ctx.dev(c["P_inv"]), ctx.dev(c["center"])
        S = torch.zeros((n, D), device="cuda")
        pts = torch.zeros((n, D, 4), device="cuda")
        depth = torch.zeros((n,), device="cuda")
        ctx.mvcnn_depth(r_d, f_d, P_d, Pi_d, cc_d, S, pts, depth)
        S9 = torch.zeros((n, D), device="cuda")
        ctx.mvcnn_similarities(r_d, f_d, P_d, Pi_d, cc_d, S9)
        rvi = torch.zeros((n, o.M, 3), dtype=torch.int32, device="cuda")
        rvc = torch.zeros((n,), dtype=torch.int32, device="cuda")
        Sv = torch.zeros((n, o.M), device="cuda")
        ctx.mvcnn_voxel_space(r_d, f_d, P_d, Pi_d, cc_d, rvi, rvc, Sv)
        res[mode] = [t.cpu().numpy() for t in (S, pts, depth, S9, rvi, rvc, Sv)]
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    S, pts, depth, S9, rvi, rvc, Sv = res[0]
    assert np.array_equal(S, S9)
    s, e = o.sample(ridx, c["P_inv"], c["center"])
    So = o.similarities(feats, P, s, e)
    assert np.abs(S - So).max() <= 1e-5
    k = np.argmax(So, axis=1)
    srt = np.sort(So, axis=1)
    sure = srt[:, -1] - srt[:, -2] > 1e-5
    expect = np.sqrt(((pts[np.arange(n), k, :3] - c["center"][:3]) ** 2).sum(1))
    assert np.abs(depth - expect)[sure].max() <= 1e-5
    rvi_o, rvc_o = o.traversal(s, e)
    Sv_o = o.planes_to_voxels(vg, rvi_o, rvc_o, s, e, So)
    assert np.array_equal(rvc, rvc_o) and np.array_equal(rvi, rvi_o)
    assert np.abs(Sv - Sv_o).max() <= 2e-5


@pytest.mark.parametrize("D", [32, 16, 21])
def test_resident_prepare_same_bits(torch, oracle_mod, D):
    """rn_scene_prepare (MAPMODE 2: clipped + renormalised columns) on packed voxel lists: LDS-DMA
    of the first ray's row under the sweep, of the others' when their turn comes."""
    c, o, P, feats, ridx, vg = _case(oracle_mod, D, 5)
    n = len(ridx) - 1
    ridx = ridx[:n]
    res = {}
    for mode in (1, 0):
        ctx = _ctx(o, mode)
        ctx.set_voxel_grid(torch.from_numpy(vg).cuda())
        f_d = ctx.dev(feats)
        vox = torch.zeros((n, o.M), dtype=torch.int32, device="cuda")
        rvc = torch.zeros((n,), dtype=torch.int32, device="cuda")
        Sr = torch.zeros((n, o.M), device="cuda")
        ctx.scene_prepare(ctx.dev(ridx), [f_d[v] for v in range(o.N)], ctx.dev(P),
                          ctx.dev(c["P_inv"]), ctx.dev(c["center"]), vox, rvc, Sr)
        res[mode] = (vox.cpu().numpy(), rvc.cpu().numpy(), Sr.cpu().numpy())
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    _, rvc, Sr = res[0]
    live = rvc > 1
    assert live.sum() > n // 2
    assert np.abs(Sr[live].sum(1) - 1).max() < 1e-5


@pytest.mark.parametrize("D", [32, 16])
def test_forward_pass_same_bits_and_meets_the_oracle(torch, oracle_mod, D):
    """The whole path (folded first BP iteration = MAPMODE 3 in the packed kernel) with fixed-point
    sums: bit-identical accumulator and maps whichever kernel sweeps; and against the oracle."""
    from test_forward_pass_gpu import _depth_close, _gp, _oracle_forward
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import make_synthetic_scene
    H, W, M, grid = 30, 38, 96, (32, 32, 32)          # 1140 rays per image: not a multiple of 4
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(D, M, grid)
    cls = get_forward_pass_factory("raynet")
    res = {}
    for mode in (1, 0):
        fp = cls(bank, gp, "sample_in_bbox", (H, W), 300,
                 options=PathOptions(deterministic=True, sweep_rays_per_wave=mode))
        maps = [np.array(m) for m in fp.forward_pass(scene, (0, 3, 1))]
        res[mode] = (maps, fp.accumulator.cpu().numpy())
        del fp
    assert np.array_equal(res[0][1], res[1][1])
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    oracle_mod.Oracle.set_robust_messages(True)
    try:
        acc_o, _, depth_o, dist_o = _oracle_forward(oracle_mod, scene, bank, gp, [0, 1, 2], H, W)
    finally:
        oracle_mod.Oracle.set_robust_messages(False)
    assert np.abs(res[0][1] - acc_o).max() <= 2e-4
    for i in range(3):
        assert _depth_close(res[0][0][i], depth_o[i], dist_o[i], W, H) <= 0.01
