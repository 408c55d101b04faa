"""The CPU oracle (oracle/raynet_oracle.c) against vectors produced by the
reference's own code (tests/golden/gen_from_reference.py) -- the pin that lets
the GPU parity tests trust the oracle.  CPU only."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO, load_cases

TRAV = load_cases("ref_traversal.npz")
MRF = load_cases("ref_mrf_np.npz")
MAP = load_cases("ref_mapping_np.npz")


def _oracle(oracle_mod, grid, M, bbox=(0, 0, 0, 1, 1, 1), D=8):
    return oracle_mod.Oracle(M=M, D=D, N=2, F=4, H=4, W=4, padding=3, bbox=bbox,
                             grid_shape=grid)


@pytest.mark.parametrize("case", sorted(TRAV))
def test_traversal_bit_exact(oracle_mod, case):
    """Index maps and counts equal the reference's Cython traversal exactly
    (ray_marching/ray_tracing.pyx:64-199; golden list tests/test_ray_marching.py:66-77)."""
    c = TRAV[case]
    o = _oracle(oracle_mod, c["grid"], int(c["M"]), c["bbox"])
    rvi, rvc = o.traversal(c["starts"], c["ends"])
    assert np.array_equal(rvc, c["rvc"])
    assert np.array_equal(rvi, c["rvi"].astype(np.int32))


def test_traversal_reference_unit_test_values(oracle_mod):
    """tests/test_ray_marching.py:20-77 restated on the oracle's drop-in signature."""
    o = _oracle(oracle_mod, (3, 3, 1), 10)
    bbox = np.array([3, 3, 0, 6, 6, 1], np.float32)
    grid = np.array([3, 3, 1], np.int32)
    voxels = np.zeros((10, 3), np.int32)
    n = o.voxel_traversal(bbox, grid, voxels, np.array([3., 4.1, .5], np.float32),
                          np.array([6., 4.9, .5], np.float32))
    assert n == 3 and np.all(voxels[:3, 1] == 1) and np.all(voxels[:3, 0] == np.arange(3))
    for s, e, cnt in [([4., 6., .5], [6., 5., .5], 2), ([3., 3., .5], [6., 6., .5], 5),
                      ([6., 6., .5], [3., 3., .5], 5)]:
        voxels.fill(0)
        assert o.voxel_traversal(bbox, grid, voxels, np.array(s, np.float32),
                                 np.array(e, np.float32)) == cnt
    bbox = np.array([0, 0, 0, 6, 6, 1], np.float32)
    grid = np.array([6, 6, 1], np.int32)
    voxels.fill(0)
    n = o.voxel_traversal(bbox, grid, voxels, np.array([0., 3.5, .5], np.float32),
                          np.array([6., .5, .5], np.float32))
    assert n == 9
    assert np.all(voxels == np.array([[0, 3, 0], [0, 2, 0], [1, 2, 0], [2, 2, 0], [2, 1, 0],
                                      [3, 1, 0], [4, 1, 0], [4, 0, 0], [5, 0, 0], [0, 0, 0]]))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REPO, "oracle", "_ref")),
                    reason="oracle/_ref not built (needs /root/reference)")
def test_traversal_against_live_reference_build(oracle_mod):
    """Fresh random chords through 64^3 against the reference's compiled Cython
    traversal (oracle/_ref), beyond the committed fixtures."""
    sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))
    try:
        import ray_tracing
    except ImportError:
        pytest.skip("oracle/_ref/ray_tracing not importable here")
    rng = np.random.default_rng(99)
    bbox = np.array([-2, -1, -0.5, 2, 1, 0.5], np.float32)
    grid = np.array([64, 48, 40], np.int32)
    M = 160
    o = _oracle(oracle_mod, grid, M, bbox)
    lo, hi = bbox[:3], bbox[3:]
    starts = (lo + rng.random((500, 3)) * (hi - lo)).astype(np.float32)
    ends = (lo + (rng.random((500, 3)) * 1.4 - 0.2) * (hi - lo)).astype(np.float32)
    rvi, rvc = o.traversal(starts, ends)
    for r in range(len(starts)):
        v = np.zeros((M, 3), np.int32)
        n = ray_tracing.voxel_traversal(bbox, grid, v, starts[r], ends[r])
        assert n == rvc[r]
        assert np.array_equal(v, rvi[r])


def _logit_close(a, b, ref_msg):
    """|a-b| <= 1e-5 + 8*eps32*exp(|m|): a log-odds message m = logit(p) computed
    in fp32 carries an error of a few ulp of p divided by p(1-p) ~ exp(-|m|); the
    reference's NumPy path forms the cumulative sums in float64 (mrf_np.py:78-112),
    the CUDA path and the oracle in fp32."""
    tol = 1e-5 + 8 * 2.0 ** -24 * np.exp(np.minimum(np.abs(ref_msg), 17.0))
    return bool(np.all(np.abs(a - b) <= tol))


@pytest.fixture(params=[False, True], ids=["literal", "robust"])
def message_form(request, oracle_mod):
    """Both message forms of the oracle (raynet_oracle.c, g_robust_messages): the literal
    restatement of mrf_bp.cu:136-167 and the robust one the full-size comparisons use."""
    oracle_mod.Oracle.set_robust_messages(request.param)
    yield request.param
    oracle_mod.Oracle.set_robust_messages(False)


@pytest.mark.parametrize("case", sorted(MRF))
def test_bp_matches_reference_numpy(oracle_mod, case, message_form):
    """Accumulator after each of 3 iterations, messages and final depth
    distribution vs mrf/mrf_np.py:243-385 (float64 cumprod inside, float32
    storage).  Tolerances from SURVEY.md Q8."""
    c = MRF[case]
    M = c["S"].shape[1]
    o = _oracle(oracle_mod, c["grid"], M)
    accs = []
    msgs = np.random.default_rng(5).random(c["S"].shape).astype(np.float32)  # ignored
    acc, msgs = o.belief_propagation(c["S"], c["rvi"], c["rvc"], msgs, gamma=0.05,
                                     bp_iterations=3,
                                     callback=lambda it, a, m: accs.append(a.copy()))
    accs = np.stack(accs)
    assert _logit_close(accs, c["accs"], ref_msg=c["accs"] - o.prior(0.05))
    assert _logit_close(msgs, c["msgs"], ref_msg=c["msgs"])
    S_new = o.depth_distribution(c["S"], c["rvi"], c["rvc"], acc, msgs)
    assert np.abs(S_new - c["S_new"]).max() < 1e-5
    # rays with count <= 1 are skipped (mrf_np.py:300, :376)
    skip = c["rvc"] <= 1
    assert np.all(msgs[skip] == 0) and np.all(S_new[skip] == 0)


def _occupancy(acc):
    # mrf_np.py:206-240 compute_occupancy_probabilities
    mx = np.maximum(0.0, acc)
    t1, t2 = np.exp(0.0 - mx), np.exp(acc - mx)
    return t2 / (t2 + t1)


def test_bp_reference_properties(oracle_mod):
    """The assertions of tests/test_mrf.py (:73-76, :140-144, :213-215, :281-304,
    :349, :414-416) evaluated on the oracle."""
    res = {}
    for case in ("single_ray", "two_rays", "two_rays_2", "three_rays", "conflict"):
        c = MRF[case]
        o = _oracle(oracle_mod, c["grid"], c["S"].shape[1])
        msgs = np.zeros_like(c["S"])
        acc, msgs = o.belief_propagation(c["S"], c["rvi"], c["rvc"], msgs)
        res[case] = (o, c, acc, msgs, _occupancy(acc))
    p = res["single_ray"][4]
    ix = np.where(p == p.max())
    assert ix[0][0] == 2 and ix[1][0] == 2
    p = res["two_rays"][4].T
    assert max(p[0, 4, 3], p[0, 2, 2]) >= p.max() - 1e-12
    p = res["two_rays_2"][4].T
    assert p[0, 2, 2] >= p.max() - 1e-12
    p = res["three_rays"][4].T
    order = np.sort(p[0].ravel())[::-1]
    assert p[0, 2, 2] == order[0] and p[0, 2, 0] == order[1] and p[0, 4, 4] == order[2]
    o, c, acc, msgs, p = res["conflict"]
    assert p.T[0, 0, 2] < 0.1
    S_new = o.depth_distribution(c["S"], c["rvi"], c["rvc"], acc, msgs)
    assert S_new[0, 2] < 0.5 and S_new[0, 6] > 0.9 and S_new[1, 4] > 0.9


@pytest.mark.parametrize("case", sorted(MAP))
def test_mapping_matches_reference_numpy(oracle_mod, case):
    """planes_voxels_mapping.cu:6-92 restatement vs the reference's NumPy `li`
    (np.interp) and `li_2` variants (planes_voxels_mapping.py:122-211), the
    comparison tests/test_planes_voxels_mapping.py:61-78 makes between variants."""
    c = MAP[case]
    C, D = len(c["voxels"]), len(c["s"])
    if D < 2:
        pytest.skip("D=1 undefined")
    o = oracle_mod.Oracle(M=C + 3, D=D, N=2, F=4, H=4, W=4, padding=3,
                          bbox=(0, 0, 0, 1, 1, 1), grid_shape=(C, 1, 1))
    grid = c["voxels"].reshape(C, 1, 1, 3)
    rvi = np.zeros((1, C + 3, 3), np.int32)
    rvi[0, :C, 0] = np.arange(C)
    S_new = o.planes_to_voxels(grid, rvi, np.array([C], np.int32), c["start"][None],
                               c["end"][None], c["s"][None])
    assert np.all(S_new[0, C:] == 0)
    assert np.allclose(S_new[0, :C], c["li"], rtol=2e-5, atol=1e-7)
    assert np.allclose(S_new[0, :C], c["li_2"], rtol=2e-5, atol=1e-7)
    assert abs(S_new[0, :C].sum() - 1) < 1e-5


def test_config1_cpu_plumbing(oracle_mod):
    """BASELINE.json configs[0] (Restrepo mock cameras, 2 views, 16 planes, 32^3) through the
    oracle on the CPU: the plumbing case the reference's NumPy path covers."""
    from conftest import GOLDEN
    from raynet_amd.common.scene import restrepo_cameras_scene
    H, W = 36, 64
    scene = restrepo_cameras_scene(os.path.join(GOLDEN, "restrepo_mock_scene_1"), (H, W),
                                   scale=W / 1280.0)
    rng = np.random.default_rng(3)
    feats = (rng.standard_normal((2, H + 12, W + 12, 32)) * 0.25).astype(np.float32)
    views = scene.view_indices_with_neighbors(0, 1)
    assert views == [0, 1]
    P = np.array([scene.get_image(v).camera.P for v in views], np.float32)
    cam = scene.get_image(0).camera
    o = oracle_mod.Oracle(M=96, D=16, N=2, F=32, H=H, W=W, padding=11, bbox=scene.bbox.ravel(),
                          grid_shape=(32, 32, 32))
    vg = oracle_mod.voxel_grid_centers(scene.bbox.ravel(), (32, 32, 32))
    ridx = np.arange(H * W, dtype=np.int32)
    acc = o.prior(0.05)
    msgs = np.zeros((H * W, 96), np.float32)
    for it in range(3):
        out = o.prior(0.05)
        rvi, rvc, Sv = o.fused_bp(ridx, feats, P, cam.P_pinv.astype(np.float32),
                                  cam.center.ravel().astype(np.float32), vg, acc, msgs, out)
        acc = out
    _, _, S_new, depth = o.fused_depth(ridx, feats, P, cam.P_pinv.astype(np.float32),
                                       cam.center.ravel().astype(np.float32), vg, acc, msgs)
    hit = rvc >= 2
    assert hit.mean() > 0.5 and np.isfinite(acc).all() and np.isfinite(depth).all()
    assert np.abs(S_new[hit].sum(1) - 1).max() < 1e-4
    # the cameras fly ~17 units from the site
    assert 10 < np.median(depth[hit]) < 25


def test_neighbour_view_rule_matches_reference():
    """get_adjacent_frames_idxs against the table the reference's own function produced
    (tests/golden/gen_adjacent_frames_from_reference.py), plus the assertions of the
    reference's tests/test_scene.py:120-128."""
    import json
    import warnings
    from raynet_amd.common.scene import adjacent_views, get_adjacent_frames_idxs
    table = json.load(open(os.path.join(GOLDEN, "ref_adjacent_frames.json")))
    assert len(table) > 600
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                  # the rule's own uint32 wrap-arounds
        for ref_idx, n_frames, n_adjacent, skip, expected in table:
            try:
                got = [int(v) for v in get_adjacent_frames_idxs(ref_idx, n_frames, n_adjacent, skip)]
            except Exception as e:                       # noqa: BLE001
                got = "error:" + type(e).__name__
            assert got == expected, (ref_idx, n_frames, n_adjacent, skip, got, expected)
    assert adjacent_views(0, 50, 4) == [1, 2, 3, 4]
    assert adjacent_views(1, 50, 4) == [0, 2, 3, 4]
    assert adjacent_views(35, 50, 4) == [33, 34, 36, 37]
    assert adjacent_views(50, 50, 4) == [46, 47, 48, 49]


def test_private_accumulators_sum_what_the_shared_one_sums(oracle_mod):
    """bench.py's cpu_baseline times the oracle's multi-threaded K1 twice: `omp atomic` adds into
    ONE accumulator, and a private accumulator per thread merged at the end
    (Oracle.set_private_accumulators).  Same lists, columns and messages; the accumulator is the
    same sum in another order."""
    from conftest import GOLDEN
    from raynet_amd.common.scene import restrepo_cameras_scene
    H, W = 36, 64
    scene = restrepo_cameras_scene(os.path.join(GOLDEN, "restrepo_mock_scene_1"), (H, W),
                                   scale=W / 1280.0)
    rng = np.random.default_rng(5)
    feats = (rng.standard_normal((2, H + 12, W + 12, 32)) * 0.25).astype(np.float32)
    P = np.array([scene.get_image(v).camera.P for v in (0, 1)], np.float32)
    cam = scene.get_image(0).camera
    Pi, cc = cam.P_pinv.astype(np.float32), cam.center.ravel().astype(np.float32)
    vg = oracle_mod.voxel_grid_centers(scene.bbox.ravel(), (32, 32, 32))
    ridx = np.arange(H * W, dtype=np.int32)
    res = {}
    for tag, threads, private in (("one", 1, False), ("shared", 4, False), ("private", 4, True)):
        o = oracle_mod.Oracle(M=96, D=16, N=2, F=32, H=H, W=W, padding=11, bbox=scene.bbox.ravel(),
                              grid_shape=(32, 32, 32), threads=threads)
        oracle_mod.Oracle.set_private_accumulators(private)
        try:
            out = o.prior(0.05)
            msgs = np.zeros((H * W, 96), np.float32)
            rvi, rvc, Sv = o.fused_bp(ridx, feats, P, Pi, cc, vg, o.prior(0.05), msgs, out)
        finally:
            oracle_mod.Oracle.set_private_accumulators(False)
        res[tag] = (rvi, rvc, Sv, msgs, out)
    for tag in ("shared", "private"):
        for a, b in zip(res[tag][:4], res["one"][:4]):
            assert np.array_equal(a, b)
        assert np.abs(res[tag][4] - res["one"][4]).max() <= 1e-4
    assert np.abs(res["one"][4] - res["one"][4][0, 0, 0]).max() > 1.0      # (messages did arrive)
