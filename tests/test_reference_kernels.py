"""The oracle and the HIP path against a RUN OF THE REFERENCE'S OWN KERNELS on an MI355X.

tests/golden/ref_cu_gfx950.npz holds inputs and outputs of the reference's .cu kernels -- the text
of cuda_implementations/*.cu with its $placeholders substituted as raynet_fp.py:230-248 does and
NOTHING else changed, compiled for gfx950 by oracle/build_ref_cu.py and launched one thread per
ray by tests/golden/gen_refcu_from_reference.py.  This is the pin of the plane sweep's VALUES (a2,
feature_similarities.cu:66-146), which no CPU implementation of the reference exists for, and a
second, GPU-side pin of a1 / a3 / a4 / a5 / a6.

Convention.  The fixture's outputs come from the -ffp-contract=off build:
single IEEE operations in the order the source writes them -- what the oracle, the reference's
NumPy `project` and its Cython traversal compute, and the one convention every compiler
reproduces.  A contracted build (nvcc's and hipcc's default) fuses SOME of the projection's
multiply-adds -- which ones is the compiler's choice (hipcc's vectoriser leaves the homogeneous
row unfused) -- and moves a ray's column only where one of its N x D projections lands within
~1e-4 px of a rounding boundary: 0.1 % of config 2's rays (profiles/r05_ref_cu_report_config2.json).
The fixture lists those rays (`fma_rows`); the tests hold the HIP sweeps to the contracted build
everywhere else and bound their number.

CPU tests: the oracle against the fixture.  `-m gpu`: the HIP kernels against the fixture and --
when oracle/_ref/*.co travelled with the snapshot -- against a live launch of the reference's
kernel over every ray of a config-2 image.
"""
import os

import numpy as np
import pytest

from conftest import load_cases

REF = load_cases("ref_cu_gfx950.npz")
CASES = sorted(REF)


def _features(c):
    rng = np.random.default_rng(int(c["seed"]))
    shape = (int(c["N"]), int(c["H"]) + int(c["padding"]) + 1, int(c["W"]) + int(c["padding"]) + 1,
             int(c["F"]))
    return rng.standard_normal(shape, dtype=np.float32) * np.float32(0.25)


def _oracle(oracle_mod, c):
    return oracle_mod.Oracle(M=int(c["M"]), D=int(c["D"]), N=int(c["N"]), F=int(c["F"]), H=int(c["H"]),
                             W=int(c["W"]), padding=int(c["padding"]), bbox=c["bbox"],
                             grid_shape=c["grid"], threads=oracle_mod.Oracle.max_threads())


def _dense(c, which):
    prior = np.float32(np.log(c["gamma"]) - np.log(1 - c["gamma"]))
    a = np.full(int(np.prod(c["grid"])), prior, np.float32)
    a[c[which + "_at"]] = c[which + "_val"]
    return a.reshape(tuple(int(g) for g in c["grid"]))


def logit_tol(ref_msg):
    return 1e-5 + 8 * 2.0 ** -24 * np.exp(np.minimum(np.abs(ref_msg), 17.0))


# =========================================================================== the oracle (CPU)
@pytest.mark.parametrize("case", CASES)
def test_oracle_sampling_vs_reference_kernel(oracle_mod, case):
    """a1: batch_sample_points_in_bbox (sampling_schemes.cu:92-122).  Point 0 of a ray IS its
    start (start + 0); the other points are start + k (end - start) / (D - 1) in float."""
    c = REF[case]
    o = _oracle(oracle_mod, c)
    s, e = o.sample(c["ray_idxs"], c["P_inv"], c["center"])
    assert np.array_equal(s, c["starts"]) and np.array_equal(e, c["ends"])
    assert np.array_equal(s, c["points_first"])
    D = o.D
    k = np.arange(D, dtype=np.float32)[None, :, None]
    pts = (s[:64, None, :] + k * (e - s)[:64, None, :] / np.float32(D - 1)).astype(np.float32)
    assert np.array_equal(pts, c["points"][..., :3]) and np.all(c["points"][..., 3] == 1)
    last = (s + np.float32(D - 1) * (e - s) / np.float32(D - 1)).astype(np.float32)
    assert np.array_equal(last, c["points_last"])


@pytest.mark.parametrize("case", CASES)
def test_oracle_similarities_vs_reference_kernel(oracle_mod, case):
    """a2 VALUES: rno_similarities against batch_compute_similarities.  Same feature indices,
    same order of the F-term sums; what differs is the device's expf (<= 2 ulp)."""
    c = REF[case]
    o = _oracle(oracle_mod, c)
    S = o.similarities(_features(c), c["P"], c["starts"], c["ends"])
    assert np.abs(S - c["S_nofma"]).max() <= 1e-7
    # a1 + a2 in one kernel (similarities.py:44-81): a2's bits on a1's end points.  ([1], the
    # contracted build, also fuses a1's `center + t * dir` and starts from other end points.)
    assert bool(c["mvcnn_equals_a2"][0])


@pytest.mark.parametrize("case", CASES)
def test_contracted_build_moves_few_rays(case):
    """What -ffp-contract=fast does to the reference's own kernel: a handful of rays (index
    flips at rounding boundaries), nothing above rounding noise elsewhere."""
    c = REF[case]
    n = len(c["ray_idxs"])
    moved = np.abs(c["S_fma_rows"] - c["S_nofma"][c["fma_rows"]]).max(1) > 1e-5 if len(c["fma_rows"]) \
        else np.zeros(0, bool)
    assert moved.sum() <= max(1, 0.005 * n)
    assert float(c["fma_other_max"]) <= 1e-6


@pytest.mark.parametrize("case", CASES)
def test_oracle_traversal_vs_reference_cuda_kernel(oracle_mod, case):
    """a3: the CUDA flavour (ray_tracing.cu, bbox literals promoted to double) gives the Cython
    flavour's lists (the bit-exactness target) on these boxes -- whose corners are
    exactly representable, so the promotion changes no rounding."""
    c = REF[case]
    o = _oracle(oracle_mod, c)
    k = int(c["m_rays"])
    rvi, rvc = o.traversal(c["starts"][:k], c["ends"][:k])
    assert np.array_equal(rvc, c["rvc"])
    assert np.array_equal(rvi, c["rvi"].astype(np.int32))


@pytest.mark.parametrize("case", CASES)
def test_oracle_mapping_vs_reference_kernel(oracle_mod, case):
    """a4: batch_planes_voxels_mapping (planes_voxels_mapping.cu:94-119)."""
    c = REF[case]
    o = _oracle(oracle_mod, c)
    k = int(c["m_rays"])
    vg = oracle_mod.voxel_grid_centers(c["bbox"], c["grid"])
    Sv = o.planes_to_voxels(vg, c["rvi"].astype(np.int32), c["rvc"], c["starts"][:k], c["ends"][:k],
                            c["S_nofma"][:k])
    assert np.abs(Sv - c["S_voxel"]).max() <= 2e-7


@pytest.mark.parametrize("case", CASES)
def test_oracle_bp_and_depth_vs_reference_kernel(oracle_mod, case):
    """a5 / a6: two sweeps of batch_belief_propagation in the driver's schedule and
    batch_depth_estimation (mrf_bp.cu:180-229) against the oracle's LITERAL message form (the
    kernel's own sequence)."""
    c = REF[case]
    o = _oracle(oracle_mod, c)
    ok = c["bp_valid"]
    rvi, rvc, Sv = c["rvi"].astype(np.int32)[ok], c["rvc"][ok], c["S_voxel"][ok]
    acc0 = o.prior(float(c["gamma"]))
    msgs = np.zeros_like(Sv)
    acc1 = o.prior(float(c["gamma"]))
    oracle_mod.Oracle.set_robust_messages(False)
    o.bp_sweep(Sv, rvi, rvc, acc0, msgs, acc1)
    assert np.all(np.abs(msgs - c["msgs1"]) <= logit_tol(c["msgs1"]))
    ref1 = _dense(c, "acc1")
    assert np.abs(acc1 - ref1).max() <= 1e-4 * max(1.0, np.abs(ref1).max())
    acc2 = o.prior(float(c["gamma"]))
    m2 = c["msgs1"].copy()
    o.bp_sweep(Sv, rvi, rvc, ref1, m2, acc2)         # from the kernel's own first-sweep state
    assert np.all(np.abs(m2 - c["msgs2"]) <= 4 * logit_tol(c["msgs2"]))
    ref2 = _dense(c, "acc2")
    assert np.abs(acc2 - ref2).max() <= 2e-4 * max(1.0, np.abs(ref2).max())
    S_new = o.depth_distribution(Sv, rvi, rvc, ref2, c["msgs2"])
    assert np.abs(S_new - c["S_new"]).max() <= 1e-5


# =========================================================================== the HIP path (GPU)
@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (no CPU fallback exists)"
    from raynet_amd import _lib
    _lib.build()
    return torch


def _ctx(c):
    from raynet_amd.hip_implementations import get_context
    return get_context(int(c["M"]), int(c["D"]), int(c["N"]), int(c["F"]), int(c["H"]), int(c["W"]),
                       int(c["padding"]), c["bbox"], c["grid"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("generic", [False, True])
def test_hip_similarities_vs_reference_kernel(torch, case, generic, monkeypatch):
    """Both plane sweeps against batch_compute_similarities' output: <= 2e-6 for the sweep that
    keeps the reference's summation order, <= 1e-5 for the cooperative one (4-term partial sums
    folded across lanes); and against the CONTRACTED build on every ray it does not move."""
    c = REF[case]
    if generic:
        monkeypatch.setenv("RAYNET_HIP_GENERIC_SWEEP", "1")
    from raynet_amd.hip_implementations.options import PathOptions
    ctx = _ctx(c)
    ctx.set_options(PathOptions.from_env())
    feats = ctx.dev(_features(c))
    n = len(c["ray_idxs"])
    S = torch.zeros((n, int(c["D"])), device="cuda")
    ctx.compute_similarities(feats, ctx.dev(c["P"]), ctx.dev(c["starts"]), ctx.dev(c["ends"]), S)
    S = S.cpu().numpy()
    tol = 2e-6 if (generic or int(c["F"]) != 32) else 1e-5
    assert np.abs(S - c["S_nofma"]).max() <= tol
    # the fused driver entry (K9: a1 + a2) as well
    S2 = torch.zeros((n, int(c["D"])), device="cuda")
    ctx.mvcnn_similarities(ctx.dev(c["ray_idxs"]), feats, ctx.dev(c["P"]), ctx.dev(c["P_inv"]),
                           ctx.dev(c["center"]), S2)
    assert np.abs(S2.cpu().numpy() - c["S_nofma"]).max() <= tol
    S_fma = c["S_nofma"].copy()
    S_fma[c["fma_rows"]] = c["S_fma_rows"]
    err = np.abs(S - S_fma).max(1)
    moved = np.zeros(n, bool)
    moved[c["fma_rows"]] = np.abs(c["S_fma_rows"] - c["S_nofma"][c["fma_rows"]]).max(1) > 1e-5
    assert err[~moved].max() <= 2e-5
    monkeypatch.delenv("RAYNET_HIP_GENERIC_SWEEP", raising=False)
    ctx.set_options(PathOptions.from_env())


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_chain_vs_reference_kernels(torch, case):
    """a1, a3, a4, a5, a6 of the HIP library against the reference kernels' outputs: end points
    and index maps bit-exact, mapped columns <= 2e-7, messages within the logit conditioning
    bound of the kernel's literal form (the library sums the suffix directly: the two forms
    agree where messages are moderate, as here), distributions <= 1e-5."""
    from raynet_amd.mrf.mrf_hip import batch_ray_belief_propagation
    from raynet_amd.planes_voxels_mapping.planes_voxels_mapping_hip import batch_depth_to_voxels_mapping
    from raynet_amd.ray_marching.ray_tracing_hip import batch_voxel_traversal
    from oracle import oracle
    c = REF[case]
    ctx = _ctx(c)
    M, D, k = int(c["M"]), int(c["D"]), int(c["m_rays"])
    n = len(c["ray_idxs"])
    s = torch.zeros((n, 3), device="cuda")
    e = torch.zeros((n, 3), device="cuda")
    ctx.sample_rays(ctx.dev(c["ray_idxs"]), ctx.dev(c["P_inv"]), ctx.dev(c["center"]), s, e)
    assert np.array_equal(s.cpu().numpy(), c["points_first"])
    assert np.array_equal(e.cpu().numpy(), c["ends"])
    vtr = batch_voxel_traversal(M, c["bbox"], c["grid"])
    rvi = torch.zeros((k, M, 3), dtype=torch.int32, device="cuda")
    rvc = torch.full((k,), -7, dtype=torch.int32, device="cuda")
    vtr(c["starts"][:k], c["ends"][:k], rvi, rvc)
    assert np.array_equal(rvc.cpu().numpy(), c["rvc"])
    assert np.array_equal(rvi.cpu().numpy(), c["rvi"].astype(np.int32))
    vg = oracle.voxel_grid_centers(c["bbox"], c["grid"])
    pvm = batch_depth_to_voxels_mapping(M, D, tuple(int(g) for g in c["grid"]), c["bbox"])
    Sv = torch.zeros((k, M), device="cuda")
    pvm(vg, rvi, rvc, c["starts"][:k], c["ends"][:k], c["S_nofma"][:k], Sv)
    assert np.abs(Sv.cpu().numpy() - c["S_voxel"]).max() <= 2e-7
    ok = c["bp_valid"]
    grid = tuple(int(g) for g in c["grid"])
    bp, de = batch_ray_belief_propagation(M, grid)
    prior = float(np.float32(np.log(c["gamma"]) - np.log(1 - c["gamma"])))
    rvi_v, rvc_v, Sv_v = c["rvi"].astype(np.int32)[ok], c["rvc"][ok], c["S_voxel"][ok]
    msgs = torch.zeros((int(ok.sum()), M), device="cuda")
    acc1 = torch.full(grid, prior, device="cuda")
    bp(Sv_v, rvi_v, rvc_v, torch.full(grid, prior, device="cuda"), msgs, acc1)
    err1 = np.abs(msgs.cpu().numpy() - c["msgs1"])
    assert np.all(err1 <= logit_tol(c["msgs1"])), "worst %g at |m| = %g" % (
        err1.max(), np.abs(c["msgs1"]).ravel()[err1.argmax()])
    ref1 = _dense(c, "acc1")
    assert np.abs(acc1.cpu().numpy() - ref1).max() <= 1e-4 * max(1.0, np.abs(ref1).max())
    m2 = torch.from_numpy(c["msgs1"].copy()).cuda()
    acc2 = torch.full(grid, prior, device="cuda")
    bp(Sv_v, rvi_v, rvc_v, ref1, m2, acc2)
    # second sweep, messages no longer zero: in front of a saturated voxel the kernel's literal
    # fp32 `cumsum1 - cumsum2` (mrf_bp.cu:157) cancels, the library scans the suffix sum directly.
    # Both are held to the float64 value of the same sweep, the kernel with the cancellation term
    # its own form carries (the bound of test_bp_single_sweep_against_float64_truth).
    from bp_truth import bp_truth_f64
    truth, cancel = bp_truth_f64(Sv_v, rvi_v, rvc_v, ref1, c["msgs1"])
    eps = 2.0 ** -24
    cond = 2 + np.exp(np.minimum(np.abs(truth), 30))
    valid = np.arange(M)[None, :] < rvc_v[:, None]
    e_hip = np.where(valid, np.abs(m2.cpu().numpy() - truth), 0)
    e_ref = np.where(valid, np.abs(c["msgs2"] - truth), 0)
    assert np.all(e_hip <= 5e-5 + 32 * eps * cond), "hip worst %g" % e_hip.max()
    assert np.all(e_ref <= 1e-5 + 32 * eps * cond + 8 * eps * cancel), "reference kernel worst %g" % e_ref.max()
    easy = valid & (cancel < 50)
    assert easy.sum() > 0.9 * valid.sum() and np.abs(m2.cpu().numpy() - c["msgs2"])[easy].max() <= 1e-4
    # the accumulators: prior + the scatter-add of each side's own messages; against the float64
    # sum of the float64 messages, each within the sum of its messages' bounds
    ref2 = _dense(c, "acc2")
    acc_t = np.full(grid, float(prior), np.float64)
    slack_h = np.zeros(grid, np.float64)
    slack_r = np.zeros(grid, np.float64)
    for r in range(len(rvc_v)):
        at = tuple(rvi_v[r, :rvc_v[r]].T)
        np.add.at(acc_t, at, truth[r, :rvc_v[r]])
        np.add.at(slack_h, at, (5e-5 + 32 * eps * cond)[r, :rvc_v[r]])
        np.add.at(slack_r, at, (1e-5 + 32 * eps * cond + 8 * eps * cancel)[r, :rvc_v[r]])
    assert np.all(np.abs(acc2.cpu().numpy() - acc_t) <= 1e-5 + slack_h)
    assert np.all(np.abs(ref2 - acc_t) <= 1e-5 + slack_r)
    S_new = torch.zeros((int(ok.sum()), M), device="cuda")
    de(Sv_v, rvi_v, rvc_v, ref2, c["msgs2"], S_new)
    assert np.abs(S_new.cpu().numpy() - c["S_new"]).max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("config,images", [("config2", (0, 3)), ("config4", (4,))])
def test_hip_sweep_vs_live_reference_kernel_full_image(torch, config, images):
    """Every ray of whole reference images of the bench scenes -- config 2 (5 views, 64 planes) and
    config 4 (9 views, 128 planes: the 9-view cooperative sweep, two plane chunks) -- the library's
    sweep against a LIVE launch of the reference's batch_compute_similarities
    (oracle/_ref/raynet_ref_<config>_nofma.co).  Skipped only when the code objects did not travel
    with the snapshot."""
    import ref_cu
    if not ref_cu.available():
        pytest.skip("oracle/_ref/raynet_ref_*.co not built (oracle/build_ref_cu.py needs /root/reference)")
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.synthetic import make_synthetic_scene
    shape = ref_cu.manifest()["shapes"][config]
    M, D, N, F, H, W, pad = (shape[k] for k in ("M", "D", "N", "F", "H", "W", "padding"))
    bbox = np.asarray(shape["bbox"], np.float32)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=N, F=F, padding=pad, focal=1.5 * H, seed=1234)
    ctx = get_context(M, D, N, F, H, W, pad, bbox, shape["grid"])
    for image in images:
        views = scene.view_indices_with_neighbors(image, N - 1)
        feats = bank.stacked(views)
        P = ctx.dev(np.array([scene.get_image(v).camera.P for v in views], np.float32))
        cam = scene.get_image(image).camera
        n = H * W
        s = torch.zeros((n, 3), device="cuda")
        e = torch.zeros((n, 3), device="cuda")
        ctx.sample_rays(torch.arange(n, dtype=torch.int32, device="cuda"),
                        ctx.dev(cam.P_pinv.astype(np.float32)),
                        ctx.dev(cam.center.ravel().astype(np.float32)), s, e)
        r = ref_cu.RefCu(config, "nofma")
        S_ref = r.similarities(feats, P.reshape(-1), s, e)
        S = torch.zeros((n, D), device="cuda")
        ctx.compute_similarities(feats, P, s, e, S)
        err = (S - S_ref).abs().max(1).values
        assert float(err.max()) <= 1e-5, "image %d: %d rays above 1e-5, worst %g" % (
            image, int((err > 1e-5).sum()), float(err.max()))
        # and what contraction would have moved: a fraction of a percent of the rays (a ray has
        # N x D projections that can land on a rounding boundary: 0.11 % of config 2's rays, 0.52 %
        # of config 4's)
        S_fma = ref_cu.RefCu(config, "fma").similarities(feats, P.reshape(-1), s, e)
        moved = (S_fma - S_ref).abs().max(1).values > 1e-5
        assert float(moved.float().mean()) <= 1.6e-5 * N * D
        assert float((S - S_fma).abs().max(1).values[~moved].max()) <= 2e-5


@pytest.mark.gpu
def test_box_given_as_decimal_text_moves_only_last_bits(torch):
    """The reference bakes the bounding box into its kernels as DECIMAL TEXT (raynet_fp.py:241-246:
    str(np.float32(-0.7)) = "-0.7", read by the compiler as the double -0.7 -- with NumPy < 1.14 it
    would have been "-0.69999999"), the library holds the caller's float32 (-0.699999988).  On the
    mock Restrepo box of BASELINE.json configs[0], whose -0.7 is no float32 value, over every ray of
    the twelve cameras: ray end points differ in their last bits on a few rays, the two traversal
    flavours (CUDA: double-promoted literals; Cython, the one followed: SURVEY Q9) give the same
    lists on the same end points, and the plane-sweep columns of the fused a1 + a2 kernel agree to
    rounding.  (tools/ref_cu_bbox_census.py, profiles/r05_ref_cu_bbox_census.json: 14 starts and
    392 last points of 27,648 rays, <= 2.2e-6.)"""
    import ref_cu
    if not ref_cu.available() or "config1" not in ref_cu.manifest()["shapes"]:
        pytest.skip("oracle/_ref/raynet_ref_config1_*.co not built (oracle/build_ref_cu.py needs /root/reference)")
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "ref_cu_bbox_census.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    c = json.load(open(os.path.join(REPO, "gpurun_out", "r05_ref_cu_bbox_census.json")))
    assert c["rays"] == 12 * 36 * 64 and c["live_rays"] > 0.9 * c["rays"]
    assert c["start_differs"] <= 0.005 * c["rays"] and c["end_differs"] <= 0.05 * c["rays"]
    assert c["max_endpoint_diff"] <= 4e-6          # a few ulp of coordinates up to 5
    assert c["lists_differ_same_endpoints"] == 0
    assert c["sweep_rays_gt_1e5"] == 0 and c["sweep_max"] <= 1e-6


@pytest.mark.gpu
def test_reference_kernels_through_the_reference_schedule_at_full_size(torch):
    """BASELINE config 2 in full through the reference's OWN kernels (3 coupled BP iterations over
    5 x 307,200 rays; tools/ref_cu_fullsize.py): the kernel's literal fp32 `cumsum1 - cumsum2`
    (mrf_bp.cu:157) goes non-finite in a few dozen voxels, the oracle's literal form goes non-finite
    in the SAME voxels and agrees elsewhere to the atomics' order -- the oracle is the kernel, at
    full size -- while the library (the NumPy flavour's semantics) stays
    finite everywhere and agrees with the kernel's depth maps on all but a fraction of a percent
    of the pixels."""
    import ref_cu
    if not ref_cu.available():
        pytest.skip("oracle/_ref/raynet_ref_*.co not built (oracle/build_ref_cu.py needs /root/reference)")
    import json
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "ref_cu_fullsize.py")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.load(open(os.path.join(REPO, "gpurun_out", "r05_ref_cu_fullsize_config2.json")))
    it = rep["iterations"]
    assert it[0]["non_finite_messages"] == 0 and it[2]["non_finite_messages"] > 0
    ok = rep["oracle_literal_form_vs_the_kernel"]
    nk, no, both = ok["non_finite_voxels_kernel"], ok["non_finite_voxels_oracle"], ok["non_finite_in_both"]
    assert nk > 0 and both >= 0.9 * max(nk, no)                  # observed: 52, 52, 52
    assert ok["max_abs_accumulator_diff_where_both_finite"] <= 5e-2 and ok["voxels_beyond_1e-2"] <= 20
    lib = rep["library"]
    assert lib["accumulator_finite_everywhere"] and lib["depth_maps_finite"]
    assert rep["depth_maps"]["fraction_beyond_1e-4"] <= 0.01         # observed 0.33 %


@pytest.mark.gpu
def test_hip_chain_vs_live_reference_kernels_full_image(torch):
    """a3, a4, a5 (first sweep) and a6 over EVERY ray of a config-2 reference image, each library
    kernel against a live launch of the reference's own on the same inputs: voxel lists and counts
    bit-exact, mapped columns to 6e-6 relative (the normaliser's summation order), first-sweep messages within the logit conditioning bound,
    distributions <= 1e-5."""
    import ref_cu
    if not ref_cu.available():
        pytest.skip("oracle/_ref/raynet_ref_*.co not built (oracle/build_ref_cu.py needs /root/reference)")
    from oracle import oracle
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.mrf.mrf_hip import batch_ray_belief_propagation
    from raynet_amd.planes_voxels_mapping.planes_voxels_mapping_hip import batch_depth_to_voxels_mapping
    from raynet_amd.ray_marching.ray_tracing_hip import batch_voxel_traversal
    from raynet_amd.synthetic import make_synthetic_scene
    shape = ref_cu.manifest()["shapes"]["config2"]
    M, D, N, F, H, W, pad = (shape[k] for k in ("M", "D", "N", "F", "H", "W", "padding"))
    bbox, grid = np.asarray(shape["bbox"], np.float32), tuple(shape["grid"])
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=N, F=F, padding=pad, focal=1.5 * H, seed=1234)
    ctx = get_context(M, D, N, F, H, W, pad, bbox, grid)
    r = ref_cu.RefCu("config2", "nofma")
    image, n = 2, H * W
    views = scene.view_indices_with_neighbors(image, N - 1)
    feats = bank.stacked(views)
    P = ctx.dev(np.array([scene.get_image(v).camera.P for v in views], np.float32))
    cam = scene.get_image(image).camera
    s = torch.zeros((n, 3), device="cuda")
    e = torch.zeros((n, 3), device="cuda")
    ctx.sample_rays(torch.arange(n, dtype=torch.int32, device="cuda"), ctx.dev(cam.P_pinv.astype(np.float32)),
                    ctx.dev(cam.center.ravel().astype(np.float32)), s, e)
    # a3
    rvi_r, rvc_r = r.traversal(s, e)
    rvi = torch.zeros((n, M, 3), dtype=torch.int32, device="cuda")
    rvc = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    batch_voxel_traversal(M, bbox, np.array(grid, np.int32))(s, e, rvi, rvc)
    assert torch.equal(rvc, rvc_r) and torch.equal(rvi, rvi_r)
    assert int(rvc.max()) > 200 and int((rvc == 0).sum()) > 0          # long rays, and rays that miss
    # a4 (on the reference's own column)
    S = r.similarities(feats, P.reshape(-1), s, e)
    vg = oracle.voxel_grid_centers(bbox, grid)
    Sv_r = r.planes_to_voxels(vg, rvi, rvc, s, e, S)
    Sv = torch.zeros((n, M), device="cuda")
    batch_depth_to_voxels_mapping(M, D, grid, bbox)(vg, rvi, rvc, s, e, S, Sv)
    # (the column's normaliser is a sum of up to M terms -- sequential in the kernel, where its
    # rounding error grows with the count, a wave reduction here: every value of a ray carries the
    # same relative shift, 2.2e-6 observed on a 254-voxel ray)
    dmap = (Sv - Sv_r).abs()
    assert bool((dmap <= 2e-7 + 6e-6 * Sv_r.abs()).all()), float(dmap.max())
    # a5: the first sweep (accumulator = the prior, messages zero); rays with < 2 voxels send nothing
    prior = float(np.float32(np.log(0.05) - np.log(0.95)))
    rvc_bp = torch.where(rvc >= 2, rvc, torch.zeros_like(rvc))
    acc0 = torch.full(grid, prior, device="cuda")
    m_r = torch.zeros((n, M), device="cuda")
    acc_r = torch.full(grid, prior, device="cuda")
    r.bp_sweep(Sv_r.clone(), rvi, rvc_bp, acc0, m_r, acc_r)
    bp, de = batch_ray_belief_propagation(M, grid)
    m_h = torch.zeros((n, M), device="cuda")
    acc_h = torch.full(grid, prior, device="cuda")
    bp(Sv_r, rvi, rvc, acc0, m_h, acc_h)
    assert bool(torch.isfinite(m_r).all())
    tol = 1e-5 + 8 * 2.0 ** -24 * torch.exp(m_r.abs().clamp(max=17.0))
    assert bool(((m_h - m_r).abs() <= tol).all()), float(((m_h - m_r).abs() / tol).max())
    scale = float(acc_r.abs().max())
    assert float((acc_h - acc_r).abs().max()) <= 2e-5 * scale          # atomics in another order
    # a6 on the reference's accumulator and messages
    S_new_r = r.depth_estimation(Sv_r.clone(), rvi, rvc_bp, acc_r, m_r)
    S_new = torch.zeros((n, M), device="cuda")
    de(Sv_r, rvi, rvc, acc_r, m_r, S_new)
    assert float((S_new - S_new_r).abs().max()) <= 1e-5


@pytest.mark.gpu
def test_hip_k10_vs_live_reference_kernel_full_image(torch):
    """K10, the kernel of the reference's MultiViewCNNForwardPass driver
    (similarities.py:168-227: a1 + a2 + plane points + first arg-max plane + distance to the
    camera), over every ray of a config-2 image: columns <= 1e-5, plane points bit-exact, depths
    equal except where the reference's two best planes are within the columns' tolerance."""
    import ref_cu
    if not ref_cu.available():
        pytest.skip("oracle/_ref/raynet_ref_*.co not built (oracle/build_ref_cu.py needs /root/reference)")
    from raynet_amd.hip_implementations.similarities import \
        perform_multi_view_cnn_forward_pass_with_depth_estimation
    from raynet_amd.synthetic import make_synthetic_scene
    shape = ref_cu.manifest()["shapes"]["config2"]
    M, D, N, F, H, W, pad = (shape[k] for k in ("M", "D", "N", "F", "H", "W", "padding"))
    bbox = np.asarray(shape["bbox"], np.float32)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=N, F=F, padding=pad, focal=1.5 * H, seed=1234)
    image, n = 1, H * W
    views = scene.view_indices_with_neighbors(image, N - 1)
    feats = bank.stacked(views)
    cam = scene.get_image(image).camera
    r = ref_cu.RefCu("config2", "nofma")
    ridx = torch.arange(n, dtype=torch.int32, device="cuda")
    P = r.dev(np.array([scene.get_image(v).camera.P for v in views], np.float32).reshape(-1))
    P_inv, cc = r.dev(cam.P_pinv.astype(np.float32).reshape(-1)), r.dev(cam.center.ravel().astype(np.float32))
    S_r = torch.zeros((n, D), device="cuda")
    pts_r = torch.zeros((n, D, 4), device="cuda")
    depth_r = torch.zeros((n,), device="cuda")
    r.launch("batch_multi_view_cnn_forward_pass_with_depth", n, ridx, feats, P, P_inv, cc, S_r, pts_r, depth_r)
    fp = perform_multi_view_cnn_forward_pass_with_depth_estimation(D, N, F, H, W, pad, bbox, "sample_in_bbox")
    S = torch.zeros((n, D), device="cuda")
    pts = torch.zeros((n, D, 4), device="cuda")
    depth = torch.zeros((n,), device="cuda")
    fp(ridx, feats, P.reshape(N, 3, 4), P_inv.reshape(4, 3), cc, S, pts, depth)
    assert float((S - S_r).abs().max()) <= 1e-5
    assert torch.equal(pts, pts_r)
    top = torch.topk(S_r, 2, dim=1).values
    tie = (top[:, 0] - top[:, 1]) <= 2e-5
    dd = (depth - depth_r).abs()
    # (1.4 % of this image's rays see noise only: flat columns whose two best planes are that close)
    assert float(dd[~tie].max()) <= 1e-5 and float(tie.float().mean()) < 0.05
    assert float((dd <= 1e-5).float().mean()) >= 0.995
