"""a1 (`sample_in_bbox`) pinned by the reference's own NumPy sampling scheme.

tests/golden/ref_sampling_np.npz holds the output of the reference's
`SamplingInBboxScheme._sample_points_across_rays` (raynet/common/sampling_schemes.py:121-156)
and `sample_points_across_ray` (:100-119) on the mock Restrepo cameras and on the ring
cameras of bench.py's scene (generator: tests/golden/gen_sampling_from_reference.py).  The
CUDA flavour the forward pass runs (sampling_schemes.cu:15-122) differs from it only in
precision (fp32 P_inv and points with an fp64 back-projection, against float64 throughout)
and in the ORDER of the two end points of a ray that misses the box (the `.cu` puts the
intersection with the smaller |t| first, :78-84; NumPy keeps (t_near, t_far)).

CPU tests hold the C oracle to those vectors; the `gpu` tests hold `rn_sample_rays` /
`rn_sample_points` to them through the C ABI.  Tolerance: 1e-5 relative to the scene's
extent (observed: 1.1e-6).  The three properties of the reference's
tests/test_sampling_schemes.py:33-182 (collinear with the camera centre, re-projection to
the source pixel, inside the bounding box) are restated on every ray instead of on one
random sample."""
import numpy as np
import pytest

from conftest import load_cases

REF = load_cases("ref_sampling_np.npz")
REL_TOL = 1e-5


def _scale(c):
    return float(np.abs(np.concatenate([c["bbox"], c["center"][:3]])).max())


def _hits(c):
    """Rays whose reference points all lie inside the box (the others miss it)."""
    bb = c["bbox"]
    p = c["points"][..., :3]
    eps = 1e-4 * _scale(c)
    return np.all((p >= bb[:3] - eps) & (p <= bb[3:] + eps), axis=(1, 2))


def _oracle_for(oracle_mod, c):
    H, W, D = (int(v) for v in c["HWD"])
    return oracle_mod.Oracle(M=8, D=D, N=2, F=4, H=H, W=W, padding=1, bbox=c["bbox"],
                             grid_shape=(4, 4, 4))


def check_ends_against_reference(c, s, e):
    gs, ge = c["points"][:, 0, :3], c["points"][:, -1, :3]
    hit = _hits(c)
    tol = REL_TOL * _scale(c)
    assert hit.sum() >= 0.85 * len(hit)
    assert np.abs(s[hit] - gs[hit]).max() <= tol and np.abs(e[hit] - ge[hit]).max() <= tol
    if (~hit).any():        # missing rays: the same two points, smaller |t| first
        assert np.abs(s[~hit] - ge[~hit]).max() <= tol
        assert np.abs(e[~hit] - gs[~hit]).max() <= tol


def check_points_against_reference(c, pts):
    """All D points of the hitting rays: np.linspace(t_near, t_far, D) in float64 against
    s + k (e - s) / (D - 1) in fp32 (sampling_schemes.cu:104-121)."""
    hit = _hits(c)
    assert np.abs(pts[hit][..., :3] - c["points"][hit][..., :3]).max() <= REL_TOL * _scale(c)
    assert np.all(pts[..., 3] == 1.0)       # (the generator asserts the same of the reference's)


def check_properties(c, pts):
    """tests/test_sampling_schemes.py:33-182 on every ray that hits the box."""
    H, W, D = (int(v) for v in c["HWD"])
    hit = _hits(c)
    p = pts[hit][..., :3].astype(np.float64)
    center = c["center"][:3].astype(np.float64)
    # collinearity (utils/geometry.py:148-164): cross(p_far - p_near, p_near - centre) ~ 0;
    # the reference's atol of 2e-5 is for unit-sized vectors, scaled here by their lengths
    v0 = p[:, -1] - p[:, 0]
    v1 = p[:, 0] - center
    cr = np.linalg.norm(np.cross(v0, v1), axis=1)
    assert np.all(cr <= 2e-5 * np.maximum(1.0, np.linalg.norm(v0, axis=1) * np.linalg.norm(v1, axis=1)))
    # every point of a ray re-projects onto the ray's own pixel (:83-131)
    ridx = c["ray_idxs"][hit]
    px, py = ridx // H, ridx % H
    ph = np.concatenate([p, np.ones(p.shape[:2] + (1,))], axis=2) @ c["P"].T
    uv = np.round(ph[..., :2] / ph[..., 2:])
    assert np.array_equal(uv[..., 0], np.broadcast_to(px[:, None], uv.shape[:2]))
    assert np.array_equal(uv[..., 1], np.broadcast_to(py[:, None], uv.shape[:2]))
    # first and last point inside the bounding box (:133-182, is_between_simple), up to the
    # fp32 rounding of a point that lies ON a face
    eps = 4 * np.finfo(np.float32).eps * _scale(c)
    for q in (p[:, 0], p[:, -1]):
        assert np.all(q >= c["bbox"][:3] - eps) and np.all(q <= c["bbox"][3:] + eps)


# ---------------------------------------------------------------- the fixture itself
@pytest.mark.parametrize("case", sorted(REF))
def test_reference_vectors_are_self_consistent(case):
    """The vectorised and the per-ray entry points of the reference agree, and the
    reference's own properties hold on its own output."""
    c = REF[case]
    check_properties(c, c["points"])
    hit = _hits(c)[:8]
    for j in range(8):
        if c["single_hit"][j]:
            assert np.allclose(c["single_points"][j], c["points"][j], atol=2e-5 * _scale(c))
        else:       # sample_points_across_ray returns None exactly for the missing rays
            assert not hit[j]


# ---------------------------------------------------------------- oracle (CPU)
@pytest.mark.parametrize("case", sorted(REF))
def test_oracle_sample_in_bbox_vs_reference_numpy(oracle_mod, case):
    c = REF[case]
    o = _oracle_for(oracle_mod, c)
    s, e = o.sample(c["ray_idxs"], c["P_pinv"].astype(np.float32), c["center"])
    check_ends_against_reference(c, s, e)
    D = o.D
    k = np.arange(D, dtype=np.float32)[None, :, None]
    pts = np.ones((len(s), D, 4), np.float32)
    pts[..., :3] = s[:, None, :] + k * (e - s)[:, None, :] / np.float32(D - 1)
    check_points_against_reference(c, pts)
    check_properties(c, pts)


# ---------------------------------------------------------------- HIP (through the C ABI)
@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (no CPU fallback exists)"
    from raynet_amd import _lib
    _lib.build()
    return torch


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(REF))
def test_hip_sample_rays_vs_reference_numpy(torch, oracle_mod, case):
    from raynet_amd.hip_implementations import get_context
    c = REF[case]
    o = _oracle_for(oracle_mod, c)
    ctx = get_context(o.M, o.D, o.N, o.F, o.H, o.W, o.padding, o.bbox, o.grid_shape)
    ridx = ctx.dev(c["ray_idxs"])
    n = len(ridx)
    P_inv, center = ctx.dev(c["P_pinv"].astype(np.float32)), ctx.dev(c["center"])
    s = torch.zeros((n, 3), device="cuda")
    e = torch.zeros((n, 3), device="cuda")
    ctx.sample_rays(ridx, P_inv, center, s, e)
    s, e = s.cpu().numpy(), e.cpu().numpy()
    check_ends_against_reference(c, s, e)
    so, eo = o.sample(c["ray_idxs"], c["P_pinv"].astype(np.float32), c["center"])
    assert np.array_equal(s, so) and np.array_equal(e, eo)      # and bit-equal to the oracle
    pts = torch.zeros((n, o.D, 4), device="cuda")
    ctx.sample_points(ridx, P_inv, center, pts)
    pts = pts.cpu().numpy()
    check_points_against_reference(c, pts)
    check_properties(c, pts)
