"""SURVEY.md 8(a) row a2, the GEOMETRY of the plane sweep pinned by code the reference can run:
`project` of raynet/utils/geometry.py:9-34 (tests/golden/gen_projection_from_reference.py ->
ref_projection_np.npz).  For every ray, view and depth plane of the fixture the reference's pixel
coordinates decide the feature vector the sweep has to gather,

    f = round(pixel) + padding - (padding - 1) / 2,  clamped to [0, W] x [0, H],
    (0, 0) when either coordinate clamps to 0                 (feature_similarities.cu:42-61)

and the oracle's `rno_feature_index` (CPU) and the HIP sweep's own index arithmetic -- the
reference-order expressions of the generic sweep and the reciprocal shortcut of the cooperative
one (GPU) -- must give exactly that index wherever the reference's pixel is farther than 1e-3 px
from a rounding boundary (fp32 projections of coordinates of a few hundred pixels carry ~1e-4 px).
What stays pinned by reading only: the clamp-to-(0, 0) rule, the F-term dot product, the softmax."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

PADDING = 11
Z = np.load(os.path.join(GOLDEN, "ref_projection_np.npz"))
CASES = sorted({k.rsplit("/", 1)[0] for k in Z.files if "/ref" in k})


def _round_half_away(x):
    return np.trunc(x + np.copysign(0.5, x))


def _expected_indices(pix, H, W):
    """[..., 2] reference pixels (x = column, y = row) -> (fx, fy) and a mask of the samples
    whose two coordinates are both clear of a rounding boundary."""
    shift = PADDING - (PADDING - 1) // 2
    f = _round_half_away(pix) + shift
    fx = np.clip(f[..., 0], 0, W).astype(np.int64)
    fy = np.clip(f[..., 1], 0, H).astype(np.int64)
    zero = (fx == 0) | (fy == 0)
    fx[zero] = 0
    fy[zero] = 0
    frac = np.abs(pix - np.floor(pix) - 0.5)
    return fx, fy, np.all(frac > 1e-3, -1) & np.isfinite(pix).all(-1)


def _views(case):
    group = case.split("/")[0]
    return Z[group + "/P"]


@pytest.mark.parametrize("case", CASES)
def test_oracle_feature_indices_follow_the_reference_projection(oracle_mod, case):
    H, W, D = (int(v) for v in Z[case + "/HWD"])
    P64 = _views(case)
    o = oracle_mod.Oracle(M=8, D=D, N=len(P64), F=4, H=H, W=W, padding=PADDING,
                          bbox=[-1, -1, -1, 1, 1, 1], grid_shape=(2, 2, 2))
    start, end = Z[case + "/start"], Z[case + "/end"]
    got = o.feature_indices(P64.astype(np.float32), start, end)             # [n, V, D, 2]
    for name in ("pixels64", "pixels32"):
        pix = Z[case + "/" + name].transpose(1, 0, 2, 3)                     # [n, V, D, 2]
        fx, fy, clear = _expected_indices(pix.astype(np.float64), H, W)
        assert clear.mean() > 0.99
        assert np.array_equal(got[..., 0][clear], fx[clear]), (case, name)
        assert np.array_equal(got[..., 1][clear], fy[clear]), (case, name)
    # float32 and float64 inputs of the reference's function agree to a small fraction of a pixel
    # on these scenes: the tolerance band above is two orders of magnitude wider
    d = np.abs(Z[case + "/pixels64"] - Z[case + "/pixels32"])
    assert np.nanmax(d) < 5e-3, np.nanmax(d)
    # both regimes occur: projections inside the views and projections that clamp
    fx, fy, _ = _expected_indices(Z[case + "/pixels64"], H, W)
    assert (fx > 0).any() and ((fx == 0) | (fx == W) | (fy == H)).any()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_sweep_indices_follow_the_reference_projection(oracle_mod, case):
    import torch
    from raynet_amd.hip_implementations import get_context
    H, W, D = (int(v) for v in Z[case + "/HWD"])
    P64 = _views(case)
    V = len(P64)
    ctx = get_context(M=64, D=D, N=V, F=32, H=H, W=W, padding=PADDING, bbox=(-1, -1, -1, 1, 1, 1),
                      grid_shape=(4, 4, 4))
    start, end = Z[case + "/start"], Z[case + "/end"]
    n = len(start)
    out = torch.full((n, V, D, 2), -1, dtype=torch.int32, device="cuda")
    ctx.selftest_feature_offsets(ctx.dev(P64.astype(np.float32).reshape(-1)), ctx.dev(start),
                                 ctx.dev(end), out)
    out = out.cpu().numpy().astype(np.int64)
    Wf = W + PADDING + 1
    o = oracle_mod.Oracle(M=8, D=D, N=V, F=4, H=H, W=W, padding=PADDING,
                          bbox=[-1, -1, -1, 1, 1, 1], grid_shape=(2, 2, 2))
    idx = o.feature_indices(P64.astype(np.float32), start, end).astype(np.int64)
    lin_oracle = idx[..., 1] * Wf + idx[..., 0]
    # bit-exact with the oracle, both forms, every sample (boundaries included)
    assert np.array_equal(out[..., 0], lin_oracle) and np.array_equal(out[..., 1], lin_oracle)
    # and with the reference's own projection wherever that is clear of a rounding boundary
    pix = Z[case + "/pixels64"].transpose(1, 0, 2, 3)
    fx, fy, clear = _expected_indices(pix, H, W)
    for form in (0, 1):
        assert np.array_equal(out[..., form][clear], (fy * Wf + fx)[clear]), (case, form)


@pytest.mark.gpu
def test_hip_sweep_indices_on_zero_sums_and_zero_denominators(oracle_mod):
    """The cooperative sweep starts the projection's numerators at their first product instead of
    `0.0f +` (raynet_kernels.h, project_fast): sums that come out as a zero of the other sign, zero
    and negative-zero matrix entries, points with zero / negative-zero coordinates, and a
    denominator that is exactly +0 or -0 (infinite or NaN quotients) must still give the
    literal form's -- and the oracle's -- index, every sample."""
    import torch
    from raynet_amd.hip_implementations import get_context
    H, W, D, V = 24, 32, 8, 5
    rng = np.random.default_rng(17)
    vals = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 3.0, -7.0, 1e-30, -1e-30], np.float32)
    n = 4096
    P = vals[rng.integers(0, 6, (n // 64, V, 12))].astype(np.float32)    # many exact cancellations
    P[::3, :, 8:12] = vals[rng.integers(0, 2, (len(P[::3]), V, 4))]       # n = +-0 rows
    P[1::5, :, 0:4] = vals[rng.integers(0, 2, (len(P[1::5]), V, 4))]      # x = +-0 rows
    pts = vals[rng.integers(0, len(vals), (n, 3))].astype(np.float32)
    ctxs = {}
    bad = 0
    for g in range(P.shape[0]):                  # one matrix set per 64 rays (a context per call is cheap)
        start = pts[64 * g:64 * g + 64]
        ctx = ctxs.setdefault(0, get_context(M=64, D=D, N=V, F=32, H=H, W=W, padding=PADDING,
                                             bbox=(-1, -1, -1, 1, 1, 1), grid_shape=(4, 4, 4)))
        out = torch.full((64, V, D, 2), -1, dtype=torch.int32, device="cuda")
        ctx.selftest_feature_offsets(ctx.dev(P[g].reshape(-1)), ctx.dev(start), ctx.dev(start), out)
        out = out.cpu().numpy().astype(np.int64)
        o = oracle_mod.Oracle(M=8, D=D, N=V, F=4, H=H, W=W, padding=PADDING,
                              bbox=[-1, -1, -1, 1, 1, 1], grid_shape=(2, 2, 2))
        idx = o.feature_indices(P[g], start, start).astype(np.int64)
        lin = idx[..., 1] * (W + PADDING + 1) + idx[..., 0]
        bad += int((out[..., 0] != lin).sum() + (out[..., 1] != lin).sum())
    assert bad == 0, bad
