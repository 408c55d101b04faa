"""Inputs of the coupled, saturated BP golden (tests/golden/ref_mrf_np_saturated.npz).

Five ring cameras look at a 48^3 grid with 120x160 rays each (96,000 rays, M = 160): every
voxel is crossed by ~45 rays, the rays agree on a planted surface, so after three BP
iterations the accumulators reach the hundreds in log-odds -- the regime the 48-ray fixtures
of ref_mrf_np.npz never reach and bench.py's scene lives in.

The inputs are REBUILT here by generator and tests alike (96,000 x 160 columns do not belong
in a fixture); they are made of operations that are bit-reproducible on any IEEE machine:
  * ray segments and voxel lists from the C oracle (rno_batch_sample, rno_batch_traversal:
    +, -, *, / in fp32 / fp64, built with -ffp-contract=off; the traversal is pinned
    bit-exact to the reference's compiled Cython by tests/test_oracle_golden.py);
  * the per-ray voxel-space column S: a rational bump around the voxel nearest to where the
    ray meets a sphere / a ground plane (sqrt, +, *, /), times an integer-hash ripple.
The fixture records a SHA-256 of the inputs; the tests refuse to compare anything if the
rebuilt inputs differ."""
import hashlib
import os

import numpy as np

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                       "ref_mrf_np_saturated.npz")

H, W, VIEWS, M = 120, 160, 5, 160
GRID = (48, 48, 48)
BBOX = np.array([-1, -1, -1, 1, 1, 1], np.float32)
GAMMA = 0.05
ITERS = 3
SUBSAMPLE = 128         # every 128th ray's messages / S_new are kept in the fixture


def _dot3(a, b):
    # explicit order: a reduction's summation order is an implementation detail of NumPy
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def build_inputs(oracle_mod):
    """-> dict(S [n,M] f32, rvi [n,M,3] i32, rvc [n] i32, centers [VIEWS,3], view [n] i32,
    sha256 str); rays of view v are rows [v*H*W, (v+1)*H*W) in ray-index order."""
    from raynet_amd.synthetic import ring_cameras
    o = oracle_mod.Oracle(M=M, D=8, N=2, F=4, H=H, W=W, padding=1, bbox=BBOX, grid_shape=GRID,
                          threads=oracle_mod.Oracle.max_threads())
    cams = ring_cameras(VIEWS, H, W, focal=1.5 * H)
    ridx = np.arange(H * W, dtype=np.int32)
    S_all, rvi_all, rvc_all, centers = [], [], [], []
    vg = oracle_mod.voxel_grid_centers(BBOX, GRID).astype(np.float64)
    for v, cam in enumerate(cams):
        center = np.asarray(cam.center, np.float32).ravel()
        s, e = o.sample(ridx, np.asarray(cam.P_pinv, np.float32), center)
        rvi, rvc = o.traversal(s, e)
        # where the ray meets the planted scene: sphere |x - c0| = 0.55, else the ground
        # plane z = -0.7 inside radius 0.95, else nothing (flat column)
        org = s.astype(np.float64)
        d = e.astype(np.float64) - org
        L = np.sqrt(_dot3(d, d))
        d /= np.maximum(L, 1e-30)[:, None]
        c0 = np.array([0.0, 0.0, -0.1])
        oc = org - c0
        b = _dot3(oc, d)
        disc = b * b - (_dot3(oc, oc) - 0.55 ** 2)
        t_sph = np.where(disc > 0, -b - np.sqrt(np.maximum(disc, 0.0)), np.inf)
        t_sph = np.where(t_sph > 0, t_sph, np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            t_gr = (-0.7 - org[:, 2]) / d[:, 2]
            pg = org + t_gr[:, None] * d
        ok = np.isfinite(t_gr) & (t_gr > 0) & ((pg[:, 0] ** 2 + pg[:, 1] ** 2) < 0.95 ** 2)
        t_gr = np.where(ok, t_gr, np.inf)
        t_hit = np.minimum(t_sph, t_gr)
        hit = np.isfinite(t_hit) & (t_hit < L)
        # index of the list voxel nearest to the hit point
        n = len(ridx)
        i = np.arange(M)[None, :]
        valid = i < rvc[:, None]
        pts = vg[rvi[..., 0], rvi[..., 1], rvi[..., 2]]            # [n, M, 3] voxel centres
        tv = _dot3(pts - org[:, None, :], d[:, None, :])
        dist = np.where(valid, np.abs(tv - np.where(hit, t_hit, 0.0)[:, None]), np.inf)
        peak = dist.argmin(1)
        x = (i - peak[:, None]).astype(np.float64) / 1.5
        bump = np.where(hit[:, None], 60.0 / (1.0 + x * x), 0.0)
        # integer-hash ripple in [0.75, 1.25): the same bits everywhere
        hsh = ((ridx.astype(np.uint64)[:, None] * np.uint64(2654435761) +
                i.astype(np.uint64) * np.uint64(40503) + np.uint64(v * 977)) >> np.uint64(7)) & np.uint64(0xffff)
        ripple = 0.75 + 0.5 * (hsh.astype(np.float64) / 65536.0)
        col = np.where(valid, (1.0 + bump) * ripple, 0.0)
        tot = np.cumsum(col, axis=1)[:, -1]      # strictly sequential
        S = np.where(valid, col / np.maximum(tot, 1e-30)[:, None], 0.0).astype(np.float32)
        S_all.append(S)
        rvi_all.append(rvi)
        rvc_all.append(rvc)
        centers.append(center[:3])
    S = np.ascontiguousarray(np.concatenate(S_all))
    rvi = np.ascontiguousarray(np.concatenate(rvi_all)).astype(np.int32)
    rvc = np.ascontiguousarray(np.concatenate(rvc_all)).astype(np.int32)
    h = hashlib.sha256()
    for a in (S, rvi, rvc):
        h.update(a.tobytes())
    view = np.repeat(np.arange(VIEWS, dtype=np.int32), H * W)
    return dict(S=S, rvi=rvi, rvc=rvc, centers=np.array(centers, np.float32), view=view,
                sha256=h.hexdigest())
