"""Oracle vs the reference's CUDA device functions executed on the host
(tests/golden/gen_cu_crosscheck.py -- supplementary evidence for a1/a2/a4 and
the fused composition, which the reference itself never tests).  CPU only."""
import numpy as np
import pytest

from conftest import load_cases

CU = load_cases("crosscheck_cu_host.npz")


def make(oracle_mod, c):
    o = oracle_mod.Oracle(M=int(c["M"]), D=int(c["D"]), N=int(c["N"]), F=int(c["F"]),
                          H=int(c["H"]), W=int(c["W"]), padding=int(c["padding"]),
                          bbox=c["bbox"], grid_shape=c["grid"])
    rng = np.random.default_rng(int(c["seed"]))
    feats = rng.standard_normal((o.N, o.H + o.padding + 1, o.W + o.padding + 1, o.F),
                                dtype=np.float32) * np.float32(0.25)
    vg = oracle_mod.voxel_grid_centers(c["bbox"], c["grid"])
    return o, feats, vg


@pytest.mark.parametrize("case", sorted(CU))
def test_stages_match_cu_host(oracle_mod, case):
    c = CU[case]
    o, feats, vg = make(oracle_mod, c)
    starts, ends = o.sample(c["ray_idxs"], c["P_inv"], c["center"])
    # a1: same fp32/fp64 mix, no transcendental -> identical bits
    assert np.array_equal(starts, c["starts"]) and np.array_equal(ends, c["ends"])
    # a2: sequential fp32 dot + expf; glibc expf both sides here
    S = o.similarities(feats, c["P"], starts, ends)
    assert np.abs(S - c["S"]).max() <= 1e-7
    # a3: CUDA-flavour traversal equals the Cython-flavour oracle on this bbox
    rvi, rvc = o.traversal(starts, ends)
    assert np.array_equal(rvc, c["rvc"]) and np.array_equal(rvi, c["rvi"].astype(np.int32))
    # a4
    Sv = o.planes_to_voxels(vg, rvi, rvc, starts, ends, S)
    assert np.abs(Sv - c["S_voxel"]).max() <= 1e-7


@pytest.mark.parametrize("case", sorted(CU))
def test_fused_matches_cu_host(oracle_mod, case):
    c = CU[case]
    o, feats, vg = make(oracle_mod, c)
    prior = o.prior(float(c["gamma"]))
    acc_out = prior.copy()
    msgs = np.zeros((len(c["ray_idxs"]), o.M), np.float32)
    rvi, rvc, Sv = o.fused_bp(c["ray_idxs"], feats, c["P"], c["P_inv"], c["center"], vg, prior,
                              msgs, acc_out)
    assert np.array_equal(rvc, c["rvc"])
    assert np.abs(msgs - c["msgs"]).max() <= 2e-5
    assert np.abs(acc_out - c["acc_out"]).max() <= 1e-4
    _, _, S_new, depth = o.fused_depth(c["ray_idxs"], feats, c["P"], c["P_inv"], c["center"],
                                       vg, acc_out, msgs)
    assert np.abs(S_new - c["S_new"]).max() <= 1e-6
    assert np.abs(depth - c["depth"]).max() <= 1e-5


@pytest.mark.parametrize("case", ["small", "wide"])
def test_plane_sweep_against_a_float64_statement(oracle_mod, case):
    """a2 has no reference-held pin.  Second opinion of another kind:
    feature_similarities.cu:10-124 stated once more, in NumPy float64 and vectorised --
    projection with the reference's own `project` semantics (utils/geometry.py:9-34: P x,
    divide by the last row), round half away from zero (CUDA `round`), the pixel -> feature
    shift `+ padding - (padding - 1) / 2` in integer arithmetic, clamps to [0, W] x [0, H],
    collapse to (0, 0) when either clamps to 0, mean of the pairwise dot products, softmax.
    Indices must agree wherever float64 is not within 1e-3 of a rounding boundary, the
    distributions to 1e-6."""
    c = CU[case]
    o, feats, _ = make(oracle_mod, c)
    starts, ends = o.sample(c["ray_idxs"], c["P_inv"], c["center"])
    n, N, D, pad = len(starts), o.N, o.D, o.padding
    idx = o.feature_indices(c["P"], starts, ends)                       # [n, N, D, 2]
    k = np.arange(D, dtype=np.float64)[None, :, None]
    s, e = starts.astype(np.float64), ends.astype(np.float64)
    pts = s[:, None, :] + k * (e - s)[:, None, :] / (D - 1)             # [n, D, 3]
    ph = np.concatenate([pts, np.ones((n, D, 1))], axis=2)
    P = c["P"].astype(np.float64).reshape(N, 3, 4)
    proj = np.einsum("vij,ndj->nvdi", P, ph)                            # [n, N, D, 3]
    xy = proj[..., :2] / proj[..., 2:]
    frac = np.abs(xy - np.trunc(xy))
    safe = np.all(np.abs(frac - 0.5) > 1e-3, axis=-1) & np.all(np.isfinite(xy), axis=-1)
    r = np.sign(xy) * np.floor(np.abs(xy) + 0.5)                        # half away from zero
    half = (pad - 1) // 2
    fx = np.clip(r[..., 0] + pad - half, 0, o.W).astype(np.int64)
    fy = np.clip(r[..., 1] + pad - half, 0, o.H).astype(np.int64)
    zero = (fx == 0) | (fy == 0)
    fx[zero] = 0
    fy[zero] = 0
    assert safe.mean() > 0.99
    assert np.array_equal(idx[..., 0][safe], fx[safe]) and np.array_equal(idx[..., 1][safe], fy[safe])
    # similarities from the ORACLE's indices (so that a boundary case cannot differ), float64
    f64 = feats.astype(np.float64)
    vecs = np.stack([f64[v][idx[:, v, :, 1], idx[:, v, :, 0]] for v in range(N)], axis=1)  # [n,N,D,F]
    tot = np.zeros((n, D))
    for i in range(N):
        for j in range(i + 1, N):
            tot += (vecs[:, i] * vecs[:, j]).sum(-1)
    tot /= N * (N - 1) / 2
    tot -= tot.max(1, keepdims=True)
    p = np.exp(tot)
    p /= p.sum(1, keepdims=True)
    S = o.similarities(feats, c["P"], starts, ends)
    assert np.abs(S - p).max() <= 1e-6
