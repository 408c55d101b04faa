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
