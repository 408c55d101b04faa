"""oracle/cpu_reference.py (the NumPy restatement of mrf/mrf_np.py that bench.py times as the
reference's own CPU path) against outputs of the reference's functions: bit for bit."""
import numpy as np
import pytest

from conftest import load_cases

MRF = load_cases("ref_mrf_np.npz")


@pytest.mark.parametrize("case", sorted(MRF))
def test_numpy_restatement_is_bit_equal_to_mrf_np(case):
    from oracle import cpu_reference as R
    c = MRF[case]
    accs = []
    msgs = np.random.default_rng(3).random(c["S"].shape).astype(np.float32)   # must be ignored
    acc, msgs = R.belief_propagation(c["S"], c["rvi"], c["rvc"], msgs, c["grid"], gamma=0.05,
                                     bp_iterations=3,
                                     callback=lambda it, a, m: accs.append(a.copy()))
    S_new = R.compute_depth_distribution(c["S"], c["rvi"], c["rvc"], msgs, acc)
    assert acc.dtype == np.float32 and msgs.dtype == np.float32 and S_new.dtype == np.float32
    assert np.array_equal(np.stack(accs), c["accs"])
    assert np.array_equal(msgs, c["msgs"])
    assert np.array_equal(S_new, c["S_new"])


def test_numpy_restatement_on_the_saturated_scene(oracle_mod):
    """One BP iteration over the 96,000 coupled rays of the saturated golden: the accumulator
    equals the reference's after its first iteration, bit for bit (~7 s of the per-ray loop)."""
    import saturated_case as C
    from oracle import cpu_reference as R
    g = np.load(C.FIXTURE)
    inp = C.build_inputs(oracle_mod)
    assert inp["sha256"] == bytes(g["sha256"]).decode(), "saturated inputs not reproduced"
    msgs = np.zeros_like(inp["S"])
    acc, msgs = R.belief_propagation(inp["S"], inp["rvi"], inp["rvc"], msgs, C.GRID,
                                     gamma=C.GAMMA, bp_iterations=1)
    assert np.array_equal(acc, g["accs"][0])
