"""a5 / a6 / a9 in the SATURATED, coupled regime, against the reference's own mrf/mrf_np.py.

tests/golden/ref_mrf_np_saturated.npz (generator: tests/golden/gen_saturated_from_reference.py)
holds what `belief_propagation` + `compute_depth_distribution` of the reference produce on
96,000 coupled rays (5 views, 48^3 voxels, 3 iterations) whose accumulators reach |119|
in log-odds.  The inputs are rebuilt by tests/saturated_case.py and verified by SHA-256.

What is held to it:
  * the C oracle in its ROBUST message form (`set_robust_messages(True)`: suffix sum scanned
    directly, log pos - log neg) -- the form the HIP kernels implement and the comparator of
    the full-size parity runs (tools/fullsize_parity.py);
  * the HIP path through the C ABI: the K3 / K4 entry points behind
    `get_bp_backend("hip")`, and the resident-scene kernels (packed voxel lists, patch-ordered
    rows, LDS-box scatter, bricked accumulators) -- `gpu` tests.
And what is recorded, not required: the oracle's LITERAL restatement of mrf_bp.cu:136-167
(fp32 running sums, (cumsum1 - cumsum2)) goes non-finite on this scene where the reference's
NumPy path (float64 cumulative sums, mrf_np.py:78-112) stays finite -- the reason the robust
form exists.

Tolerances (fp32 against the reference's mixed float32 / float64, three coupled iterations):
messages <= 1e-4 abs, accumulator <= 1e-4 x the number of messages summed
into the voxel and <= 1e-4 of the largest |acc - prior|, distributions <= 5e-5 abs, arg-max
identical except at near-ties (two best probabilities within 5e-5 in the reference)."""
import numpy as np
import pytest

import saturated_case as C

MSG_TOL = 1e-4
DIST_TOL = 5e-5
TIE = 5e-5


@pytest.fixture(scope="module")
def case(oracle_mod):
    g = dict(np.load(C.FIXTURE))
    inp = C.build_inputs(oracle_mod)
    assert inp["sha256"] == bytes(g["sha256"]).decode(), \
        "saturated inputs not reproduced on this machine: nothing can be compared"
    assert int(g["nonfinite"].sum()) == 0          # the reference stays finite
    hits = np.zeros(C.GRID, np.int64)
    rvi, rvc = inp["rvi"], inp["rvc"]
    live = (np.arange(C.M)[None, :] < rvc[:, None]) & (rvc[:, None] > 1)
    v = rvi[live]
    np.add.at(hits, (v[:, 0], v[:, 1], v[:, 2]), 1)
    inp["hits"] = hits
    inp["prior"] = np.float32(np.log(C.GAMMA) - np.log(1 - C.GAMMA))
    return inp, g


def check_against_reference(inp, g, accs, msgs, S_new, what):
    prior = inp["prior"]
    report = {}
    for it in range(len(accs)):
        ref = g["accs"][it]
        d = np.abs(accs[it] - ref)
        assert np.isfinite(accs[it]).all(), "%s: non-finite accumulator, iteration %d" % (what, it)
        assert np.all(d <= MSG_TOL * np.maximum(inp["hits"], 1)), (what, it, float(d.max()))
        assert d.max() <= 1e-4 * np.abs(ref - prior).max(), (what, it, float(d.max()))
        report["acc_it%d" % it] = float(d.max())
    sub = slice(None, None, C.SUBSAMPLE)
    d = np.abs(msgs[sub] - g["msgs_sub"])
    assert d.max() <= MSG_TOL, (what, float(d.max()))
    report["msgs"] = float(d.max())
    # all rays, in aggregate: sum of a ray's messages (<= count x tolerance)
    ds = np.abs(msgs.astype(np.float64).sum(1) - g["msg_sum"])
    assert np.all(ds <= MSG_TOL * np.maximum(inp["rvc"], 1) + 1e-5 * np.abs(g["msg_sum"]))
    d = np.abs(S_new[sub] - g["S_new_sub"])
    assert d.max() <= DIST_TOL, (what, float(d.max()))
    report["S_new"] = float(d.max())
    flips = S_new.argmax(1) != g["argmax"]
    assert not np.any(flips & (g["top2_gap"] > TIE)), \
        "%s: %d arg-max differences away from a near-tie" % (what, int((flips & (g["top2_gap"] > TIE)).sum()))
    report["argmax_flips_at_near_ties"] = int(flips.sum())
    return report


@pytest.fixture
def robust(oracle_mod):
    oracle_mod.Oracle.set_robust_messages(True)
    yield
    oracle_mod.Oracle.set_robust_messages(False)


def _oracle(oracle_mod):
    return oracle_mod.Oracle(M=C.M, D=8, N=2, F=4, H=C.H, W=C.W, padding=1, bbox=C.BBOX,
                             grid_shape=C.GRID, threads=min(8, oracle_mod.Oracle.max_threads()))


def _run_oracle(oracle_mod, inp):
    o = _oracle(oracle_mod)
    accs = []
    msgs = np.zeros_like(inp["S"])
    acc, msgs = o.belief_propagation(inp["S"], inp["rvi"], inp["rvc"], msgs, gamma=C.GAMMA,
                                     bp_iterations=C.ITERS,
                                     callback=lambda it, a, m: accs.append(a.copy()))
    S_new = o.depth_distribution(inp["S"], inp["rvi"], inp["rvc"], acc, msgs)
    return np.stack(accs), msgs, S_new


def test_oracle_robust_form_vs_reference_numpy(oracle_mod, case, robust):
    inp, g = case
    accs, msgs, S_new = _run_oracle(oracle_mod, inp)
    rep = check_against_reference(inp, g, accs, msgs, S_new, "oracle (robust form)")
    assert rep["argmax_flips_at_near_ties"] <= 20


def test_literal_cu_arithmetic_leaves_the_reference_numpy_path_here(oracle_mod, case):
    """Record, not requirement: mrf_bp.cu's fp32 (cumsum1 - cumsum2) cancels to zero in front
    of a saturated voxel, p rounds to 1 and the message is +inf; mrf_np.py forms the same
    sums in float64 and stays finite.  The first iteration (no saturation yet) agrees."""
    inp, g = case
    accs, msgs, S_new = _run_oracle(oracle_mod, inp)
    assert np.abs(accs[0] - g["accs"][0]).max() <= 1e-4
    assert (~np.isfinite(accs[1:])).sum() > 0
    assert (~np.isfinite(g["accs"])).sum() == 0


# ------------------------------------------------------------------ HIP, through the C ABI
@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (no CPU fallback exists)"
    from raynet_amd import _lib
    _lib.build()
    return torch


@pytest.mark.gpu
def test_hip_bp_plugin_vs_reference_numpy(torch, case):
    """K3 / K4 behind the reference's plugin interface (mrf/bp_inference.py:340-409)."""
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.mrf.bp_inference import get_bp_backend
    inp, g = case
    S, rvi, rvc = inp["S"], inp["rvi"], inp["rvc"]
    # accumulator after every iteration: the sweeps one by one through the K3 closure
    ctx = get_context(M=C.M, grid_shape=C.GRID)
    from raynet_amd.mrf.mrf_hip import batch_ray_belief_propagation
    bp, de = batch_ray_belief_propagation(C.M, C.GRID)
    prior = float(inp["prior"])
    acc = torch.full(C.GRID, prior, device="cuda")
    out = torch.full(C.GRID, prior, device="cuda")
    S_d, rvi_d, rvc_d = ctx.dev(S), ctx.dev(rvi), ctx.dev(rvc)
    msgs_d = torch.zeros(S.shape, device="cuda")
    accs = []
    for it in range(C.ITERS):
        bp(S_d, rvi_d, rvc_d, acc, msgs_d, out)
        acc, out = out, acc
        out.fill_(prior)
        accs.append(acc.cpu().numpy())
    S_new = de(S_d, rvi_d, rvc_d, acc, msgs_d, torch.zeros(S.shape, device="cuda")).cpu().numpy()
    check_against_reference(inp, g, np.stack(accs), msgs_d.cpu().numpy(), S_new, "HIP K3/K4")
    # and the plugin's one-call form
    gp = GenerationParameters(grid_shape=np.array(C.GRID, np.int32),
                              max_number_of_marched_voxels=C.M)
    plugin = get_bp_backend("hip", gp, bp_iterations=C.ITERS, batch_size=40000)
    msgs = np.zeros_like(S)
    acc_p, msgs = plugin.update_bp_messages(S, rvi, rvc, msgs)
    assert np.abs(acc_p - g["accs"][-1]).max() <= 1e-4 * np.abs(g["accs"][-1] - prior).max()
    assert np.abs(msgs[::C.SUBSAMPLE] - g["msgs_sub"]).max() <= MSG_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("deterministic", [False, True])
def test_hip_resident_kernels_vs_reference_numpy(torch, case, deterministic):
    """The kernels bench.py times: packed lists, rows in 16x16 patches, LDS-box scatter into a
    partial accumulator, combine with the prior, bricked accumulators; also in the
    fixed-point (deterministic) mode."""
    from raynet_amd.forward_pass import tile_order
    from raynet_amd.hip_implementations import get_context
    inp, g = case
    ctx = get_context(M=C.M, grid_shape=C.GRID)
    prior = float(inp["prior"])
    n_img = C.H * C.W
    npad = (n_img + 255) // 256 * 256
    rays = tile_order(torch.arange(n_img, dtype=torch.int32, device="cuda"), C.H, C.W, 16, 16)
    rows = torch.cat([rays.long() + v * n_img for v in range(C.VIEWS)])           # row -> ray
    dst = torch.cat([torch.arange(n_img, device="cuda") + v * npad for v in range(C.VIEWS)])
    S = ctx.dev(inp["S"])
    # resident column: clipped + renormalised (mrf_bp.cu:103-111), as k_sweep_map stores it
    cnt = ctx.dev(inp["rvc"]).long()
    valid = torch.arange(C.M, device="cuda")[None, :] < cnt[:, None]
    Sc = torch.where(valid, S.clamp(1e-5, 1 - 1e-5), torch.zeros_like(S))
    Sc = Sc / Sc.sum(1, keepdim=True).clamp_min(1e-30)
    rvi = ctx.dev(inp["rvi"])
    packed = (rvi[..., 0] << 20) | (rvi[..., 1] << 10) | rvi[..., 2]
    total = C.VIEWS * npad
    Sr = torch.zeros((total, C.M), device="cuda")
    vox = torch.zeros((total, C.M), dtype=torch.int32, device="cuda")
    rvc = torch.zeros((total,), dtype=torch.int32, device="cuda")
    Sr[dst] = Sc[rows]
    vox[dst] = packed[rows].to(torch.int32)
    rvc[dst] = ctx.dev(inp["rvc"])[rows]
    msgs = torch.empty((total, C.M), device="cuda")
    G = ctx.acc_size()
    acc_in = torch.full((G,), prior, device="cuda")
    acc_next = torch.empty((G,), device="cuda")
    part = (torch.zeros((G,), dtype=torch.int64, device="cuda") if deterministic
            else torch.zeros((1, G), device="cuda"))
    ctx.scatter_reset()
    accs = []
    for it in range(C.ITERS):
        if deterministic:
            ctx.scene_bp_sweep_fixed(Sr, vox, rvc, acc_in, msgs, part, first_sweep=it == 0,
                                     patch_rows=True, uniform_acc=it == 0)
            ctx.acc_combine_fixed(part, prior, acc_next)
        else:
            ctx.scene_bp_sweep(Sr, vox, rvc, acc_in, msgs, part, first_sweep=it == 0,
                               patch_rows=True, uniform_acc=it == 0)
            ctx.acc_combine(part, prior, acc_next)
        acc_in, acc_next = acc_next, acc_in
        accs.append(ctx.acc_to_grid(acc_in).cpu().numpy())
    S_new = torch.zeros((total, C.M), device="cuda")
    ctx.set_voxel_grid(ctx.dev(_voxel_grid()))
    ctx.scene_depth(Sr, vox, rvc, acc_in, msgs, None, S_new, None)
    # back to ray order; tails beyond the count are not written by the kernels
    live = (torch.arange(C.M, device="cuda")[None, :] < rvc[:, None]) & (rvc[:, None] > 1)
    msgs = torch.where(live, msgs, torch.zeros_like(msgs))
    m_ray = torch.zeros((C.VIEWS * n_img, C.M), device="cuda")
    s_ray = torch.zeros((C.VIEWS * n_img, C.M), device="cuda")
    m_ray[rows] = msgs[dst]
    s_ray[rows] = S_new[dst]
    check_against_reference(inp, g, np.stack(accs), m_ray.cpu().numpy(), s_ray.cpu().numpy(),
                            "HIP resident kernels" + (" (fixed point)" if deterministic else ""))


def _voxel_grid():
    from oracle import oracle
    return oracle.voxel_grid_centers(C.BBOX, C.GRID)
