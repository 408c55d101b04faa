"""RayNetForwardPass (the reference's ForwardPass API) on the GPU: the resident
schedule vs the literal K1/K2 schedule vs the oracle, the MV-CNN twin as model, the
other two drivers, and size-independent properties at BASELINE.json's full size."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (no CPU fallback exists)"
    from raynet_amd import _lib
    _lib.build()
    return torch


def _gp(D, M, grid, neighbors=4, padding=11):
    from raynet_amd.common.generation_parameters import GenerationParameters
    return GenerationParameters(depth_planes=D, neighbors=neighbors,
                                grid_shape=np.array(grid, np.int32),
                                max_number_of_marched_voxels=M, padding=padding, gamma_mrf=0.05)


def _oracle_forward(oracle_mod, scene, bank, gp, refs, H, W, iters=3, quirks=False, F=32):
    o = oracle_mod.Oracle(M=gp.max_number_of_marched_voxels, D=gp.depth_planes,
                          N=gp.neighbors + 1, F=F, H=H, W=W, padding=gp.padding,
                          bbox=scene.bbox.ravel(), grid_shape=gp.grid_shape,
                          threads=oracle_mod.Oracle.max_threads())
    vg = oracle_mod.voxel_grid_centers(scene.bbox.ravel(), gp.grid_shape)
    ridx = np.arange(H * W, dtype=np.int32)
    cams = {}
    for r in refs:
        views = scene.view_indices_with_neighbors(r, gp.neighbors)
        cams[r] = (bank.stacked(views).cpu().numpy(),
                   np.array([scene.get_image(v).camera.P for v in views], np.float32),
                   scene.get_image(r).camera.P_pinv.astype(np.float32),
                   scene.get_image(r).camera.center.ravel().astype(np.float32))
    acc = o.prior(0.05)
    msgs = {r: np.zeros((H * W, o.M), np.float32) for r in refs}
    for it in range(iters):
        out = o.prior(0.05)
        for r in refs:
            if quirks and it > 0:
                msgs[r][...] = 0
            f, P, Pi, c = cams[r]
            o.fused_bp(ridx, f, P, Pi, c, vg, acc, msgs[r], out)
        acc = out
    depths, dists = [], []
    for r in refs:
        f, P, Pi, c = cams[r]
        m = msgs[refs[-1]] if quirks else msgs[r]
        _, _, S_new, depth = o.fused_depth(ridx, f, P, Pi, c, vg, acc, m)
        depths.append(depth.reshape(W, H).T)
        dists.append(S_new)
    return acc, msgs, depths, dists


def _depth_close(a, b, dist=None, W=None, H=None):
    """<= 1e-4 everywhere except arg-max near-ties (a 1-ulp change flips a voxel)."""
    d = np.abs(a - b)
    bad = d > 1e-4
    if dist is not None and bad.any():
        top = np.sort(dist, axis=1)[:, -2:]
        gap = (top[:, 1] - top[:, 0]).reshape(W, H).T
        assert np.all(gap[bad] <= 5e-5), "depth differs away from a tie"
    return bad.mean()


@pytest.mark.parametrize("quirks,M", [(False, 96), (True, 96), (False, 100)])
def test_resident_vs_reference_schedule_vs_oracle(torch, oracle_mod, quirks, M):
    """M = 100: not a multiple of 16 and no list can reach it (32 + 32 + 32 - 2 = 94): the resident
    buffers' rows are 112 long (RayNetForwardPass._row_stride), the results those of M = 100 (the
    literal K1 / K2 schedule and the oracle run with rows of exactly 100)."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    H, W, D, grid = 24, 32, 16, (32, 32, 32)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(D, M, grid)
    refs = (0, 3, 1)
    cls = get_forward_pass_factory("raynet")
    fa = cls(bank, gp, "sample_in_bbox", (H, W), 300, reference_quirks=quirks)
    da = list(fa.forward_pass(scene, refs))
    assert fa._rows_M() == (112 if M == 100 else 96)
    assert tuple(fa.messages[0].shape) == (H * W, M)
    fb = cls(bank, gp, "sample_in_bbox", (H, W), 300, schedule="reference",
             reference_quirks=quirks)
    db = list(fb.forward_pass(scene, refs))
    assert fb._rows_M() == M
    assert len(da) == len(db) == 3 and da[0].shape == (H, W) and da[0].dtype == np.float32
    # the comparator is the oracle's robust message form (pinned to the reference's NumPy path,
    # tests/test_saturated_golden.py): on this planted scene messages reach |m| = 12, where the
    # literal fp32 (cumsum1 - cumsum2) of mrf_bp.cu:157 is itself wrong by 1e-2
    oracle_mod.Oracle.set_robust_messages(True)
    try:
        acc_o, msgs_o, depth_o, dist_o = _oracle_forward(oracle_mod, scene, bank, gp, [0, 1, 2],
                                                         H, W, quirks=quirks)
    finally:
        oracle_mod.Oracle.set_robust_messages(False)
    acc_a, acc_b = fa.accumulator.cpu().numpy(), fb.accumulator.cpu().numpy()
    assert np.abs(acc_a - acc_b).max() <= 2e-4
    assert np.abs(acc_a - acc_o).max() <= 2e-4
    for i, r in enumerate([0, 1, 2]):
        assert _depth_close(da[i], depth_o[i], dist_o[i], W, H) <= 0.01
        assert _depth_close(db[i], depth_o[i], dist_o[i], W, H) <= 0.01
        rows = fa.messages[r].cpu().numpy()        # row i belongs to ray fa.ray_index[r][i]
        m = np.zeros_like(rows)
        m[fa.ray_index[r].cpu().numpy().astype(np.int64)] = rows
        err = np.abs(m - msgs_o[r])
        assert err.max() <= 1e-4, "worst %g at |m|=%g" % (
            err.max(), np.abs(msgs_o[r]).ravel()[err.argmax()])
        assert np.abs(fb.messages[r].cpu().numpy() - msgs_o[r]).max() <= 1e-4


def test_mvcnn_twin_as_model_and_other_drivers(torch, oracle_mod):
    """End to end with the PyTorch MV-CNN twin producing the features
    (models.py:90-111 architecture), through all three drivers."""
    from raynet_amd.common.camera import Camera
    from raynet_amd.common.scene import Image, Scene
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.models import get_nn
    from raynet_amd.synthetic import ring_cameras
    H, W = 20, 28
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    cams = ring_cameras(5, H, W, focal=1.5 * H)
    scene = Scene([Image(rng.random((H, W, 3)).astype(np.float32), c) for c in cams],
                  [-1, -1, -1, 1, 1, 1])
    model = get_nn("simple_cnn")().cuda()
    gp = _gp(16, 64, (16, 16, 16))
    feats = model.predict(np.zeros((1, H + 22, W + 22, 3), np.float32))
    assert tuple(feats.shape) == (1, H + 12, W + 12, 32)     # H + padding + 1
    out = {}
    for name in ("multi_view_cnn", "multi_view_cnn_voxel_space", "raynet"):
        fp = get_forward_pass_factory(name)(model, gp, "sample_in_bbox", (H, W), 1000)
        out[name] = list(fp.forward_pass(scene, (0, 2, 1)))
        assert len(out[name]) == 2
        for d in out[name]:
            assert d.shape == (H, W) and d.dtype == np.float32 and np.isfinite(d).all()
            assert (d > 0.0).all() and (d < 8.0).all(), (name, d.min(), d.max())   # ring radius 3


def test_full_size_properties(torch):
    """BASELINE.json config 2 size (480x640 rays, 64 planes, 128^3, M=384), two
    reference images: size-independent properties of the path."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    H, W, D, M, grid = 480, 640, 64, 384, (128, 128, 128)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(D, M, grid)
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 130000)
    depths = list(fp.forward_pass(scene, (0, 2, 1)))
    ctx = fp._ctx
    acc = fp.accumulator
    assert torch.isfinite(acc).all()
    for r in (0, 1):
        rvc = fp.voxel_count[r]
        msgs = fp.messages[r]
        assert int(rvc.min()) >= 0 and int(rvc.max()) <= M
        assert torch.isfinite(msgs).all()
        # nothing is written beyond a ray's count
        idx = torch.arange(M, device="cuda")[None, :]
        assert float(msgs[idx >= rvc[:, None]].abs().max()) == 0.0
        d = depths[r]
        assert d.shape == (H, W) and np.isfinite(d).all()
    # planted sphere of radius 0.5 centred at (0,0,-0.1): the centre
    # pixel's depth is the distance to the sphere's front, within a couple of voxels
    c0 = np.linalg.norm(scene.get_image(0).camera.center.ravel()[:3])
    centre = depths[0][H // 2 - 4:H // 2 + 4, W // 2 - 4:W // 2 + 4]
    assert np.abs(np.median(centre) - (np.linalg.norm(scene.get_image(0).camera.center.ravel()[:3] - np.array([0, 0, -0.1])) - 0.5)) < 0.06
    # voxel lists: consecutive voxels differ by one step along exactly one axis
    st_vox = None
    # re-run the prefix for a slice through the C ABI to inspect the packed lists
    n = 4096
    ridx = torch.arange(100000, 100000 + n, dtype=torch.int32, device="cuda")
    views = scene.view_indices_with_neighbors(0, 4)
    P = ctx.dev(np.array([scene.get_image(v).camera.P for v in views], np.float32))
    Pi = ctx.dev(scene.get_image(0).camera.P_pinv.astype(np.float32))
    cc = ctx.dev(scene.get_image(0).camera.center.ravel().astype(np.float32))
    vox = torch.zeros((n, M), dtype=torch.int32, device="cuda")
    rvc = torch.zeros((n,), dtype=torch.int32, device="cuda")
    Sr = torch.zeros((n, M), device="cuda")
    ctx.scene_prepare(ridx, [bank.view_features(scene, v) for v in views], P, Pi, cc, vox, rvc, Sr)
    v = vox.cpu().numpy().astype(np.int64)
    xyz = np.stack([v >> 20, (v >> 10) & 1023, v & 1023], -1)
    cnt = rvc.cpu().numpy()
    step = np.abs(np.diff(xyz, axis=1)).sum(-1)
    valid = (np.arange(M - 1)[None, :] + 1) < cnt[:, None]
    assert np.all(step[valid] == 1)
    assert np.all((xyz[valid.nonzero()[0], valid.nonzero()[1]] < 128))
    # resident columns are probability distributions
    s = Sr.cpu().numpy()
    has = cnt >= 1
    assert np.abs(s.sum(1)[has] - 1).max() < 1e-4 and (s >= 0).all()
    # determinism up to float-atomic ordering: a second run agrees closely
    fp2 = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 130000)
    depths2 = list(fp2.forward_pass(scene, (0, 2, 1)))
    assert float((fp2.accumulator - acc).abs().max()) < 1e-2
    assert (np.abs(depths2[0] - depths[0]) > 1e-4).mean() < 1e-3


def _config1_scene(torch, device):
    """BASELINE.json configs[0]: Restrepo mock scene_1 cameras + bbox, 2 views, 16 planes,
    32^3 voxels, on a 64x36 crop-scaled image."""
    import os
    from conftest import GOLDEN
    from raynet_amd.common.scene import restrepo_cameras_scene
    from raynet_amd.synthetic import FeatureBank
    H, W = 36, 64
    scene = restrepo_cameras_scene(os.path.join(GOLDEN, "restrepo_mock_scene_1"), (H, W),
                                   scale=W / 1280.0)
    g = torch.Generator(device="cpu").manual_seed(7)
    bank = FeatureBank([(torch.randn((H + 12, W + 12, 32), generator=g) * 0.25).to(device)
                        for _ in range(scene.n_images)])
    return scene, bank, H, W


def test_config1_restrepo_cameras(torch, oracle_mod):
    scene, bank, H, W = _config1_scene(torch, "cuda")
    assert scene.n_images == 12
    assert np.allclose(scene.bbox.ravel(), [-5, -5, -0.7, 5, 5, 1.5])
    from raynet_amd.forward_pass import get_forward_pass_factory
    gp = _gp(16, 96, (32, 32, 32), neighbors=1)
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
    depths = list(fp.forward_pass(scene, (0, 3, 1)))
    o = oracle_mod.Oracle(M=96, D=16, N=2, F=32, H=H, W=W, padding=11, bbox=scene.bbox.ravel(),
                          grid_shape=(32, 32, 32), threads=oracle_mod.Oracle.max_threads())
    vg = oracle_mod.voxel_grid_centers(scene.bbox.ravel(), (32, 32, 32))
    ridx = np.arange(H * W, dtype=np.int32)
    cams = {}
    for r in range(3):
        views = scene.view_indices_with_neighbors(r, 1)
        cams[r] = (bank.stacked(views).cpu().numpy(),
                   np.array([scene.get_image(v).camera.P for v in views], np.float32),
                   scene.get_image(r).camera.P_pinv.astype(np.float32),
                   scene.get_image(r).camera.center.ravel().astype(np.float32))
    oracle_mod.Oracle.set_robust_messages(True)      # (see test_resident_vs_reference_schedule_vs_oracle)
    try:
        acc = o.prior(0.05)
        msgs = {r: np.zeros((H * W, 96), np.float32) for r in range(3)}
        for it in range(3):
            out = o.prior(0.05)
            for r in range(3):
                f, P, Pi, c = cams[r]
                rvi, rvc, _ = o.fused_bp(ridx, f, P, Pi, c, vg, acc, msgs[r], out)
            acc = out
    finally:
        oracle_mod.Oracle.set_robust_messages(False)
    assert rvc.max() > 10      # the aerial cameras do see the box
    assert np.abs(fp.accumulator.cpu().numpy() - acc).max() <= 2e-4
    for r in range(3):
        f, P, Pi, c = cams[r]
        _, _, S_new, depth = o.fused_depth(ridx, f, P, Pi, c, vg, acc, msgs[r])
        assert _depth_close(depths[r], depth.reshape(W, H).T, S_new, W, H) <= 0.02


def _rank_main(rank, world, port, out_dir, deterministic=False):
    import os
    import sys
    import torch
    import torch.distributed as dist
    from conftest import REPO
    sys.path.insert(0, REPO)
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                                world_size=world)
    H, W = 48, 64
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    from raynet_amd.forward_pass import map_owner
    fp = get_forward_pass_factory("raynet")(bank, _gp(32, 192, (64, 64, 64)), "sample_in_bbox",
                                            (H, W), 0, deterministic=deterministic)
    depths = list(fp.forward_pass(scene, (0, 5, 1)))
    if world > 1:
        # image k's map is handed out by its owner only; the others get None for it
        owned = np.array([map_owner(k, 5, world) == rank for k in range(5)])
        assert [d is not None for d in depths] == owned.tolist()
        depths = [d if d is not None else np.zeros((H, W), np.float32) for d in depths]
    else:
        owned = np.ones(5, bool)
    np.savez(os.path.join(out_dir, "%sw%d_r%d.npz" % ("d" if deterministic else "", world, rank)),
             depth=np.stack(depths), owned=owned, acc=fp.accumulator.cpu().numpy(),
             rows=np.array([len(fp.ray_index[r]) for r in range(5)]),
             balance=np.array(fp.shard_balance if fp.shard_balance is not None else []),
             alpha=np.float64(fp.shard_alpha if fp.shard_alpha is not None else 0.0))
    if world > 1:
        dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_ranks_on_real_kernels(torch, tmp_path, world):
    """2 / 4 / 8 ranks (gloo, all on cuda:0 -- RCCL needs one GPU per rank) run the real HIP
    kernels on their voxel-balanced ray shards; the merged accumulator and depth maps equal
    the single-rank run (prior counted once) and are the same on every rank."""
    import torch.multiprocessing as mp
    out = str(tmp_path)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(0, 1, 0, out))]
    procs[0].start()
    procs[0].join(300)
    assert procs[0].exitcode == 0
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, out, False))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    one = np.load(out + "/w1_r0.npz")
    ranks = [np.load(out + "/w%d_r%d.npz" % (world, q)) for q in range(world)]
    r0 = ranks[0]
    for rq in ranks[1:]:
        assert np.array_equal(r0["acc"], rq["acc"])
        assert np.array_equal(r0["balance"], rq["balance"])
    # every image's map is handed out by exactly one rank, its owner (map_owner)
    owned = np.stack([rq["owned"] for rq in ranks])                         # [world, images]
    assert np.all(owned.sum(0) == 1)
    assert owned.sum(1).max() == -(-5 // world)                             # dealt out evenly
    merged = np.zeros_like(one["depth"])
    for q, rq in enumerate(ranks):
        for k in np.where(rq["owned"])[0]:
            merged[k] = rq["depth"][k]
    assert np.abs(one["acc"] - r0["acc"]).max() < 5e-4
    assert (np.abs(one["depth"] - merged) > 1e-4).mean() < 0.01
    # every ray owned once; what the cuts equalise is a rank's WEIGHT -- its traversed voxels
    # plus alpha x the mean count for every ray (the plane sweep costs the same for every ray;
    # alpha from the shape, options.shard_alpha_for: 0.16 N D / mean count) -- to 10 % here (cuts
    # on 64-row boundaries of 3072-row images; tools/shard_proxy.py shows config-2 size)
    rows = np.stack([rq["rows"] for rq in ranks]).astype(np.float64)       # [world, images]
    assert np.all(rows.sum(0) == 48 * 64)
    vox = r0["balance"].astype(np.float64)                                   # [images, world]
    assert vox.shape == (5, world)
    alpha = float(r0["alpha"])
    from raynet_amd.hip_implementations.options import shard_alpha_for
    assert abs(alpha - shard_alpha_for(5, 32, vox.sum() / (5 * 48 * 64))) < 1e-9
    weight = (vox + alpha * vox.sum(1, keepdims=True) / (48 * 64) * rows.T).sum(0)
    # (8 ranks share 3072 rows in units of 64: a cut can be off by 32 rows of a 384-row shard)
    assert np.all(np.abs(weight / weight.mean() - 1) < (0.22 if world == 8 else 0.10)), weight


def test_plan_path_equals_the_launch_by_launch_path(torch):
    """One C call per phase of a pass, no combine kernel (the prior added where an accumulator
    is read), maps into plan-owned host buffers -- against the same pass launch by launch:
    the same bits in fixed-point mode for 0 / 1 / 2 / 3 BP iterations, the usual float
    tolerance otherwise; a second pass over the same plan gives the first one's results."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import make_synthetic_scene
    H, W = 48, 64
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(32, 192, (64, 64, 64))
    cls = get_forward_pass_factory("raynet")
    for T in (0, 1, 2, 3):
        res = {}
        for plan_path in (True, False):
            for det in (True, False):
                fp = cls(bank, gp, "sample_in_bbox", (H, W), 0, bp_iterations=T,
                         options=PathOptions(plan_path=plan_path, deterministic=det))
                d = np.stack([m.copy() for m in fp.forward_pass(scene, (0, 5, 1))])
                assert (fp._plan["fast"] is not None) == plan_path
                acc = fp.accumulator.cpu().numpy()
                msgs = fp.messages[2].cpu().numpy()
                res[plan_path, det] = (d, acc, msgs)
                if plan_path:                     # again over the cached plan (other host slot)
                    d2 = np.stack(list(fp.forward_pass(scene, (0, 5, 1))))
                    if det:
                        assert np.array_equal(d2, d)
                        assert np.array_equal(fp.accumulator.cpu().numpy(), acc)
                    else:
                        assert (np.abs(d2 - d) > 1e-4).mean() < 0.01
        for k in range(3):                         # fixed point: not one bit apart
            assert np.array_equal(res[True, True][k], res[False, True][k]), (T, k)
        assert np.abs(res[True, False][1] - res[False, False][1]).max() <= 2e-4
        assert np.abs(res[True, False][2] - res[False, False][2]).max() <= 1e-4
        assert (np.abs(res[True, False][0] - res[False, False][0]) > 1e-4).mean() < 0.01
        assert np.abs(res[True, False][1] - res[True, True][1]).max() <= 5e-4


def test_captured_step_replays_the_eager_pass(torch):
    """Once the scatter's adaptive tile shape has settled, a plan's whole step -- K1 prefix, BP
    sweeps, scatters, depth launches, the maps' copies -- is recorded into ONE HIP graph per
    pinned host set and replayed: the same bits as the eager passes (fixed-point mode), the
    same accumulator view, and `capture=False` never captures."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import make_synthetic_scene
    H, W = 48, 64
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(32, 192, (64, 64, 64))
    cls = get_forward_pass_factory("raynet")
    for T in (3, 2):
        fp = cls(bank, gp, "sample_in_bbox", (H, W), 0, bp_iterations=T,
                 options=PathOptions(deterministic=True, capture="on", maps="lease" if T == 3 else "copy"))
        first = np.stack([m.copy() for m in fp.forward_pass(scene, (0, 5, 1))])
        acc = fp.accumulator.cpu().numpy()
        assert not fp.captured
        seen = []
        # (the adaptive scatter steps through its tile shapes on this small, coarse scene, a
        # dozen probe launches each: it settles within ~40 scatter launches)
        for i in range(40):
            d = np.stack([m.copy() for m in fp.forward_pass(scene, (0, 5, 1))])
            seen.append(fp.captured)
            assert np.array_equal(d, first), (T, i)
            assert np.array_equal(fp.accumulator.cpu().numpy(), acc)
            if sum(seen) >= 4:
                break
        # one graph per pinned host set a pass wrote: this caller drops a pass's maps before the
        # next pass, so leased maps and copies alike stay on ONE set
        assert seen[-1] and seen[-2] and len(fp._plan["graphs"]) == 1
        # another iteration count on the same plan is another step: never the recorded one's replay
        fp.bp_iterations = T - 1
        fewer = np.stack([m.copy() for m in fp.forward_pass(scene, (0, 5, 1))])
        assert not np.array_equal(fewer, first)
        fp.bp_iterations = T
        again = np.stack([m.copy() for m in fp.forward_pass(scene, (0, 5, 1))])
        assert fp.captured and np.array_equal(again, first)
        msgs = fp.messages[1].cpu().numpy()
        eager = cls(bank, gp, "sample_in_bbox", (H, W), 0, bp_iterations=T,
                    options=PathOptions(deterministic=True, capture="off"))
        for i in range(len(seen)):
            d = np.stack([m.copy() for m in eager.forward_pass(scene, (0, 5, 1))])
        assert not eager.captured and np.array_equal(d, first)
        assert np.array_equal(eager.messages[1].cpu().numpy(), msgs)
        # "auto": without a process group the step is never captured
        auto = cls(bank, gp, "sample_in_bbox", (H, W), 0, bp_iterations=T,
                   options=PathOptions(deterministic=True))
        for i in range(len(seen)):
            d = np.stack([m.copy() for m in auto.forward_pass(scene, (0, 5, 1))])
        assert not auto.captured and np.array_equal(d, first)


def test_maps_of_a_pass_are_never_overwritten_under_a_caller(torch):
    """PathOptions.maps.  "lease" (default): a pass yields arrays that ARE the pinned host memory
    the GPU wrote (no copy) and hold a lease on it until they -- and every view or slice of them --
    are garbage-collected; no later pass touches leased memory, so whatever a caller keeps stays
    what it was, like the reference's fresh arrays from `.get()` (forward_pass.py:739-744).
    Dropped maps give their set back to the plan's pool.  "copy": pageable copies.  bp_iterations
    is not part of the plan key, so passes with different T share one plan -- and differ."""
    import gc
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import make_synthetic_scene
    H, W = 48, 64
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(32, 192, (64, 64, 64))
    cls = get_forward_pass_factory("raynet")

    def run(fp, T):
        fp.bp_iterations = T
        return list(fp.forward_pass(scene, (0, 5, 1)))

    truth = {}
    fp = cls(bank, gp, "sample_in_bbox", (H, W), 0, options=PathOptions(deterministic=True, maps="copy"))
    for T in (3, 1, 2):
        truth[T] = np.stack(run(fp, T))
    assert not np.array_equal(truth[3], truth[1])
    a, b = run(fp, 3), run(fp, 1)
    assert not any(np.shares_memory(x, y) for x in a for y in b)
    assert fp._plan["sets"] == [] and fp._plan["scratch"] is not None

    assert PathOptions().maps == "lease"
    fp = cls(bank, gp, "sample_in_bbox", (H, W), 0, options=PathOptions(deterministic=True))
    for T in (3, 1, 2, 3):                  # a caller that lets go of a pass's maps: ONE set
        got = np.stack(run(fp, T))          # (np.stack copies; the leases end here)
        assert np.array_equal(got, truth[T])
    pool = fp._plan["sets"]
    assert len(pool) == 1 and len(pool[0]["live"]) == 0 and fp._plan["scratch"] is None
    a = run(fp, 3)
    base = pool[0]["host"]
    lo, hi = base.data_ptr(), base.data_ptr() + base.numel() * 4
    assert all(lo <= x.ctypes.data < hi for x in a)          # zero-copy: the pinned memory itself
    keep = a[2][5:9, 7:11]                  # a slice of one map, kept; the rest dropped
    del a
    gc.collect()
    assert len(pool[0]["live"]) == 1
    b = run(fp, 1)                          # set 0 is leased: another set
    assert len(pool) == 2 and np.array_equal(np.stack(b), truth[1])
    c = run(fp, 2)                          # `keep` holds set 0, `b` set 1: a third
    assert len(pool) == 3 and np.array_equal(np.stack(c), truth[2])
    assert np.array_equal(keep, truth[3][2][5:9, 7:11]) and np.array_equal(np.stack(b), truth[1])
    del keep, b, c
    gc.collect()
    assert [len(st["live"]) for st in pool] == [0, 0, 0]
    d = run(fp, 3)                          # back on set 0, nothing new
    assert len(pool) == 3 and len(pool[0]["live"]) == 5 and np.array_equal(np.stack(d), truth[3])
    # a caller that hoards: beyond MAX_LEASED_SETS sets of pinned memory it gets pageable copies
    hoard = [d] + [run(fp, 1 + i % 3) for i in range(fp.MAX_LEASED_SETS + 1)]
    assert len(pool) == fp.MAX_LEASED_SETS and fp._plan["scratch"] is not None
    last = hoard[-1]
    assert not any(st["host"].data_ptr() <= x.ctypes.data < st["host"].data_ptr() + st["host"].numel() * 4
                   for st in pool for x in last)
    for i, maps in enumerate(hoard[1:]):
        assert np.array_equal(np.stack(maps), truth[1 + i % 3])
    assert np.array_equal(np.stack(hoard[0]), truth[3])


def test_pixel_order_maps_straight_from_the_depth_sweep(torch):
    """Without a process group the depth launches write the maps in ray-index (pixel) order
    themselves (rn_scene_plan.depth_image) and a group of images leaves in one copy: the same
    bits as the launch-by-launch path's row-order maps re-ordered behind the sweep, for image
    counts with and without the head launch, rows that do not fill their last tile (45 x 61
    rays), ray-index rows, and a sub-range of the images."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import make_synthetic_scene
    H, W = 45, 61
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(32, 192, (64, 64, 64))
    cls = get_forward_pass_factory("raynet")
    for rng, tile in (((0, 5, 1), (16, 16)), ((0, 2, 1), (16, 16)), ((1, 5, 2), None), ((0, 5, 1), None)):
        out = {}
        for direct in (True, False):
            fp = cls(bank, gp, "sample_in_bbox", (H, W), 0,
                     options=PathOptions(deterministic=True, plan_path=direct, ray_tile=tile))
            out[direct] = np.stack([m.copy() for m in fp.forward_pass(scene, rng)])
            assert (fp._plan["fast"] is not None) == direct and fp._plan["direct"] == direct
            again = np.stack(list(fp.forward_pass(scene, rng)))
            assert np.array_equal(again, out[direct])
        assert out[True].shape == (len(range(*rng)), H, W)
        assert np.array_equal(out[True], out[False]), (rng, tile)
        assert out[True].min() > 0


def test_plan_is_keyed_on_geometry_and_refreshed_for_moved_features(torch):
    """The plan of a pass is reused while cameras, neighbour selection, image range, shapes and
    options stay the same; feature maps that moved (recomputed into new allocations) only refresh
    the pointer table; a replaced camera, another image range or another option builds a new plan
    (a stale camera table would silently cast the old rays)."""
    from raynet_amd.common.camera import Camera
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import FeatureBank, make_synthetic_scene
    H, W = 32, 48
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(16, 96, (32, 32, 32))
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0,
                                            options=PathOptions(deterministic=True))
    d0 = np.stack([m.copy() for m in fp.forward_pass(scene, (0, 5, 1))])
    plan = fp._plan
    vox_ptr = plan["vox"].data_ptr()
    # the same maps at new addresses: same plan object, same buffers, refreshed table, same bits
    moved = FeatureBank([bank.view_features(scene, v).clone() for v in range(5)])
    fp._model = moved
    d1 = np.stack([m.copy() for m in fp.forward_pass(scene, (0, 5, 1))])
    assert fp._plan is plan and plan["vox"].data_ptr() == vox_ptr
    assert plan["table"].cpu().numpy().ravel()[0] == moved.view_features(scene, 0).data_ptr()
    assert np.array_equal(d0, d1)
    # other CONTENT at those addresses is picked up (nothing of a pass's results is cached)
    moved.view_features(scene, 1).mul_(-1.0)
    d2 = np.stack([m.copy() for m in fp.forward_pass(scene, (0, 5, 1))])
    assert fp._plan is plan and not np.array_equal(d1, d2)
    moved.view_features(scene, 1).mul_(-1.0)
    # a camera replaced in place (new object): new plan, other depths for that image
    old = scene.get_image(0).camera
    K2 = np.array(old.K, np.float64).copy()
    K2[0, 0] *= 1.1
    K2[1, 1] *= 1.1
    scene.get_image(0).camera = Camera(K2, old.R, old.t)
    d3 = np.stack([m.copy() for m in fp.forward_pass(scene, (0, 5, 1))])
    assert fp._plan is not plan and not np.array_equal(d3[0], d0[0])
    scene.get_image(0).camera = old
    d4 = np.stack([m.copy() for m in fp.forward_pass(scene, (0, 5, 1))])
    assert np.array_equal(d4, d0)
    # another range / another option: new plans
    plan = fp._plan
    list(fp.forward_pass(scene, (0, 3, 1)))
    assert fp._plan is not plan
    plan = fp._plan
    fp.ray_tile = None
    list(fp.forward_pass(scene, (0, 3, 1)))
    assert fp._plan is not plan and fp._plan["patch_rows"] is False


def test_ranks_without_rays_take_part_in_the_exchange(torch, monkeypatch):
    """More ranks than a tiny image has rows to give: a rank that owns NO rays still runs every
    phase of the plan (its partial sums are cleared, not left as they were) and every collective.
    The exchange is stubbed in-process (all-reduce: nothing; all-gather: own rows everywhere), so
    only the code path is under test, not the merged result."""
    import types
    import raynet_amd.forward_pass as F
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import make_synthetic_scene

    class Stub(object):
        ReduceOp = types.SimpleNamespace(SUM=0, MIN=1, MAX=2)

        def __init__(self, world):
            self.world, self.calls = world, []

        def all_reduce(self, t, op=None):
            self.calls.append("all_reduce")

        def all_gather_into_tensor(self, out, inp):
            self.calls.append("all_gather")
            out.view(self.world, -1).copy_(inp.view(1, -1).expand(self.world, -1))

    world = 8
    empty = 0
    # (2 x 2 rays: the plan path; 2 x 3: H * W is no multiple of 4, rn_stitch_rows cannot put the
    # owners' maps together with 16-byte stores and the pass runs launch by launch)
    for (H, W), det in (((2, 2), False), ((2, 2), True), ((2, 3), True)):
        scene, bank = make_synthetic_scene(H=H, W=W, n_views=3, focal=1.5 * H)
        for rank in range(world):
            stub = Stub(world)
            monkeypatch.setattr(F, "_dist", lambda s=stub, r=rank: (s, r, world))
            fp = F.get_forward_pass_factory("raynet")(
                bank, _gp(8, 48, (16, 16, 16), neighbors=2), "sample_in_bbox", (H, W), 0,
                options=PathOptions(shard="rays", deterministic=det))
            maps = list(fp.forward_pass(scene, (0, 3, 1)))
            assert (fp._plan["fast"] is not None) == ((H * W) % 4 == 0)
            assert len(maps) == 3
            for k in range(3):      # owner-only maps: image k from rank k * world // images only
                assert (maps[k] is not None) == (F.map_owner(k, 3, world) == rank)
                if maps[k] is not None:
                    assert maps[k].shape == (H, W) and np.isfinite(maps[k]).all()
            calls = list(stub.calls)
            n = len(fp.ray_index[0])
            empty += n == 0
            # three exchanges of the sums (+ the agreement on the path), ONE all-gather of the rows
            assert calls.count("all_gather") == 1 and calls.count("all_reduce") >= 3
            if n == 0 and fp._plan["fast"] is not None:
                # nothing of an earlier pass survives in a rank's partial sums: all zero
                acc = fp._acc_flat if not det else None
                if acc is not None:
                    assert float(acc.abs().max()) == 0.0
    assert empty >= 2 * 4 + 2           # 4 / 4 / 6 rays, 8 ranks


def test_unaligned_slices_are_accepted(torch):
    """Per-ray arrays are read element by element: a ray batch of 50, an odd row offset, an M
    that is no multiple of 4 (rows then start 4-byte aligned only) all run -- the reference
    takes any slice (raynet_fp.py:291-301 checks shapes and dtypes, nothing else)."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import make_synthetic_scene
    H, W = 20, 30
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=3, focal=1.5 * H)
    cls = get_forward_pass_factory("raynet")
    out = {}
    for M in (96, 98):
        gp = _gp(16, M, (32, 32, 32), neighbors=2)
        for batch in (0, 50, 300):
            fp = cls(bank, gp, "sample_in_bbox", (H, W), batch,
                     options=PathOptions(deterministic=True, ray_tile=None))
            out[M, batch] = np.stack(list(fp.forward_pass(scene, (0, 3, 1))))
            assert np.isfinite(out[M, batch]).all()
        assert np.array_equal(out[M, 0], out[M, 50]) and np.array_equal(out[M, 0], out[M, 300])
    assert (np.abs(out[96, 0] - out[98, 0]) > 1e-4).mean() < 0.01      # M only pads the rows
    assert fp._rows_M() == 112      # (no list can reach 98 in a 32^3 grid: rows of 112, _row_stride)
    # ... and rows of exactly 98 (a grid whose lists CAN reach M keeps it: 36 + 36 + 36 - 2 = 106)
    gp = _gp(16, 98, (36, 36, 36), neighbors=2)
    scene, _ = make_synthetic_scene(H=H, W=W, n_views=3, focal=1.5 * H)     # (a scene caches its grid)
    for batch in (0, 50):
        fp = cls(bank, gp, "sample_in_bbox", (H, W), batch,
                 options=PathOptions(deterministic=True, ray_tile=None))
        out[36, batch] = np.stack(list(fp.forward_pass(scene, (0, 3, 1))))
    assert fp._rows_M() == 98 and np.isfinite(out[36, 0]).all()
    assert np.array_equal(out[36, 0], out[36, 50])
    # the entry point itself on slices that start at odd elements
    ctx = fp._ctx
    n = 37
    ridx = torch.arange(1, n + 1, dtype=torch.int32, device="cuda")[1:]
    rvc = torch.zeros(n + 3, dtype=torch.int32, device="cuda")[3:3 + len(ridx)]
    vox = torch.zeros((n + 1, 98), dtype=torch.int32, device="cuda")[1:1 + len(ridx)]
    Sr = torch.zeros((n + 1, 98), dtype=torch.float32, device="cuda")[1:1 + len(ridx)]
    views = scene.view_indices_with_neighbors(0, 2)
    cam = fp._plan["cam_dev"]
    ctx.scene_prepare(ridx, [bank.view_features(scene, v) for v in views], cam[0, :36], cam[0, 36:48],
                      cam[0, 48:], vox, rvc, Sr)
    torch.cuda.synchronize()
    assert int(rvc.max()) > 1 and ridx.data_ptr() % 16 != 0 and vox.data_ptr() % 16 != 0


def test_a_new_scene_reuses_the_old_plans_memory(torch):
    """A caller looping over scenes (scripts/forward_pass.py:120-142): every new scene gets a new
    plan, and the old plan's buffers must go back to the allocator BEFORE the new ones are taken --
    round 6 found every plan ever built still allocated (a ctypes byref kept on the plan struct: a
    cycle the collector does not see; 6.6 GB per scene at config 2).  Allocated memory after the
    fifth scene is what it was after the first."""
    from raynet_amd.common.scene import Scene
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import _FeatureOnlyImage, make_synthetic_scene, ring_cameras
    H, W, V = 120, 160, 5
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H)
    gp = _gp(64, 384, (128, 128, 128))
    from raynet_amd.hip_implementations.options import PathOptions
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0,
                                            options=PathOptions(deterministic=True))
    allocated, maps = [], []
    for i in range(5):
        sc = Scene([_FeatureOnlyImage(H, W, c) for c in ring_cameras(V, H, W, focal=1.5 * H)], scene.bbox)
        for _ in range(3):                   # (the second pass binds the scatter's work list)
            out = list(fp.forward_pass(sc, (0, V, 1)))
        maps.append(np.stack(out))
        del out
        torch.cuda.synchronize()
        allocated.append(torch.cuda.memory_allocated())
    plan_bytes = fp._plan["bytes"]
    assert plan_bytes > 100e6
    assert max(allocated) - allocated[0] < 0.05 * plan_bytes, (allocated, plan_bytes)
    for m in maps[1:]:
        assert np.array_equal(m, maps[0])    # (the same cameras, fixed-point sums: the same maps)


def _nccl_single_main(port, out_dir):
    import os
    import sys
    import torch
    import torch.distributed as dist
    from conftest import REPO
    sys.path.insert(0, REPO)
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    H, W = 48, 64
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    from raynet_amd.hip_implementations.options import PathOptions
    res = {}
    for tag, opt in (("f", PathOptions(capture="off")), ("d", PathOptions(deterministic=True, capture="off")),
                     ("rs", PathOptions(deterministic=True, exchange="reduce_scatter", capture="off")),
                     ("cap", PathOptions(deterministic=True)),
                     ("capf", PathOptions()),
                     ("g", PathOptions(plan_path=False))):
        fp = get_forward_pass_factory("raynet")(bank, _gp(32, 192, (64, 64, 64)), "sample_in_bbox",
                                                (H, W), 0, options=opt)
        depths = [m.copy() for m in fp.forward_pass(scene, (0, 5, 1))]
        assert (fp._plan["fast"] is not None) == opt.plan_path
        if tag == "rs":
            assert "slab_i" in fp._plan       # the reduce-scatter / all-gather pair did run
        if opt.plan_path:
            assert fp._plan["rows"]["mine"] == [0, 1, 2, 3, 4]       # owner-only epilogue, one owner
        if tag.startswith("cap"):
            # the step as ONE captured graph, RCCL's collectives in it: once the scatter's
            # tile shape has settled every pass is a replay -- with the first pass's bits
            for _ in range(40):
                again = [m.copy() for m in fp.forward_pass(scene, (0, 5, 1))]
                if fp.captured:
                    break
            again = [m.copy() for m in fp.forward_pass(scene, (0, 5, 1))]
            assert fp.captured, "the step was never captured under RCCL"
            if opt.deterministic:
                assert all(np.array_equal(a, b) for a, b in zip(again, depths))
        res[tag] = (np.stack(depths), fp.accumulator.cpu().numpy())
    np.savez(os.path.join(out_dir, "nccl.npz"), depth=res["f"][0], acc=res["f"][1],
             depth_fixed=res["d"][0], acc_fixed=res["d"][1], depth_rs=res["rs"][0],
             acc_rs=res["rs"][1], depth_granular=res["g"][0], acc_granular=res["g"][1],
             depth_cap=res["cap"][0], acc_cap=res["cap"][1], depth_capf=res["capf"][0],
             acc_capf=res["capf"][1])
    dist.destroy_process_group()


def test_rccl_code_path_in_a_one_rank_world(torch, tmp_path):
    """The collectives of the sharded path -- float all-reduce of the partial sums, int64
    all-reduce in the deterministic mode (or int64 reduce-scatter + sharded combine + float
    all-gather), all_gather_into_tensor of every image's depth rows on the side stream --
    executed by RCCL itself (backend "nccl") in a process group of ONE rank, through the plan
    path (one C call per phase) and launch by launch: what a single-GPU box can run of
    BASELINE.json config 3's exchange.  Results equal the run without a process group."""
    import torch.multiprocessing as mp
    out = str(tmp_path)
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_single_main, args=(_free_port(), out))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    for det in (False, True):
        q = ctx.Process(target=_rank_main, args=(0, 1, 0, out, det))
        q.start()
        q.join(300)
        assert q.exitcode == 0
    got = np.load(out + "/nccl.npz")
    ref, ref_d = np.load(out + "/w1_r0.npz"), np.load(out + "/dw1_r0.npz")
    assert np.abs(got["acc"] - ref["acc"]).max() < 5e-4
    assert (np.abs(got["depth"] - ref["depth"]) > 1e-4).mean() < 0.01
    assert np.array_equal(got["acc_fixed"], ref_d["acc"])          # fixed point: the same bits
    assert np.array_equal(got["depth_fixed"], ref_d["depth"])
    assert np.array_equal(got["acc_rs"], ref_d["acc"])             # ... whatever the exchange
    assert np.array_equal(got["depth_rs"], ref_d["depth"])
    assert np.array_equal(got["acc_cap"], ref_d["acc"])            # ... eager or captured
    assert np.array_equal(got["depth_cap"], ref_d["depth"])
    assert np.abs(got["acc_capf"] - ref["acc"]).max() < 5e-4        # the float step, captured
    assert (np.abs(got["depth_capf"] - ref["depth"]) > 1e-4).mean() < 0.01
    assert np.abs(got["acc_granular"] - ref["acc"]).max() < 5e-4
    assert (np.abs(got["depth_granular"] - ref["depth"]) > 1e-4).mean() < 0.01


def test_resident_schedule_in_memory_bounded_groups(torch, monkeypatch):
    """When the per-ray columns of all reference images do not fit the HBM budget, the
    resident schedule keeps the messages and recomputes lists + columns group by group in
    every sweep (what the reference does for everything): same results as all-resident."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    H, W = 48, 64
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(32, 192, (64, 64, 64))
    runs = []
    for budget in (None, "0.0295", "0.0255"):      # room for 2 / 1 images' columns
        if budget is None:
            monkeypatch.delenv("RAYNET_RESIDENT_GB", raising=False)
        else:
            monkeypatch.setenv("RAYNET_RESIDENT_GB", budget)
        fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0,
                                                deterministic=True)
        depths = np.stack(list(fp.forward_pass(scene, (0, 5, 1))))
        runs.append((depths, fp.accumulator.cpu().numpy(), [len(g) for g in fp._plan["groups"]]))
    assert runs[0][2] == [5] and runs[1][2] == [2, 2, 1] and runs[2][2] == [1] * 5
    for d, acc, _ in runs[1:]:                      # fixed-point sums: the same bits
        assert np.array_equal(acc, runs[0][1]) and np.array_equal(d, runs[0][0])
    monkeypatch.setenv("RAYNET_RESIDENT_GB", "0.001")
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
    with pytest.raises(MemoryError):
        list(fp.forward_pass(scene, (0, 5, 1)))


@pytest.mark.parametrize("filter_rays", [False, True])
def test_row_layout_does_not_change_results(torch, filter_rays, monkeypatch):
    """16x16-patch rows (default) vs ray-index rows: same depth maps, same accumulator up to
    the summation order of the scatter; ray_index maps rows back to rays."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    H, W, D, M, grid = 40, 56, 16, 96, (32, 32, 32)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    if filter_rays:
        rng = np.random.default_rng(0)
        masks = {i: (rng.random((H, W)) > 0.3).astype(np.float32) for i in range(5)}
        monkeypatch.setattr(type(scene), "get_depth_map", lambda self, i: masks[i], raising=False)
    cls = get_forward_pass_factory("raynet")
    # fixed-point sums: the row layout (and with it the scatter kernel and its summation order)
    # must not change a single bit
    det = {}
    for tile in ((16, 16), None, (8, 32)):
        fp = cls(bank, _gp(D, M, grid), "sample_in_bbox", (H, W), 0, filter_out_rays=filter_rays,
                 deterministic=True)
        fp.ray_tile = tile
        d = np.stack(list(fp.forward_pass(scene, (0, 3, 1))))
        det[tile] = (d, fp.accumulator.cpu().numpy())
    for tile in ((16, 16), (8, 32)):
        assert np.array_equal(det[tile][1], det[None][1]) and np.array_equal(det[tile][0], det[None][0])
    res = {}
    for tile in ((16, 16), None, (8, 32)):
        fp = cls(bank, _gp(D, M, grid), "sample_in_bbox", (H, W), 0, filter_out_rays=filter_rays)
        fp.ray_tile = tile
        depths = list(fp.forward_pass(scene, (0, 3, 1)))
        rows = fp.messages[1].cpu().numpy()
        m = np.zeros((H * W, M), np.float32)
        m[fp.ray_index[1].cpu().numpy().astype(np.int64)] = rows
        res[tile] = (np.stack(depths), fp.accumulator.cpu().numpy(), m)
    ref = res[None]
    if filter_rays:
        assert (ref[0][0][masks[0] == 0] == 0).all() and (ref[0][0][masks[0] != 0] > 0).any()
    for tile in ((16, 16), (8, 32)):
        d, acc, m = res[tile]
        assert np.abs(acc - ref[1]).max() <= 2e-4         # float atomics: the order of the sums
        assert (np.abs(d - ref[0]) > 1e-4).mean() < 0.01
        assert np.abs(m - ref[2]).max() <= 1e-4


def test_deterministic_mode_is_bit_identical_across_runs_and_ranks(torch, tmp_path):
    """with 64-bit fixed-point sums (scatter, accumulator, all-reduce) the
    accumulator and the depth maps are the SAME BITS run to run, and for 1 and 2 ranks; and
    they agree with the default (float-atomic) mode to its usual tolerance."""
    import socket
    import torch.multiprocessing as mp
    out = str(tmp_path)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    runs = []
    for k in range(2):                                  # two single-rank runs
        p = ctx.Process(target=_rank_main, args=(0, 1, 0, out, True))
        p.start()
        p.join(300)
        assert p.exitcode == 0
        d = np.load(out + "/dw1_r0.npz")
        runs.append((d["acc"].copy(), d["depth"].copy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out, True)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    r0, r1 = np.load(out + "/dw2_r0.npz"), np.load(out + "/dw2_r1.npz")
    assert np.array_equal(r0["acc"], r1["acc"])
    assert np.array_equal(r0["owned"] ^ r1["owned"], np.ones(5, bool))      # each map from ONE rank
    depth = np.where(r0["owned"][:, None, None], r0["depth"], r1["depth"])
    assert np.array_equal(r0["acc"], runs[0][0])        # 2 ranks == 1 rank, bit for bit
    assert np.array_equal(depth, runs[0][1])
    p = ctx.Process(target=_rank_main, args=(0, 1, 0, out, False))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    ref = np.load(out + "/w1_r0.npz")
    assert np.abs(ref["acc"] - runs[0][0]).max() < 5e-4
    assert (np.abs(ref["depth"] - runs[0][1]) > 1e-4).mean() < 0.01


@pytest.mark.parametrize("H,W,V,nb,tile", [(37, 53, 5, 4, (16, 16)), (37, 53, 3, 2, None),
                                            (50, 70, 4, 3, (16, 16)), (16, 16, 2, 1, (16, 16)),
                                            (33, 17, 5, 4, (8, 32))])
def test_ragged_image_sizes_view_counts_and_tiles(torch, H, W, V, nb, tile):
    """Image sizes that are no multiple of the patch, 2..5 views, odd neighbour counts, other
    patch shapes: finite maps of the right shape, and the fixed-point mode agrees with the
    float-atomic one."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H)
    gp = _gp(16, 96, (32, 32, 32), neighbors=nb)
    outs = []
    for det in (False, True):
        fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0,
                                                deterministic=det)
        fp.ray_tile = tile
        d = list(fp.forward_pass(scene, (0, V, 1)))
        assert len(d) == V and d[0].shape == (H, W) and np.isfinite(np.stack(d)).all()
        assert (np.stack(d) > 0).all()
        outs.append((np.stack(d), fp.accumulator.cpu().numpy()))
    assert np.abs(outs[0][1] - outs[1][1]).max() < 1e-3
    assert (np.abs(outs[0][0] - outs[1][0]) > 1e-4).mean() < 0.01


def test_full_size_parity_with_the_oracle(torch, oracle_mod):
    """BASELINE.json config 2 in full (5 reference images of 480x640 rays, 64 planes, 128^3,
    M=384), per pixel: the HIP path against the C oracle run with the same schedule on the
    host's cores.  The oracle uses its robust message form: the literal
    reference sequence overflows to +inf in a few voxels at this size, which is asserted too.

    Held for the FIRST pass of a fresh driver (what the reference's caller makes:
    scripts/forward_pass.py:120-142), for the THIRD pass over the cached plan -- the steady state
    bench.py times: scatter work list bound, the scatter's tile shape settled -- and for a
    replay of the step as a captured HIP graph (`capture="on"`)."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import make_synthetic_scene
    H, W, D, M, grid = 480, 640, 64, 384, (128, 128, 128)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(D, M, grid)
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0,
                                            options=PathOptions(capture="off"))
    o = oracle_mod.Oracle(M=M, D=D, N=5, F=32, H=H, W=W, padding=11, bbox=scene.bbox.ravel(),
                          grid_shape=grid, threads=oracle_mod.Oracle.max_threads())
    vg = oracle_mod.voxel_grid_centers(scene.bbox.ravel(), grid)
    ridx = np.arange(H * W, dtype=np.int32)
    cams = {}
    for r in range(5):
        views = scene.view_indices_with_neighbors(r, 4)
        cams[r] = (bank.stacked(views).cpu().numpy(),
                   np.array([scene.get_image(v).camera.P for v in views], np.float32),
                   scene.get_image(r).camera.P_pinv.astype(np.float32),
                   scene.get_image(r).camera.center.ravel().astype(np.float32))

    def run_oracle():
        acc = o.prior(0.05)
        msgs = {r: np.zeros((H * W, M), np.float32) for r in range(5)}
        for it in range(3):
            out = o.prior(0.05)
            for r in range(5):
                f, P, Pi, c = cams[r]
                o.fused_bp(ridx, f, P, Pi, c, vg, acc, msgs[r], out)
            acc = out
        return acc, msgs

    def check(what, depth_hip, acc_hip):
        """this pass's maps, accumulator and (fp.messages) messages against the oracle's"""
        assert np.abs(acc_hip - acc).max() < 2e-5 * np.abs(acc).max(), what
        inherited = differing = 0
        for r in range(5):
            f, P, Pi, c = cams[r]
            depth, gap = depth_o[r], gap_o[r]
            d = np.abs(depth - depth_hip[r].T.ravel())
            # north star: depth maps within 1e-4 of the reference.  A pixel may differ only
            #  (a) at an arg-max near-tie of the reference (two best probabilities within 5e-5:
            #      a last-bit change picks the neighbouring voxel), or
            #  (b) where the difference is INHERITED from the accumulator: hundreds of messages
            #      of either sign are summed per voxel in a different order (atomics here, the
            #      oracle's loop there), their fp32 rounding (asserted above: 2e-5 of the
            #      largest value) moves sigma(acc - msg) of a voxel whose sum nearly cancels.
            #      Proof per pixel: the ORACLE's own K2 arithmetic on this run's accumulator
            #      and this ray's messages picks the voxel the HIP path picked.
            rows = None
            for idx in np.where(d > 1e-4)[0]:
                if gap[idx] <= 5e-5:
                    continue
                if rows is None:
                    rows = {int(q): k for k, q in enumerate(fp.ray_index[r].cpu().numpy())}
                m_hip = fp.messages[r][rows[int(idx)]].cpu().numpy()[None]
                rvi_1, rvc_1, Sv_o = o.fused_bp(ridx[idx:idx + 1], f, P, Pi, c, vg, acc,
                                                msgs[r][idx:idx + 1].copy(), o.prior(0.05))
                again = o.depth_distribution(Sv_o, rvi_1, rvc_1, acc_hip, m_hip)
                d2 = o.depth_from_distribution(again, rvi_1, vg, c)
                assert abs(float(d2[0]) - float(depth_hip[r].T.ravel()[idx])) <= 1e-4, (what, r, idx)
                inherited += 1
            differing += int((d > 1e-4).sum())
        # observed: 2 - 5 of the scene's 1,536,000 pixels (profiles/r04_a_fullsize_parity.json),
        # each a near-tie or inherited as proven above
        assert differing <= 8, (what, differing)
        assert inherited <= 4, (what, inherited)

    try:
        oracle_mod.Oracle.set_robust_messages(True)
        acc, msgs = run_oracle()
        assert np.isfinite(acc).all()
        depth_o, gap_o = {}, {}
        for r in range(5):
            f, P, Pi, c = cams[r]
            _, _, S_new, depth_o[r] = o.fused_depth(ridx, f, P, Pi, c, vg, acc, msgs[r])
            top = np.partition(S_new, M - 2, axis=1)[:, M - 2:]
            gap_o[r] = np.abs(top[:, 1] - top[:, 0])
            del S_new
        first = np.stack(list(fp.forward_pass(scene, (0, 5, 1))))
        check("first pass", first, fp.accumulator.cpu().numpy())
        list(fp.forward_pass(scene, (0, 5, 1)))
        third = np.stack(list(fp.forward_pass(scene, (0, 5, 1))))
        assert fp._plan["passes"] == 3 and fp._plan.get("items") is not None       # work list bound
        check("third pass", third, fp.accumulator.cpu().numpy())
        fp.options = fp.options.replace(capture="on")       # (not part of a plan's key: same plan)
        for _ in range(12):     # (the scatter's probe launches end within a few passes)
            replay = np.stack(list(fp.forward_pass(scene, (0, 5, 1))))
            if fp.captured:
                break
        assert fp.captured and fp._ctx.scatter_settled()
        replay = np.stack(list(fp.forward_pass(scene, (0, 5, 1))))      # a replay, not the recording
        assert fp.captured
        check("captured replay", replay, fp.accumulator.cpu().numpy())
    finally:
        oracle_mod.Oracle.set_robust_messages(False)


def test_forward_pass_script_on_a_restrepo_directory(torch, tmp_path):
    """The caller of the path (scripts/forward_pass.py:29-146): a Restrepo-layout directory in,
    depth_%03d.npy files out, through the loader, the MV-CNN twin and the raynet factory."""
    from PIL import Image as PILImage
    from raynet_amd.scripts.forward_pass import main
    from raynet_amd.synthetic import ring_cameras
    from test_scene_loaders import _write_scene_info
    H, W = 24, 32
    base = tmp_path / "scene"
    (base / "imgs").mkdir(parents=True)
    (base / "cams_krt").mkdir()
    rng = np.random.default_rng(0)
    for i, cam in enumerate(ring_cameras(5, H, W, focal=1.5 * H)):
        PILImage.fromarray((rng.random((H, W, 3)) * 255).astype(np.uint8)).save(
            str(base / "imgs" / ("frame_%03d.png" % i)))
        with open(str(base / "cams_krt" / ("camera_%03d.txt" % i)), "w") as f:
            np.savetxt(f, cam.K)
            f.write("\n")
            np.savetxt(f, cam.R)
            f.write("\n")
            np.savetxt(f, cam.t.reshape(1, 3))
    _write_scene_info(str(base), [-1, -1, -1, 1, 1, 1])
    out = tmp_path / "out"
    rc = main([str(base), str(out), "--depth_planes", "16", "--grid_shape", "32,32,32",
               "--maximum_number_of_marched_voxels", "96", "--start_end", "0,3",
               "--forward_pass_factory", "raynet", "--rays_batch", "500"])
    assert rc == 0
    files = sorted(os.listdir(str(out)))
    assert files == ["depth_000.npy", "depth_001.npy", "depth_002.npy"]
    for f in files:
        d = np.load(str(out / f))
        assert d.shape == (H, W) and d.dtype == np.float32 and np.isfinite(d).all() and (d > 0).all()


def test_config4_full_size_properties_and_subset_parity(torch, oracle_mod):
    """BASELINE.json configs[3] at its FULL size on one GPU -- 9 views x 640x480 rays, 128
    depth planes, 256^3 voxels, M = 768, 3 BP iterations + depth sweep (2.76 M rays, 0.73 G
    voxel visits).  The oracle would need ~15 minutes for the coupled run, so parity is taken
    where the path is per-ray independent, on 300 random rays of every image, with the HIP
    run's own state as input:
      * K1 prefix (a1-a4): voxel lists bit-exact, clipped columns <= 2e-5;
      * iteration 3 of the BP sweep (a5): the oracle's messages from the run's accumulator
        and messages after iteration 2;
      * depth sweep (a6): the oracle's distribution from the run's final accumulator and
        messages -> the same depth except at arg-max near-ties.
    And size-independent properties of the whole run: fixed-point mode bit-identical run to
    run, float mode equal to it within the summation tolerance, message sum = accumulator sum
    (a checksum of checksums), every depth inside the camera-to-box range."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    H, W, V, D, M, grid = 480, 640, 9, 128, 768, (256, 256, 256)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H)
    gp = _gp(D, M, grid, neighbors=V - 1)
    cls = get_forward_pass_factory("raynet")

    def run(iters, det):
        fp = cls(bank, gp, "sample_in_bbox", (H, W), 0, bp_iterations=iters, deterministic=det)
        depth = np.stack(list(fp.forward_pass(scene, (0, V, 1))))
        return fp, depth

    rng = np.random.default_rng(11)
    pick = {r: np.sort(rng.choice(H * W, 300, replace=False)) for r in range(V)}

    def rows_of(fp, r):
        return torch.from_numpy(pick[r]).cuda()

    def state(fp, r):
        """(ray indices, messages, counts) of the picked ROWS of image r"""
        rows = rows_of(fp, r)
        return (fp.ray_index[r][rows].cpu().numpy(), fp.messages[r][rows].cpu().numpy(),
                fp.voxel_count[r][rows].cpu().numpy())

    fp2, _ = run(2, True)
    acc2 = fp2.accumulator.cpu().numpy()
    st2 = {r: state(fp2, r) for r in range(V)}
    del fp2
    torch.cuda.empty_cache()
    fp3, depth3 = run(3, True)
    acc3 = fp3.accumulator.cpu().numpy()
    st3 = {r: state(fp3, r) for r in range(V)}
    plan = fp3._plan
    cols = {}
    for r in range(V):
        rows = rows_of(fp3, r) + plan["per_image"][r]["row0"]
        cols[r] = (plan["Sr"][rows].cpu().numpy(), plan["vox"][rows].cpu().numpy())

    # ---- properties of the whole run
    assert np.isfinite(depth3).all() and np.isfinite(acc3).all()
    centers = np.array([scene.get_image(r).camera.center.ravel()[:3] for r in range(V)])
    far = np.linalg.norm(centers, axis=1).max() + np.sqrt(3.0)
    assert depth3.min() > 0 and depth3.max() <= far
    prior = float(np.float32(np.log(0.05) - np.log(0.95)))
    total_msgs = 0.0
    for r in range(V):                      # rows beyond a ray's count are cleared on access
        total_msgs += float(fp3.messages[r].double().sum().item())
    total_acc = float((torch.from_numpy(acc3).double() - prior).sum().item())
    assert abs(total_acc - total_msgs) <= 2e-6 * max(1.0, abs(total_msgs)) + 64.0   # G * 2^-24 * |prior|
    fp3b, depth3b = run(3, True)
    assert np.array_equal(depth3b, depth3) and np.array_equal(fp3b.accumulator.cpu().numpy(), acc3)
    del fp3b
    torch.cuda.empty_cache()
    fpf, depthf = run(3, False)
    accf = fpf.accumulator.cpu().numpy()
    assert np.abs(accf - acc3).max() <= 2e-5 * np.abs(acc3).max()
    assert (np.abs(depthf - depth3) > 1e-4).mean() < 1e-4
    del fpf

    # ---- subset parity with the oracle
    o = oracle_mod.Oracle(M=M, D=D, N=V, F=32, H=H, W=W, padding=11, bbox=scene.bbox.ravel(),
                          grid_shape=grid, threads=oracle_mod.Oracle.max_threads())
    vg = oracle_mod.voxel_grid_centers(scene.bbox.ravel(), grid)
    oracle_mod.Oracle.set_robust_messages(True)
    try:
        flips = 0
        for r in range(V):
            views = scene.view_indices_with_neighbors(r, V - 1)
            f = bank.stacked(views).cpu().numpy()
            P = np.array([scene.get_image(v).camera.P for v in views], np.float32)
            Pi = scene.get_image(r).camera.P_pinv.astype(np.float32)
            c = scene.get_image(r).camera.center.ravel().astype(np.float32)
            ridx, m2, cnt2 = st2[r]
            ridx3, m3, cnt3 = st3[r]
            assert np.array_equal(ridx, ridx3) and np.array_equal(cnt2, cnt3)
            m_o = m2.copy()
            rvi_o, rvc_o, Sv_o = o.fused_bp(ridx, f, P, Pi, c, vg, acc2, m_o, o.prior(0.05))
            Sr_h, vox_h = cols[r]
            assert np.array_equal(rvc_o, cnt3)
            packed = (rvi_o[..., 0] << 20) | (rvi_o[..., 1] << 10) | rvi_o[..., 2]
            live = np.arange(M)[None, :] < rvc_o[:, None]
            assert np.array_equal(vox_h[live], packed[live])                     # a1 + a3
            send = live & (rvc_o[:, None] > 1)
            Sc = np.where(live, np.clip(Sv_o, 1e-5, 1 - 1e-5), 0).astype(np.float32)
            Sc = Sc / np.maximum(Sc.sum(1, keepdims=True), 1e-30)
            assert np.abs(Sr_h[send] - Sc[send]).max() <= 2e-5                   # a2 + a4
            tol = 1e-5 + 64 * 2.0 ** -24 * np.exp(np.minimum(np.abs(m_o), 17.0))
            assert np.all(np.abs(m3 - m_o)[send] <= tol[send])                   # a5
            S_new_o = o.depth_distribution(Sv_o, rvi_o, rvc_o, acc3, m3)         # a6
            depth_o = o.depth_from_distribution(S_new_o, rvi_o, vg, c)
            got = depth3[r].T.ravel()[ridx]
            for i in np.where(np.abs(got - depth_o) > 1e-4)[0]:
                top = np.sort(S_new_o[i])[::-1]
                assert top[0] - top[1] <= 5e-5, (r, int(ridx[i]), top[:2])
                flips += 1
        assert flips <= 0.01 * V * 300
    finally:
        oracle_mod.Oracle.set_robust_messages(False)


def test_slab_boxes_and_second_stream_change_nothing(torch, monkeypatch):
    """The scatter's bounding boxes taken from the traversal's slab boxes
    (rn_scene_bind_slab_boxes) instead of from the lists, and the two-stream launchers
    (RAYNET_HIP_OVERLAP=1): schedule and bookkeeping only -- in fixed-point mode the
    accumulator and the depth maps are the same bits."""
    import raynet_amd.hip_implementations as hi
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    H, W = 112, 128         # 5 x 14336 rows: above the launchers' 65536-row split threshold
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(32, 192, (64, 64, 64))
    runs = {}
    for slab, overlap in (("1", "0"), ("0", "0"), ("1", "1"), ("0", "1")):
        monkeypatch.setenv("RAYNET_SLAB_BOXES", slab)
        monkeypatch.setenv("RAYNET_HIP_OVERLAP", overlap)
        hi._CONTEXTS.clear()            # RAYNET_HIP_OVERLAP is read when a context is created
        fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0,
                                                deterministic=True)
        d = np.stack(list(fp.forward_pass(scene, (0, 5, 1))))
        d2 = np.stack(list(fp.forward_pass(scene, (0, 5, 1))))          # cached plan
        assert np.array_equal(d, d2)
        runs[(slab, overlap)] = (d, fp.accumulator.cpu().numpy(), fp._ctx.scatter_state()[0])
        del fp
    hi._CONTEXTS.clear()
    ref = runs[("0", "0")]
    assert ref[2] in (0, 1)             # the LDS-box scatter is what ran
    for key, (d, acc, _) in runs.items():
        assert np.array_equal(acc, ref[1]) and np.array_equal(d, ref[0]), key


def test_edge_ranges_and_zero_iterations(torch, oracle_mod):
    """Empty image range, a single image, a strided range, and bp_iterations = 0 (the depth
    sweep then runs on the prior and zero messages: the arg-max of o T s with a constant o)."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    H, W, D, M, grid = 24, 32, 16, 96, (32, 32, 32)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = _gp(D, M, grid)
    cls = get_forward_pass_factory("raynet")
    fp = cls(bank, gp, "sample_in_bbox", (H, W), 0)
    assert list(fp.forward_pass(scene, (2, 2, 1))) == []
    one = list(fp.forward_pass(scene, (3, 4, 1)))
    assert len(one) == 1 and one[0].shape == (H, W) and np.isfinite(one[0]).all()
    strided = list(fp.forward_pass(scene, (0, 5, 2)))
    assert len(strided) == 3 and all(d.shape == (H, W) for d in strided)
    fp0 = cls(bank, gp, "sample_in_bbox", (H, W), 0, bp_iterations=0)
    d0 = list(fp0.forward_pass(scene, (0, 2, 1)))
    acc, msgs, depths, _ = _oracle_forward(oracle_mod, scene, bank, gp, [0, 1], H, W, iters=0)
    assert np.allclose(fp0.accumulator.cpu().numpy(), acc)         # the prior everywhere
    for r in (0, 1):
        assert (np.abs(d0[r] - depths[r]) > 1e-4).mean() < 0.01


@pytest.mark.parametrize("D,M,grid,nb,F,pad", [
    (2, 96, (32, 32, 32), 4, 32, 11),        # the fewest planes
    (48, 16, (32, 32, 32), 2, 32, 11),       # every ray truncated at M
    (100, 96, (30, 33, 17), 3, 32, 11),      # 2 plane chunks, grid no multiple of the bricks
    (64, 200, (64, 8, 64), 1, 32, 11),       # flat grid, 2 views
    (16, 96, (32, 32, 32), 2, 8, 5),         # F = 8: the generic sweep (incl. the folded first sweep)
    (24, 64, (32, 32, 32), 3, 20, 7),        # F no power of two, 4 views
    (32, 700, (160, 150, 170), 2, 32, 11),   # 11 list chunks: too long for the fold's LDS rows
    (16, 1024, (200, 180, 190), 1, 32, 11)]) # the longest lists the library takes
def test_resident_path_odd_shapes_vs_oracle(torch, oracle_mod, D, M, grid, nb, F, pad):
    """Plane counts that are no multiple of 64 (and the minimum, 2), lists cut off at M, grid
    sizes that are no multiple of the 4x4x4 accumulator bricks, 2 - 5 views: the resident
    schedule against the oracle's K1 / K2 schedule.  The planted surface makes these columns
    peaky and the messages large (|m| to 12), where the literal fp32 (cumsum1 - cumsum2) of
    mrf_bp.cu:157 is itself off by 1e-2 (tools/odd_shapes_probe.py); the comparator is the
    oracle's robust message form, the one pinned to the reference's NumPy path in that regime
    (tests/test_saturated_golden.py): messages and accumulator to 1e-4, depth maps equal
    except at arg-max near-ties."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    H, W, V = 24, 32, 5
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H, F=F, padding=pad)
    gp = _gp(D, M, grid, neighbors=nb, padding=pad)
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
    depths = list(fp.forward_pass(scene, (0, 3, 1)))
    assert fp._plan["fast"] is not None
    oracle_mod.Oracle.set_robust_messages(True)
    try:
        acc, msgs, depths_o, dists = _oracle_forward(oracle_mod, scene, bank, gp, [0, 1, 2], H, W,
                                                     F=F)
    finally:
        oracle_mod.Oracle.set_robust_messages(False)
    cnt = fp.voxel_count[0].cpu().numpy()
    if M == 16:
        assert (cnt == M).mean() > 0.5                 # really truncated
    assert np.isfinite(acc).all()
    assert np.abs(fp.accumulator.cpu().numpy() - acc).max() <= 1e-4
    for r in range(3):
        assert _depth_close(depths[r], depths_o[r], dists[r], W, H) <= 0.02
        m = np.zeros((H * W, M), np.float32)
        m[fp.ray_index[r].cpu().numpy().astype(np.int64)] = fp.messages[r].cpu().numpy()
        assert np.abs(m - msgs[r]).max() <= 1e-4


def test_per_launch_profiling_and_family_selection(torch):
    """rn_prof_begin / rn_prof_select / rn_prof_end: every launch of a pass is bracketed; with a
    family selected only its launches are; start offsets and durations are consistent."""
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    H, W, D, M, grid = 24, 32, 16, 96, (32, 32, 32)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    fp = get_forward_pass_factory("raynet")(bank, _gp(D, M, grid), "sample_in_bbox", (H, W), 0)
    refs = (0, 3, 1)
    ref_maps = list(fp.forward_pass(scene, refs))
    ctx = fp._ctx
    ctx.prof_begin(capacity=256)
    maps = list(fp.forward_pass(scene, refs))
    everything = ctx.prof_end()
    starts = list(ctx.prof_starts)
    names = [n for n, _, _ in everything]
    # one traversal, one plane sweep (which also writes BP iteration 0's messages: no k_bp
    # launch for it), 3 scatters, 2 more BP sweeps, the depth sweep of all images but the last
    # and of the last; no combine
    assert names.count("sweep_map") == 1 and names.count("traverse") == 1
    assert names.count("bp") == 2 and names.count("scatter") == 3 and names.count("depth") == 2
    assert names.count("acc") == 0
    assert all(ms > 0 for _, _, ms in everything) and len(starts) == len(everything)
    assert starts[0] == 0.0 and all(b >= a for a, b in zip(starts, starts[1:]))
    assert [r for n, r, _ in everything if n == "sweep_map"] == [3 * H * W]
    ctx.prof_begin(capacity=256, only=["bp"])
    list(fp.forward_pass(scene, refs))
    only_bp = ctx.prof_end()
    assert [n for n, _, _ in only_bp] == ["bp"] * 2
    ctx.prof_begin(capacity=256)                   # the selection does not stick
    list(fp.forward_pass(scene, refs))
    assert len(ctx.prof_end()) == len(everything)
    for a, b in zip(ref_maps, maps):               # profiling changes nothing but the timing
        assert np.mean(np.abs(a - b) > 1e-4) < 0.01     # (float atomics: near-ties may flip)
