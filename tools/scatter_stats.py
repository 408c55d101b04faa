"""Debug: per-round statistics of k_scatter_slab on the config-2 scene (needs the
-DRN_SCATTER_STATS build at tools/libraynet_hip_stats.so)."""
import ctypes, os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd import _lib
_lib.LIB_PATH = os.path.join(REPO, "tools", "libraynet_hip_stats.so")
from raynet_amd.hip_implementations import get_context
from raynet_amd.synthetic import make_synthetic_scene
from oracle import oracle
H, W, D, M, grid = 480, 640, 64, 384, (128, 128, 128)
scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
ctx = get_context(M, D, 5, 32, H, W, 11, scene.bbox.ravel(), grid)
ctx.set_voxel_grid(torch.from_numpy(oracle.voxel_grid_centers(scene.bbox.ravel(), grid)).cuda())
lib = _lib.load()
stats = (ctypes.c_ulonglong * 8)()
for r in range(5):
    views = scene.view_indices_with_neighbors(r, 4)
    P = ctx.dev(np.array([scene.get_image(v).camera.P for v in views], np.float32))
    Pi = ctx.dev(scene.get_image(r).camera.P_pinv.astype(np.float32))
    cc = ctx.dev(scene.get_image(r).camera.center.ravel().astype(np.float32))
    n = H * W
    ridx = torch.arange(n, dtype=torch.int32, device="cuda")
    vox = torch.zeros((n, M), dtype=torch.int32, device="cuda")
    rvc = torch.zeros((n,), dtype=torch.int32, device="cuda")
    Sr = torch.zeros((n, M), device="cuda")
    ctx.scene_prepare(ridx, [bank.view_features(scene, v) for v in views], P, Pi, cc, vox, rvc, Sr)
    msgs = torch.zeros((n, M), device="cuda")
    acc = torch.full((ctx.acc_size(),), -2.94, device="cuda")
    part = torch.zeros((ctx.acc_copies(), ctx.acc_size()), device="cuda")
    lib.rn_debug_scatter_stats(None, 1)
    ctx.scene_bp_sweep(Sr, vox, rvc, acc, msgs, part)
    torch.cuda.synchronize()
    lib.rn_debug_scatter_stats(stats, 0)
    rounds, lanes, tails, segs, chunks = stats[0], stats[1], stats[2], stats[3], stats[4]
    print("image %d: pairs %d chunks %d rounds/chunk %.1f lanes/round %.1f tails/round %.1f "
          "64B-segments/round %.2f total segments %.2fM" % (
              r, int(rvc.sum()), chunks, rounds / chunks, lanes / rounds, tails / rounds,
              segs / rounds, segs / 1e6))
