import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.forward_pass import get_forward_pass_factory, sweep_direction, shard_bounds
from raynet_amd.synthetic import make_synthetic_scene
H, W, V = 480, 640, 5
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H)
gp = GenerationParameters(depth_planes=64, neighbors=4, grid_shape=np.array([128]*3, np.int32),
                          max_number_of_marched_voxels=384, padding=11, gamma_mrf=0.05)
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
for _ in range(3): list(fp.forward_pass(scene, (0, V, 1)))
ctx = fp._ctx
dev = ctx.device
refs = list(range(V))
def T(label, f, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    print("%-34s %7.1f us" % (label, (time.perf_counter() - t0) / n * 1e6))
T("_view_features", lambda: fp._view_features(scene, refs))
b = fp._view_features(scene, refs)
T("bank .to(dev).contiguous()", lambda: {v: f.to(dev, torch.float32).contiguous() for v, f in b.items()})
T("_context", lambda: fp._context(scene, 32))
T("_prior", lambda: fp._prior())
G = ctx.acc_size()
T("torch.full acc_in", lambda: torch.full((G,), -2.9, dtype=torch.float32, device=dev))
T("torch.zeros acc_part", lambda: torch.zeros((ctx.acc_copies(), G), dtype=torch.float32, device=dev))
T("torch.empty acc_next", lambda: torch.empty((G,), dtype=torch.float32, device=dev))
N = 5
def cams():
    cam_host = np.zeros((len(refs), 12 * N + 16), dtype=np.float32)
    for k, r in enumerate(refs):
        vs = scene.view_indices_with_neighbors(r, gp.neighbors)
        P, P_inv, center = fp._camera_arrays([scene.get_image(v) for v in vs])
        cam_host[k, :12 * N] = P.ravel()
        cam_host[k, 12 * N:12 * N + 12] = P_inv.ravel()
        cam_host[k, 12 * N + 12:] = center
    return cam_host.tobytes()
T("camera arrays + key", cams)
T("view_indices_with_neighbors x5", lambda: [scene.view_indices_with_neighbors(r, 4) for r in refs])
T("_camera_arrays x5", lambda: [fp._camera_arrays([scene.get_image(v) for v in scene.view_indices_with_neighbors(r, 4)]) for r in refs])
T("sweep_direction", lambda: sweep_direction(H, W, [scene.get_image(v) for v in scene.view_indices_with_neighbors(0, 4)]))
npad = 307200
T("4 big torch.empty/zeros", lambda: (torch.empty((V * npad, 384), dtype=torch.int32, device=dev), torch.empty((V * npad, 384), dtype=torch.float32, device=dev), torch.empty((V * npad, 384), dtype=torch.float32, device=dev), torch.zeros((V * npad,), dtype=torch.int32, device=dev)))
