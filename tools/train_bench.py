"""Training-step timing (BASELINE.json configs[4] shape: MV-CNN under PyTorch-ROCm + the HIP
MRF block forward and analytic backward), on synthetic patches: n rays x D depth hypotheses x
`views` 11x11 patches through the 5-layer MV-CNN twin, similarities + softmax, the MRF block
(3 BP sweeps + depth distribution), squared-EMD loss, backward, Adam step.
Prints the step time and the share of the HIP MRF kernels (forward + backward)."""
import os
import sys
import time

# MIOpen's exhaustive kernel search for the 11x11-patch convolutions takes >10 min on a fresh
# box; the fast find mode is what a training run would use after its first epoch anyway
os.environ.setdefault("MIOPEN_FIND_MODE", "2")

import numpy as np   # noqa: E402
import torch         # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd.hip_implementations.forward_backward_pass import forward_backward_pass   # noqa: E402
from raynet_amd.models import get_nn                                                    # noqa: E402
from raynet_amd.mrf import mrf_train                                                    # noqa: E402


def main(n=2048, D=32, M=192, views=5, grid=(64, 64, 64), steps=10):
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    bbox = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    axes = [np.linspace(bbox[i], bbox[i + 3], grid[i] + 1)[:-1] + (bbox[i + 3] - bbox[i]) / grid[i] / 2
            for i in range(3)]
    vg = np.stack(np.meshgrid(*axes, indexing="ij"), -1).astype(np.float32)
    hip = mrf_train.training_context(M, D, bbox, grid, vg)
    starts = np.c_[rng.random((n, 2)) * 1.6 - 0.8, -np.ones(n)].astype(np.float32)
    ends = np.c_[rng.random((n, 2)) * 1.0 - 0.5, np.ones(n)].astype(np.float32)
    st, en = torch.from_numpy(starts).cuda(), torch.from_numpy(ends).cuda()
    rvi = torch.zeros((n, M, 3), dtype=torch.int32, device="cuda")
    rvc = torch.zeros((n,), dtype=torch.int32, device="cuda")
    hip.voxel_traversal(st, en, rvi, rvc)
    t = torch.linspace(0, 1, D, device="cuda")[None, :, None]
    points = torch.cat([st[:, None] + t * (en - st)[:, None], torch.ones((n, D, 1), device="cuda")], -1)
    target = torch.zeros((n, M), device="cuda")
    target[torch.arange(n), (rvc // 2).long()] = 1.0
    model = get_nn("simple_cnn")().cuda()
    images = [torch.randn((n, D, 3, 11, 11), device="cuda") for _ in range(views)]
    gamma = torch.tensor(0.031, device="cuda", requires_grad=True)
    opt = torch.optim.Adam(list(model.parameters()) + [gamma], lr=1e-4)
    cams = torch.cat([st, torch.ones((n, 1), device="cuda")], 1)
    vg_d = torch.from_numpy(vg).cuda()

    def step():
        opt.zero_grad()
        loss = forward_backward_pass(model, images, vg_d, rvi, rvc, target, points, cams, hip,
                                     views=views, gamma=gamma, bp_iterations=3, loss="squared_emd")
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # the MRF block alone (forward + backward), same shapes
    S = torch.softmax(torch.randn((n, D), device="cuda"), -1).requires_grad_(True)
    for _ in range(3):
        out = mrf_train.mrf_depth_distribution(S, rvi, rvc, st, en, gamma, 3, hip)
        out.square().sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = mrf_train.mrf_depth_distribution(S, rvi, rvc, st, en, gamma, 3, hip)
        out.square().sum().backward()
    torch.cuda.synchronize()
    dm = (time.perf_counter() - t0) / steps
    print("train step: n=%d rays, D=%d, views=%d, M=%d, mean voxels/ray %.1f: %.2f ms/step "
          "(%.0f k rays/s), MRF block fwd+bwd %.2f ms (%.0f %%), loss %.4f" % (
              n, D, views, M, float(rvc.float().mean()), dt * 1e3, n / dt / 1e3, dm * 1e3,
              100 * dm / dt, float(loss.detach())))


def config5(n=1000, D=32, M=160, grid=(64, 64, 32), steps=10, H=90, W=160, views=5):
    """BASELINE.json configs[4] as stated: the batch of tests/test_config5_gpu.py -- n rays of a
    mock Restrepo camera, real sample points (K8) and voxel lists (K5), 11x11 patches of the 5
    views around the projected points, SimpleCNN twin, HIP MRF forward + analytic backward, Adam."""
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.common.scene import restrepo_cameras_scene
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.train_network.raynet_batch_provider import get_batch_of_rays
    torch.manual_seed(0)
    scene = restrepo_cameras_scene(os.path.join(REPO, "tests", "golden", "restrepo_mock_scene_1"),
                                   (H, W), n_images=views, scale=W / 1280.0)
    bbox = np.asarray(scene.bbox, np.float32).ravel()
    gp = GenerationParameters(depth_planes=D, neighbors=views - 1, grid_shape=np.array(grid, np.int32),
                              max_number_of_marched_voxels=M, padding=11, gamma_mrf=0.031)
    hip = get_context(M, D, views, 32, H, W, 11, bbox, grid)
    hip.set_voxel_grid(np.ascontiguousarray(scene.voxel_grid(gp.grid_shape).transpose(1, 2, 3, 0)))
    images = {v: torch.from_numpy(scene.get_image(v).image).permute(2, 0, 1).contiguous().cuda()
              for v in range(views)}
    rng = np.random.default_rng(3)
    # pixels of the image interior (their rays cross the box); the target point is synthetic
    # (3 n candidates: rays with a patch over an image border are redrawn, as the reference does)
    px, py = rng.integers(20, W - 20, 3 * n), rng.integers(15, H - 15, 3 * n)
    ray_idxs = (px * H + py).astype(np.int32)
    targets = rng.uniform(bbox[:3] + 0.5, bbox[3:] - 0.5, (3 * n, 3))
    t0 = time.perf_counter()
    batch = get_batch_of_rays(scene, 2, ray_idxs, gp, hip, images, targets)
    batch = [b_ if i == views else b_[:n] for i, b_ in enumerate(batch)]
    assert len(batch[views + 1]) == n, "too few candidates survive the border rule"
    torch.cuda.synchronize()
    t_batch = time.perf_counter() - t0
    patches, (vg_d, rvi, rvc, target, points, cams) = batch[:views], batch[views:]
    model = get_nn("simple_cnn")().cuda().train()
    gamma = torch.tensor(0.031, device="cuda", requires_grad=True)
    opt = torch.optim.Adam(list(model.parameters()) + [gamma], lr=1e-4)

    def step():
        opt.zero_grad()
        loss = forward_backward_pass(model, patches, vg_d, rvi, rvc, target, points, cams, hip,
                                     views=views, gamma=gamma, bp_iterations=3, loss="squared_emd")
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st, en = points[:, 0, :3].contiguous(), points[:, -1, :3].contiguous()
    S = torch.softmax(torch.randn((n, D), device="cuda"), -1).requires_grad_(True)
    for _ in range(3):
        mrf_train.mrf_depth_distribution(S, rvi, rvc, st, en, gamma, 3, hip).square().sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        mrf_train.mrf_depth_distribution(S, rvi, rvc, st, en, gamma, 3, hip).square().sum().backward()
    torch.cuda.synchronize()
    dm = (time.perf_counter() - t0) / steps
    print("config 5 (restrepo mock cameras, SimpleCNN twin): n=%d rays, D=%d, views=%d, M=%d, grid %s, "
          "mean voxels/ray %.1f: batch assembly %.1f ms (first call), train step %.2f ms (%.1f k rays/s), "
          "MRF block fwd+bwd %.2f ms (%.1f %%), loss %.4f" % (
              n, D, views, M, "x".join(map(str, grid)), float(rvc.float().mean()), t_batch * 1e3,
              dt * 1e3, n / dt / 1e3, dm * 1e3, 100 * dm / dt, float(loss.detach())))


if __name__ == "__main__":
    if os.environ.get("CONFIG5"):
        config5()
    else:
        main()
