"""BASELINE.json configs[4] in its stated form: the MV-CNN twin (PyTorch-ROCm, raynet_amd/models.py)
on 11x11 patches cut around the projections of the D sample points of ~1000 real rays into the 5
mock Restrepo cameras, the HIP MRF block forward and analytic backward, a few optimiser steps.

Reference: raynet/tf_implementations/forward_backward_pass.py:128-248 (the training graph),
raynet/train_network/raynet_batch_provider.py:101-144 (a batch = the `inputs` list of n rays),
raynet/common/image.py:145-200 (patches around projected points).  The mock dataset's PNGs are
not shipped, so the five views are rendered from a textured plane inside the scene's bounding
box (the cameras, the box and the image size ratio are the dataset's); the rays, their sample
points and their voxel lists come from the path's own kernels (K8, K5)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

H, W, VIEWS, D, M = 90, 160, 5, 32, 160
GRID = (64, 64, 32)
PLANE_Z = 0.3


def _pixel_rays(cam, px, py):
    """Camera centre and unit directions of the rays through pixels (px = column, py = row):
    back-projection with P_pinv as sampling_schemes.cu:15-39 / generic_utils.py:4-29."""
    o = np.asarray(cam.P_pinv, np.float64).dot(np.stack([px, py, np.ones_like(px)]).astype(np.float64))
    c = np.asarray(cam.center, np.float64).ravel()[:3]
    d = o[:3] / o[3] - c[:, None]
    return c, d / np.linalg.norm(d, axis=0)


def _texture(x, y):
    f = [0.5 + 0.5 * np.sin(1.9 * x + 0.7 * y + 0.3) * np.cos(0.8 * y - 0.5),
         0.5 + 0.5 * np.sin(2.7 * y - 1.1 * x + 1.0),
         0.5 + 0.25 * np.cos(3.1 * x) + 0.25 * np.sin(2.3 * y + 0.6 * x)]
    return np.stack(f, -1).astype(np.float32)


def _scene():
    from raynet_amd.common.scene import restrepo_cameras_scene
    scene = restrepo_cameras_scene(os.path.join(GOLDEN, "restrepo_mock_scene_1"), (H, W),
                                   n_images=VIEWS, scale=W / 1280.0)
    images = {}
    for v in range(VIEWS):
        cam = scene.get_image(v).camera
        py, px = np.mgrid[0:H, 0:W]
        c, d = _pixel_rays(cam, px.ravel(), py.ravel())
        t = (PLANE_Z - c[2]) / d[2]
        X = c[:, None] + t * d
        img = _texture(X[0], X[1]).reshape(H, W, 3)
        img[(t <= 0).reshape(H, W)] = 0
        images[v] = img
        scene.get_image(v).image = img
    return scene, images


def test_training_step_on_restrepo_cameras_with_the_mvcnn_twin(oracle_mod):
    import torch
    from oracle import mrf_backward as mb
    from raynet_amd import loss_functions
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.hip_implementations.forward_backward_pass import (
        depth_distribution_from_features, forward_backward_pass)
    from raynet_amd.models import get_nn
    from raynet_amd.mrf import mrf_train
    from raynet_amd.train_network.raynet_batch_provider import (get_batch_of_rays,
                                                                 patches_from_3d_points)
    torch.manual_seed(0)
    scene, images = _scene()
    bbox = np.asarray(scene.bbox, np.float32).ravel()
    gp = GenerationParameters(depth_planes=D, neighbors=VIEWS - 1,
                              grid_shape=np.array(GRID, np.int32),
                              max_number_of_marched_voxels=M, padding=11, gamma_mrf=0.031)
    vg = np.ascontiguousarray(scene.voxel_grid(gp.grid_shape).transpose(1, 2, 3, 0))
    hip = get_context(M, D, VIEWS, 32, H, W, 11, bbox, GRID)
    hip.set_voxel_grid(vg)
    ref = 2
    cam = scene.get_image(ref).camera

    # ---- ~1000 rays of the reference image whose ground-truth point lies inside the box
    rng = np.random.default_rng(3)
    cand = rng.permutation(H * W).astype(np.int32)
    c, d = _pixel_rays(cam, (cand // H).astype(np.float64), (cand % H).astype(np.float64))
    t = (PLANE_Z - c[2]) / d[2]
    X = (c[:, None] + t * d).T
    inside = (t > 0) & np.all(X > bbox[:3] + 0.05, 1) & np.all(X < bbox[3:] - 0.05, 1)
    assert inside.sum() >= 1000, inside.sum()
    images_dev = {v: torch.from_numpy(images[v]).permute(2, 0, 1).contiguous().cuda()
                  for v in range(VIEWS)}
    # candidates -> the reference's rule: a ray ANY of whose D x N patches reaches over an image
    # border is redrawn (common/image.py:189-193, raynet_batch_provider.py:81); the first 1000
    # that pass are the batch
    cand_rays, cand_points, t_cand = cand[inside], X[inside], t[inside]
    batch, valid = get_batch_of_rays(scene, ref, cand_rays, gp, hip, images_dev, cand_points,
                                     return_valid=True)
    valid = valid.cpu().numpy()
    assert valid.sum() >= 1000 and (~valid).sum() > 0, (valid.sum(), len(valid))
    keep_all = get_batch_of_rays(scene, ref, cand_rays[:64], gp, hip, images_dev, cand_points[:64],
                                 reject_border_rays=False)
    assert len(keep_all[VIEWS + 1]) == 64
    ray_idxs, target_points, t_rays = cand_rays[valid][:1000], cand_points[valid][:1000], \
        t_cand[valid][:1000]
    n = len(ray_idxs)
    batch = [b if i == VIEWS else b[:n] for i, b in enumerate(batch)]     # (the voxel grid is no row list)
    patches, (voxel_grid, rvi, rvc, S_target, points, centers) = batch[:VIEWS], batch[VIEWS:]
    assert len(patches) == VIEWS and tuple(patches[0].shape) == (n, D, 3, 11, 11)
    assert tuple(points.shape) == (n, D, 4) and tuple(rvi.shape) == (n, M, 3)
    assert int(rvc.min()) > 1 and int(rvc.max()) < M and torch.all(S_target.sum(1) == 1)
    # the batch is what the reference's provider would deliver:
    #  * the sample points lie on the ray of their pixel, inside the box, first to last
    pts = points.cpu().numpy().astype(np.float64)
    cc, dd = _pixel_rays(cam, (ray_idxs // H).astype(np.float64), (ray_idxs % H).astype(np.float64))
    off = pts[:, :, :3] - cc[None, None, :]
    along = (off * dd.T[:, None, :]).sum(-1)
    assert np.abs(off - along[..., None] * dd.T[:, None, :]).max() < 1e-3
    assert np.all(np.diff(along, axis=1) > 0)
    assert np.all(pts[:, :, :3] > bbox[:3] - 1e-3) and np.all(pts[:, :, :3] < bbox[3:] + 1e-3)
    #  * the reference view's patch centre is the ray's own pixel for every depth hypothesis, and a
    #    patch is the image around its centre (common/image.py:165-200)
    own = patches[0].cpu().numpy()
    px, py = ray_idxs // H, ray_idxs % H
    k = int(np.argmax((px > 6) & (px < W - 6) & (py > 6) & (py < H - 6)))
    want = images[ref][py[k] - 5:py[k] + 6, px[k] - 5:px[k] + 6].transpose(2, 0, 1)
    for dpt in (0, D // 2, D - 1):
        assert np.array_equal(own[k, dpt], want)
    #  * the target's voxel holds the ground-truth point and is on the ray's list
    tv = rvi.cpu().numpy()[np.arange(n), S_target.argmax(1).cpu().numpy()]
    bins = (bbox[3:] - bbox[:3]) / np.array(GRID)
    assert np.mean(np.all(tv == np.floor((target_points - bbox[:3]) / bins), 1)) > 0.97
    #  * at the hypothesis nearest the surface every view sees the same texture (what the
    #    similarity has to find): patch centres agree across views there, not elsewhere
    dist = np.abs(along - t_rays[:, None])
    near = dist.argmin(1)
    ctr = np.stack([p[np.arange(n), near, :, 5, 5].cpu().numpy() for p in patches])     # [V, n, 3]
    far = np.stack([p[np.arange(n), (near + D // 2) % D, :, 5, 5].cpu().numpy() for p in patches])
    visible = np.all(ctr.sum(-1) > 0, 0)
    assert visible.mean() > 0.5
    assert np.abs(ctr[1:, visible] - ctr[0, visible]).mean() < 0.75 * \
        np.abs(far[1:, visible] - far[0, visible]).mean()

    # ---- the MRF block's gradients on exactly these rays against the float64 statement
    model = get_nn("simple_cnn")().cuda().train()
    feats = [model(p.reshape((n * D,) + tuple(p.shape[2:]))).reshape(n, D, -1) for p in patches]
    assert feats[0].shape[-1] == 32                      # 11x11 -> 1x1 x 32 filters
    S = depth_distribution_from_features(feats, VIEWS)
    S.retain_grad()
    starts, ends = points[:, 0, :3].contiguous(), points[:, -1, :3].contiguous()
    gamma = 0.031
    S_mrf = mrf_train.mrf_depth_distribution(S, rvi, rvc, starts, ends, gamma, 3, hip)
    S_mrf.retain_grad()
    loss = loss_functions.squared_emd(S_target, S_mrf).mean()
    loss.backward()
    planes = mrf_train.plane_weights(hip, rvi, rvc, starts, ends)[0].cpu().numpy()
    args = (vg, rvi.cpu().numpy(), rvc.cpu().numpy(), starts.cpu().numpy(), ends.cpu().numpy(), GRID)
    S64 = S.detach().cpu().numpy().astype(np.float64)
    out64 = mb.forward(S64, *args, gamma=gamma, iters=3, planes=planes)
    assert np.abs(S_mrf.detach().cpu().numpy() - out64).max() < 5e-5
    dS = mb.backward(S_mrf.grad.cpu().numpy().astype(np.float64), S64, *args, gamma=gamma, iters=3,
                     planes=planes)
    got = S.grad.cpu().numpy()
    scale = np.abs(dS).max()
    assert scale > 0 and np.abs(got - dS).max() < 2e-3 * scale, (np.abs(got - dS).max(), scale)
    for p_ in model.parameters():
        assert p_.grad is not None and torch.isfinite(p_.grad).all()

    # ---- a few Adam steps through the reference's entry point lower the loss
    gamma_t = torch.tensor(gamma, device="cuda", requires_grad=True)
    opt = torch.optim.Adam(list(model.parameters()) + [gamma_t], lr=2e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = forward_backward_pass(model, patches, voxel_grid, rvi, rvc, S_target, points, centers,
                                     hip, views=VIEWS, gamma=gamma_t, bp_iterations=3,
                                     loss="squared_emd")
        loss.backward()
        assert torch.isfinite(gamma_t.grad)
        opt.step()
        with torch.no_grad():
            gamma_t.clamp_(1e-3, 0.5)
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0], losses
    # patches_from_3d_points pads what falls outside an image with zeros (expand_patch)
    edge = patches_from_3d_points(images_dev[ref], torch.eye(3, 4, device="cuda"),
                                  torch.tensor([[[0.0, 0.0, 1.0, 1.0]]], device="cuda"))
    assert tuple(edge.shape) == (1, 1, 3, 11, 11) and float(edge[0, 0, :, :5, :].abs().sum()) == 0
    assert torch.equal(edge[0, 0, :, 5:, 5:], images_dev[ref][:, :6, :6])
