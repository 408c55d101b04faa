"""One batch of rays for the training step -- what
raynet/train_network/raynet_batch_provider.py:101-144 (`SingleThreadRayNetBatchProvider.
get_batch_of_rays`) hands to forward_backward_pass, assembled on the GPU:

    [images_0 .. images_{N-1}, voxel_grid, ray_voxel_indices, ray_voxel_count, S_target,
     points, camera_centers]

images_v [n, D, C, h, w]: the 11x11 patch of view v around the projection of each of the
ray's D sample points (`Image.patches_from_3d_points`, common/image.py:145-163: centre =
round(project(P, point)), x = column) -- torch is channels-first where Keras has
[n, D, h, w, C]; points [n, D, 4]: the sampling scheme's points on the ray inside the bounding
box (K8, sampling_schemes.cu:92-122); ray_voxel_indices / ray_voxel_count: the ray's voxel
traversal (K5); S_target [n, M]: the target distribution over the traversed voxels (all mass
in the voxel that holds the ground-truth point, `target_distribution_factory`'s voxel-space
one-hot); camera_centers [n, 4].

The reference draws one sample (ray) at a time from a Python generator and fills NumPy
buffers (raynet_batch_provider.py:62-95); here the n rays of a batch are sampled, traversed
and cut out of the images in a handful of launches.  The sample generators, threading and the
Keras plumbing around it are not rebuilt."""
import numpy as np
import torch


def project_points(P, points):
    """utils/geometry.py:9-34 `project`: [n, D, 4] homogeneous points -> [n, D, 2] pixels."""
    q = points @ P.T
    return q[..., :2] / q[..., 2:3]


def patches_inside(image_hw, P, points, patch_shape=(11, 11)):
    """common/image.py:175-193: [n] bool -- every one of the ray's D patches lies inside the image
    (`min_x >= 0, min_y >= 0, max_x <= w, max_y <= h`).  The reference's `Image.patches` returns
    None for a sample with ANY patch outside, and the batch provider draws another ray
    (raynet_batch_provider.py:81: `if sample.X is not None`)."""
    h, w = patch_shape
    H, W = image_hw
    centre = torch.round(project_points(P, points)).long()
    min_x, max_x = centre[..., 0] - w // 2, centre[..., 0] + w // 2 + w % 2
    min_y, max_y = centre[..., 1] - h // 2, centre[..., 1] + h // 2 + h % 2
    return ((min_x >= 0) & (min_y >= 0) & (max_x <= W) & (max_y <= H)).all(dim=1)


def patches_from_3d_points(image_chw, P, points, patch_shape=(11, 11)):
    """common/image.py:145-200 for every point of every ray: [n, D, C, h, w].  Patches that
    reach over the image border are zero there -- such rays are not part of a reference batch
    at all (patches_inside; get_batch_of_rays drops them by default)."""
    h, w = patch_shape
    C, H, W = image_chw.shape
    pad = max(h, w)
    img = torch.nn.functional.pad(image_chw, (pad, pad, pad, pad))
    centre = torch.round(project_points(P, points)).long()                  # x = column, y = row
    cx = centre[..., 0].clamp(-pad + w // 2, W - 1 + pad - w // 2) + pad
    cy = centre[..., 1].clamp(-pad + h // 2, H - 1 + pad - h // 2) + pad
    dy = torch.arange(-(h // 2), h - h // 2, device=img.device)
    dx = torch.arange(-(w // 2), w - w // 2, device=img.device)
    rows = (cy[..., None, None] + dy[:, None])                               # [n, D, h, 1]
    cols = (cx[..., None, None] + dx[None, :])                               # [n, D, 1, w]
    return img[:, rows, cols].permute(1, 2, 0, 3, 4).contiguous()            # [n, D, C, h, w]


def one_hot_target(voxel_of_point, ray_voxel_indices, ray_voxel_count):
    """[n, M]: all mass in the traversed voxel equal to `voxel_of_point` [n, 3]; a ray whose
    list does not hold that voxel (the point lies off the marched cells by a rounding) puts it
    in the closest traversed one."""
    n, M, _ = ray_voxel_indices.shape
    d = (ray_voxel_indices.long() - voxel_of_point[:, None, :].long()).abs().sum(-1)
    d = d + (torch.arange(M, device=d.device)[None, :] >= ray_voxel_count[:, None]) * (1 << 20)
    target = torch.zeros((n, M), dtype=torch.float32, device=d.device)
    target[torch.arange(n, device=d.device), d.argmin(1)] = 1.0
    return target


def get_batch_of_rays(scene, ref_idx, ray_idxs, generation_params, hip, images, target_points,
                      patch_shape=(11, 11), reject_border_rays=True, return_valid=False):
    """The reference's `inputs` list for the rays of `ray_idxs` of reference image `ref_idx`.

    hip: HipContext of the scene (M, D, H, W, bbox, grid); images: {view: [C, H, W] CUDA
    tensor}; target_points [n, 3]: the ground-truth surface point of every ray.
    reject_border_rays (default, the reference's rule): a ray ANY of whose D x N patches reaches
    over an image border is not in the batch (common/image.py:189-193 returns None, the provider
    redraws); the batch then holds the `valid` rays only, in order -- draw more candidates than
    the batch needs.  return_valid: also return the [len(ray_idxs)] bool mask.
    Layout notes: images_v is channels-first [n, D, C, h, w] (Keras: [n, D, h, w, C]); voxel_grid
    is [gx][gy][gz][3] (the layout forward_pass.py:573-575 hands the kernels; the training
    provider's `voxel_grid` input is the reference's [3][gx][gy][gz] -- transpose(3, 0, 1, 2))."""
    gp = generation_params
    views = scene.view_indices_with_neighbors(ref_idx, gp.neighbors)
    cam = scene.get_image(ref_idx).camera
    dev = hip.device
    ridx = hip.dev(np.ascontiguousarray(ray_idxs, dtype=np.int32))
    n, D, M = len(ridx), gp.depth_planes, gp.max_number_of_marched_voxels
    P_inv = hip.dev(np.ascontiguousarray(cam.P_pinv, dtype=np.float32))
    center = hip.dev(np.ascontiguousarray(cam.center, dtype=np.float32).ravel())
    points = torch.zeros((n, D, 4), dtype=torch.float32, device=dev)
    hip.sample_points(ridx, P_inv, center, points)                                     # K8
    rvi = torch.zeros((n, M, 3), dtype=torch.int32, device=dev)
    rvc = torch.zeros((n,), dtype=torch.int32, device=dev)
    hip.voxel_traversal(points[:, 0, :3].contiguous(), points[:, -1, :3].contiguous(), rvi, rvc)   # K5
    patches = []
    valid = torch.ones((n,), dtype=torch.bool, device=dev)
    for v in views:
        P = torch.as_tensor(np.asarray(scene.get_image(v).camera.P, np.float32), device=dev)
        patches.append(patches_from_3d_points(images[v], P, points, patch_shape))
        valid &= patches_inside(images[v].shape[1:], P, points, patch_shape)
    bbox = torch.as_tensor(np.asarray(scene.bbox, np.float32).ravel(), device=dev)
    grid = torch.as_tensor(np.asarray(hip.grid_shape, np.float32), device=dev)
    voxel = torch.floor((torch.as_tensor(target_points, dtype=torch.float32, device=dev) - bbox[:3]) /
                        ((bbox[3:] - bbox[:3]) / grid)).clamp_min(0).minimum(grid - 1)
    S_target = one_hot_target(voxel, rvi, rvc)
    voxel_grid = hip.dev(np.ascontiguousarray(scene.voxel_grid(gp.grid_shape).transpose(1, 2, 3, 0)))
    centers = center[None, :].expand(n, 4).contiguous()
    if reject_border_rays:
        patches = [p[valid] for p in patches]
        rvi, rvc, S_target, points, centers = (t[valid] for t in (rvi, rvc, S_target, points, centers))
    out = patches + [voxel_grid, rvi, rvc, S_target, points, centers]
    return (out, valid) if return_valid else out
