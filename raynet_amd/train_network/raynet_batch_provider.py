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


def patches_from_3d_points(image_chw, P, points, patch_shape=(11, 11)):
    """common/image.py:145-200 for every point of every ray: [n, D, C, h, w].  Patches that
    reach over the image border are zero there (the reference's `expand_patch` padding)."""
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
                      patch_shape=(11, 11)):
    """The reference's `inputs` list for `n = len(ray_idxs)` rays of reference image `ref_idx`.

    hip: HipContext of the scene (M, D, H, W, bbox, grid); images: {view: [C, H, W] CUDA
    tensor}; target_points [n, 3]: the ground-truth surface point of every ray."""
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
    for v in views:
        P = torch.as_tensor(np.asarray(scene.get_image(v).camera.P, np.float32), device=dev)
        patches.append(patches_from_3d_points(images[v], P, points, patch_shape))
    bbox = torch.as_tensor(np.asarray(scene.bbox, np.float32).ravel(), device=dev)
    grid = torch.as_tensor(np.asarray(hip.grid_shape, np.float32), device=dev)
    voxel = torch.floor((torch.as_tensor(target_points, dtype=torch.float32, device=dev) - bbox[:3]) /
                        ((bbox[3:] - bbox[:3]) / grid)).clamp_min(0).minimum(grid - 1)
    S_target = one_hot_target(voxel, rvi, rvc)
    voxel_grid = hip.dev(np.ascontiguousarray(scene.voxel_grid(gp.grid_shape).transpose(1, 2, 3, 0)))
    centers = center[None, :].expand(n, 4).contiguous()
    return patches + [voxel_grid, rvi, rvc, S_target, points, centers]
