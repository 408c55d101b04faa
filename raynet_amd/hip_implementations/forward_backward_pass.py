"""End-to-end training step of RayNet on MI355X: the torch counterpart of
raynet/tf_implementations/forward_backward_pass.py:128-248.

MV-CNN on the per-ray patches -> pairwise feature similarities -> softmax over the D depth
hypotheses (dense torch, autograd) -> planes->voxels + clip/renorm (torch) -> BP sweeps and
the depth distribution (HIP forward, analytic HIP backward: raynet_amd/mrf/mrf_train.py)
-> loss.  The function returns the scalar loss; `loss.backward()` and the optimiser step
belong to the caller (the reference returns Keras `updates` instead,
forward_backward_pass.py:244-246)."""
import torch

from .. import loss_functions
from ..mrf.mrf_train import mrf_depth_distribution


def compute_similarities(n1, n2, features):
    """forward_backward_pass.py:10-34: per ray and depth hypothesis, <f_n1, f_n2>."""
    return (features[n1] * features[n2]).sum(-1)


def depth_distribution_from_features(features, views):
    """forward_backward_pass.py:185-193: mean over the view pairs, softmax over D."""
    S = 0
    for n1 in range(views):
        for n2 in range(n1 + 1, views):
            S = S + compute_similarities(n1, n2, features)
    S = S / ((views * (views - 1)) / 2.0)
    return torch.softmax(S, dim=-1)


def forward_backward_pass(model, images, voxel_grid, ray_voxel_indices, ray_voxel_count,
                          S_target, points, camera_centers, hip, views=5, gamma=0.031,
                          bp_iterations=3, loss="squared_emd"):
    """Arguments as forward_backward_pass.py:128-170, plus the HipContext:
    images: list of `views` tensors [n, D, C, h, w] (torch is channels-first);
    voxel_grid [gx, gy, gz, 3]; ray_voxel_indices [n, M, 3] int32; ray_voxel_count [n] int32;
    S_target [n, M]; points [n, D, 4] (first / last sample = ray start / end);
    camera_centers [n, 4]; gamma float or 0-d tensor (trainable)."""
    n, D = images[0].shape[:2]
    features = [model(img.reshape((n * D,) + tuple(img.shape[2:]))).reshape(n, D, -1)
                for img in images]
    S = depth_distribution_from_features(features, views)
    S_mrf = mrf_depth_distribution(S, ray_voxel_indices, ray_voxel_count,
                                   points[:, 0, :3].float(), points[:, -1, :3].float(), gamma,
                                   bp_iterations, hip)
    if loss == "expected_squared_error":
        return loss_functions.expected_squared_error(S_target, S_mrf, voxel_grid,
                                                     ray_voxel_indices, camera_centers).mean()
    return loss_functions.loss_factory(loss)(S_target, S_mrf).mean()
