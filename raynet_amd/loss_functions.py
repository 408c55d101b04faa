"""Training losses of the reference (raynet/tf_implementations/loss_functions.py:4-48) in
torch; all act on depth distributions over the traversed voxels, [n, M]."""
import torch


def emd(y_true, y_pred):
    """Earth mover's distance, loss_functions.py:4-6."""
    return torch.cumsum(y_true - y_pred, dim=-1).abs().mean(dim=-1)


def squared_emd(y_true, y_pred):
    """loss_functions.py:9-11."""
    return torch.cumsum(y_true - y_pred, dim=-1).pow(2).sum(dim=-1)


def expected_squared_error(y_true, y_pred, voxel_grid, ray_voxel_indices, camera_center):
    """loss_functions.py:14-34: |E_true[depth] - E_pred[depth]| with depth = distance of the
    voxel centre from the camera centre (the name says squared, the reference returns the
    absolute difference).  voxel_grid: [gx, gy, gz, 3]."""
    idx = ray_voxel_indices.long()
    centers = voxel_grid[idx[..., 0], idx[..., 1], idx[..., 2]]
    cam = camera_center[:, :3].reshape(-1, 1, 3)
    dists = (centers - cam).pow(2).sum(-1).sqrt()
    return ((y_true * dists).sum(1) - (y_pred * dists).sum(1)).abs()


def loss_factory(loss):
    """loss_functions.py:37-48 (unknown names fall back to emd there too)."""
    if loss == "categorical_crossentropy":
        return lambda t, p: -(t * torch.log(p.clamp_min(1e-7))).sum(-1)
    if loss == "squared_emd":
        return squared_emd
    if loss == "mse":
        return lambda t, p: (t - p).pow(2).mean(-1)
    return emd
