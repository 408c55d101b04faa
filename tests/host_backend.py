"""A HOST stand-in for HipContext's resident-scene methods, built on the oracle.

TEST INFRASTRUCTURE.  It exists so the world_size-2 gloo test can drive the real
RayNetForwardPass sharding / all-reduce / prior-once logic on CPU tensors; the test's
rank processes put it where the driver looks for its context (they replace
raynet_amd.forward_pass.perform_raynet_fp); it is never importable from the product package."""
import numpy as np
import torch

from oracle import oracle


class OracleBackend(object):
    def __init__(self, M, D, N, F, H, W, padding, bbox, grid_shape):
        self.o = oracle.Oracle(M, D, N, F, H, W, padding, bbox, grid_shape)
        self.device = torch.device("cpu")
        self.grid_shape = tuple(int(g) for g in grid_shape)
        self.M, self.N = int(M), int(N)
        self._grid_set = False
        self._vg = None

    def dev(self, x, dtype=None):
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous()

    def set_voxel_grid(self, vg):
        self._vg = self.dev(vg).numpy().reshape(self.grid_shape + (3,))
        self._grid_set = True

    def acc_copies(self):
        return 1

    # the stand-in keeps the reference's [gx][gy][gz] layout (flat)
    def acc_size(self):
        return int(np.prod(self.grid_shape))

    def acc_to_grid(self, acc):
        return acc.reshape(self.grid_shape).clone()

    @staticmethod
    def _unpack(vox):
        v = vox.numpy()
        return np.ascontiguousarray(np.stack([v >> 20, (v >> 10) & 1023, v & 1023], axis=-1)
                                    .astype(np.int32))

    def count_voxels(self, ray_idxs, cameras):
        """int32 [n_images, n] voxel counts (what rn_scene_count_voxels returns)."""
        cam = cameras.numpy()
        N = self.N
        out = np.zeros((len(cam), len(ray_idxs)), np.int32)
        for k in range(len(cam)):
            s, e = self.o.sample(ray_idxs.numpy(), cam[k, 12 * N:12 * N + 12], cam[k, 12 * N + 12:])
            out[k] = self.o.traversal(s, e)[1]
        return torch.from_numpy(out)

    def scene_prepare(self, ridx, feature_views, P, P_inv, center, vox, rvc, Sr, order=None):
        feats = np.stack([f.numpy() for f in feature_views])
        s, e = self.o.sample(ridx.numpy(), P_inv.numpy(), center.numpy())
        S = self.o.similarities(feats, P.numpy(), s, e)
        rvi, cnt = self.o.traversal(s, e)
        Sv = self.o.planes_to_voxels(self._vg, rvi, cnt, s, e, S)
        vox.numpy()[...] = (rvi[..., 0] << 20) | (rvi[..., 1] << 10) | rvi[..., 2]
        rvc.numpy()[...] = cnt
        Sr.numpy()[...] = Sv        # clipped + renormalised inside the oracle's BP / depth calls

    def scene_bp_sweep(self, Sr, vox, rvc, acc_in, msgs, acc_part, first_sweep=False,
                       patch_rows=False, uniform_acc=False):
        if first_sweep:
            msgs.zero_()
        m = np.ascontiguousarray(msgs.numpy())
        out = np.ascontiguousarray(acc_part[0].numpy().reshape(self.grid_shape))
        self.o.bp_sweep(Sr.numpy(), self._unpack(vox), rvc.numpy(),
                        acc_in.numpy().reshape(self.grid_shape), m, out)
        msgs.numpy()[...] = m
        acc_part[0].numpy()[...] = out.ravel()

    def acc_reduce_local(self, acc_part, acc_out):
        acc_out.copy_(acc_part.sum(0))
        acc_part.zero_()

    def acc_add_prior(self, acc, prior):
        acc.add_(np.float32(prior))

    def acc_combine(self, acc_part, prior, acc_out):
        acc_out.copy_(acc_part.sum(0) + np.float32(prior))
        acc_part.zero_()

    def scene_depth(self, Sr, vox, rvc, acc, msgs, center, S_new, depth_map, rays_per_center=0):
        rvi = self._unpack(vox)
        Sn = self.o.depth_distribution(Sr.numpy(), rvi, rvc.numpy(),
                                       acc.numpy().reshape(self.grid_shape), msgs.numpy())
        if S_new is not None:
            S_new.numpy()[...] = Sn
        if depth_map is not None:
            c = center.numpy().reshape(-1, 4)
            n = len(Sn)
            step = rays_per_center if rays_per_center > 0 else max(n, 1)
            for g, lo in enumerate(range(0, n, step)):
                hi = min(lo + step, n)
                depth_map.numpy()[lo:hi] = self.o.depth_from_distribution(
                    Sn[lo:hi], rvi[lo:hi], self._vg, c[g if rays_per_center > 0 else 0])
