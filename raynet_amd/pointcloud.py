"""Depth maps -> point clouds on MI355X: the counterpart of raynet/pointcloud.py.

Same classes, constructor arguments and conventions as the reference (points are (3, N)
arrays, depth maps are the (H, W) float32 files / arrays the forward pass produces,
`scripts/forward_pass.py:136-142`); the work runs in HIP kernels (csrc/raynet_eval.inl):
back-projection and the multi-view consistency check in float64 like the NumPy code, the
nearest-neighbour queries as an exact brute-force scan instead of a host KD-tree.
"""
import sys

import numpy as np
import torch

from .hip_implementations import get_context


def _dev(x, dtype):
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to("cuda", dtype).contiguous()


def _xyzw(points):
    """(3, N) -> [N][4] float32 device tensor."""
    p = _dev(points, torch.float32)
    out = torch.zeros((p.shape[1], 4), dtype=torch.float32, device="cuda")
    out[:, :3] = p.t()
    return out


class Pointcloud(object):
    """raynet/pointcloud.py:14-72."""

    def __init__(self, points):
        self._points = points

    @property
    def points(self):
        return self._points

    def save_ply(self, file):
        N = self.points.shape[1]
        with open(file, "wb") as f:
            f.write(("ply\nformat binary_%s_endian 1.0\ncomment Raynet pointcloud!\n"
                     "element vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                     "end_header\n" % (sys.byteorder, N)).encode())
            np.asarray(self.points).T.astype(np.float32).tofile(f)

    def save(self, file):
        np.save(file, self.points)

    def filter(self, mask):
        self._points = mask.filter(self.points)

    def index(self, leaf_size=40, metric="minkowski"):
        """The reference builds a KD-tree here; the scan needs no index, only the points on
        the device (kept until the cloud changes)."""
        pts = self.points
        if getattr(self, "_index", None) is None or self._index[0] is not pts:
            self._index = (pts, _xyzw(pts))

    def nearest_neighbors(self, X, k=1, return_distances=True):
        """Like KDTree.query(X.T, 1): ((Nq, 1) distances, (Nq, 1) indices) of the closest
        point of this cloud for every column of X."""
        assert k == 1, "only the nearest neighbour (what the metrics use) is implemented"
        self.index()
        ctx = get_context()
        q = _xyzw(X)
        dist = torch.empty((q.shape[0],), dtype=torch.float32, device="cuda")
        idx = torch.empty((q.shape[0],), dtype=torch.int32, device="cuda")
        ctx.nearest_neighbors(self._index[1], q, dist, idx)
        idx = idx.cpu().numpy().astype(np.int64).reshape(-1, 1)
        if not return_distances:
            return idx
        return dist.cpu().numpy().astype(np.float64).reshape(-1, 1), idx


class PointcloudFromDepthMaps(Pointcloud):
    """raynet/pointcloud.py:76-160.  depthmaps: .npy file names (as the reference) or arrays."""

    def __init__(self, scene, frame_idxs, depthmaps, borders=40):
        self._scene = scene
        self._frame_idxs = frame_idxs
        self._depthmaps = depthmaps
        self._borders = borders
        self._points = None

    @staticmethod
    def _load(d):
        return np.load(d) if isinstance(d, str) else np.asarray(d)

    def _selected_pixels(self, G):
        """Indices u*H + v of the pixels that survive _remove_unwanted_points
        (pointcloud.py:91-119), in the reference's order."""
        H, W = G.shape
        b = self._borders
        idxs = torch.arange(H * W, device="cuda").reshape(W, H).t()
        G = _dev(G, torch.float32)
        sl = (slice(b, H - b), slice(b, W - b))
        return idxs[sl][G[sl] != 0]

    def _all_points(self, frame, depth):
        """(3, H*W) float64 device points of every pixel of `frame` (pointcloud.py:121-147)."""
        ctx = get_context()
        cam = self._scene.get_image(frame).camera
        depth = _dev(depth, torch.float32)
        bad = torch.isnan(depth)
        if bool(bad.any()):
            depth = torch.where(bad, depth[~bad].min(), depth)      # pointcloud.py:127
        H, W = depth.shape
        pts = torch.empty((3, H * W), dtype=torch.float64, device="cuda")
        ctx.depthmap_points(H, W, _dev(cam.P_pinv, torch.float64),
                            _dev(np.asarray(cam.center).reshape(4), torch.float64), depth, pts)
        return pts

    def _generate_points_per_image(self, frame, predicted_depth):
        depth = self._load(predicted_depth)
        sel = self._selected_pixels(self._scene.get_depth_map(frame))
        return self._all_points(frame, depth)[:, sel]

    @property
    def points(self):
        if self._points is None:
            pts = [self._generate_points_per_image(i, d)
                   for i, d in zip(self._frame_idxs, self._depthmaps)]
            self._points = torch.cat(pts, dim=1).cpu().numpy()
        return self._points


class PointcloudFromDepthMapsWithConsistency(PointcloudFromDepthMaps):
    """raynet/pointcloud.py:162-246."""

    def __init__(self, scene, frame_idxs, depthmaps, borders=40, consistency_threshold=0.75,
                 n_neighbors=5):
        self._consistency_threshold = consistency_threshold
        self._n_neighbors = n_neighbors
        self._camera_neighbors = None
        self._frame_idxs_map = dict(zip(frame_idxs, range(len(frame_idxs))))
        super(PointcloudFromDepthMapsWithConsistency, self).__init__(scene, frame_idxs, depthmaps,
                                                                     borders)

    def _neighbor_frames(self, frame):
        if self._camera_neighbors is None:
            a = np.hstack([np.asarray(self._scene.get_image(i).camera.center,
                                      np.float64).reshape(4, 1) for i in self._frame_idxs])
            distances = 2 * (a * a).sum(axis=0) - 2 * (a.T.dot(a))
            self._camera_neighbors = distances.argsort()[:, 1:self._n_neighbors + 1]
        return [(self._frame_idxs[i], self._depthmaps[i])
                for i in self._camera_neighbors[self._frame_idxs_map[frame]]]

    def _generate_points_per_image(self, frame, predicted_depth):
        ctx = get_context()
        pts = super(PointcloudFromDepthMapsWithConsistency, self)._generate_points_per_image(
            frame, predicted_depth).contiguous()
        tau = torch.empty((pts.shape[1],), dtype=torch.float64, device="cuda")
        for k, (i, d) in enumerate(self._neighbor_frames(frame)):
            cam = self._scene.get_image(i).camera
            depth = _dev(self._load(d), torch.float32)      # raw map, NaNs included (:232)
            H, W = depth.shape
            ctx.consistency_tau(H, W, k == 0, pts, _dev(cam.P, torch.float64),
                                _dev(np.asarray(cam.center).reshape(4), torch.float64), depth, tau)
        return pts[:, tau < self._consistency_threshold]


def get_pointcloud(scene, frame_idxs, depthmaps, with_consistency, **kwargs):
    """raynet/pointcloud.py:248-269."""
    if with_consistency:
        return PointcloudFromDepthMapsWithConsistency(
            scene, frame_idxs, depthmaps, kwargs["borders"], kwargs["consistency_threshold"],
            kwargs["n_neighbors"])
    return PointcloudFromDepthMaps(scene, frame_idxs, depthmaps, kwargs["borders"])
