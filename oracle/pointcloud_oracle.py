"""CPU restatement (NumPy, float64) of the reference's depth-map consumers.

TEST INFRASTRUCTURE ONLY (never imported by raynet_amd).  Pinned by
tests/golden/ref_pointcloud.npz, which tests/golden/gen_pointcloud_from_reference.py
produced by running the reference's own raynet/pointcloud.py and raynet/metrics.py.

    rays                    raynet/common/image.py:242-258
    points_per_image        raynet/pointcloud.py:91-147
    consistent_points       raynet/pointcloud.py:162-246
    nearest_distances       raynet/pointcloud.py:63-72 (KDTree.query, k=1), brute force here
    per_pixel_mean_error    raynet/metrics.py:135-152
"""
import numpy as np


def project(P, point):
    """raynet/utils/geometry.py:9-34 for a (D2, N) array of column points -> (N, D1)."""
    h = np.dot(P, point).T
    return h / h[:, -1:]


def rays(P_pinv, H, W):
    """(4, H*W): pixel (u, v) at column u*H + v."""
    u, v = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")
    pixels = np.stack([u.ravel(), v.ravel(), np.ones(H * W)]).astype(np.float64)
    return project(P_pinv, pixels).T


def clean_depth(depth):
    depth = np.array(depth, copy=True)
    bad = np.isnan(depth)
    if bad.any():
        depth[bad] = depth[~bad].min()            # pointcloud.py:127
    return depth


def selected_pixels(G, borders):
    """Column indices u*H + v of the pixels inside the borders that have ground truth, in the
    reference's order (row-major over the cropped map), pointcloud.py:91-119."""
    H, W = G.shape
    idxs = np.arange(H * W).reshape(W, H).T
    sl = (slice(borders, H - borders), slice(borders, W - borders))
    mask = G[sl] != 0
    return idxs[sl][mask], mask, sl


def points_per_image(P_pinv, center, depth, G, borders):
    """(4, N) float64 points of one frame."""
    H, W = G.shape
    depth = clean_depth(depth)
    R = rays(P_pinv, H, W)
    idxs, mask, sl = selected_pixels(G, borders)
    D = depth[sl][mask].reshape(1, -1)
    R = R[:, idxs]
    center = np.asarray(center, np.float64).reshape(4, 1)
    directions = R - center
    norms = np.sqrt((directions ** 2).sum(axis=0, keepdims=True))
    return center + D * directions / norms


def camera_neighbors(centers, n_neighbors):
    """pointcloud.py:180-192: rows of neighbour frame positions, nearest first."""
    a = np.hstack([np.asarray(c, np.float64).reshape(4, 1) for c in centers])
    distances = 2 * (a * a).sum(axis=0) - 2 * (a.T.dot(a))
    return distances.argsort()[:, 1:n_neighbors + 1]


def consistent_points(frame_pos, cams, depths, gts, borders, threshold, n_neighbors):
    """(4, N') points of frame `frame_pos` that pass the consistency check.  cams: list of
    (P, P_pinv, center); depths: the predicted maps (raw, as loaded)."""
    P, P_pinv, center = cams[frame_pos]
    pts = points_per_image(P_pinv, center, depths[frame_pos], gts[frame_pos], borders)
    neigh = camera_neighbors([c[2] for c in cams], n_neighbors)[frame_pos]
    tau = None
    for i in neigh:
        Pi, _, ci = cams[i]
        H, W = depths[i].shape
        pix = project(Pi, pts).T
        x = np.round(pix[0]).astype(np.int32)
        y = np.round(pix[1]).astype(np.int32)
        valid = (0 <= x) & (x < W) & (0 <= y) & (y < H)
        x[~valid] = 0
        y[~valid] = 0
        predicted = depths[i][y, x]
        dist = np.sqrt(((pts - np.asarray(ci, np.float64).reshape(4, 1)) ** 2).sum(axis=0))
        d = np.abs(predicted - dist)
        tau = d if tau is None else np.maximum(d, tau)
        tau[~valid] = float("inf")
    return pts[:, tau < threshold]


def nearest_distances(ref, query, chunk=2048):
    """Distance from every column of query (3, Nq) to the nearest column of ref (3, Nr)."""
    ref = np.asarray(ref, np.float64)
    query = np.asarray(query, np.float64)
    out = np.empty(query.shape[1])
    for i in range(0, query.shape[1], chunk):
        q = query[:, i:i + chunk]
        d2 = ((q[:, :, None] - ref[:, None, :]) ** 2).sum(axis=0)
        out[i:i + chunk] = np.sqrt(d2.min(axis=1))
    return out


def per_pixel_mean_error(G, D, borders):
    H, W = G.shape
    sl = (slice(borders, H - borders), slice(borders, W - borders))
    g, d = G[sl], D[sl]
    pixels = g != 0
    return np.abs(g[pixels] - d[pixels]).mean()
