"""SURVEY.md 8(f) row 3 on the GPU: raynet_amd.pointcloud / raynet_amd.metrics (HIP kernels of
csrc/raynet_eval.inl) against the reference's own outputs (tests/golden/ref_pointcloud.npz)
and, at a size the KD-tree-free scan is meant for, against the NumPy oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


class _Cam(object):
    def __init__(self, P, P_pinv, center):
        self.P, self.P_pinv, self.center = P, P_pinv, center


class _Img(object):
    def __init__(self, cam, H, W):
        self.camera, self.height, self.width = cam, H, W


class _Scene(object):
    def __init__(self, g, cloud_cls):
        H, W = int(g["H"]), int(g["W"])
        self.image_shape = (H, W)
        self._g = g
        self._cloud_cls = cloud_cls
        self._images = [_Img(_Cam(g["P"][i], g["P_pinv"][i], g["center"][i]), H, W)
                        for i in range(len(g["P"]))]

    def get_image(self, i):
        return self._images[i]

    def get_depth_map(self, i):
        return self._g["gt"][i]

    def get_depthmap_file(self, i):
        return self._g["gt"][i]          # arrays are accepted wherever file names are

    def get_pointcloud(self):
        return self._cloud_cls(self._g["gt_cloud"].copy())


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "ref_pointcloud.npz"))


def test_points_and_consistency_match_reference(g):
    from raynet_amd import pointcloud as pc
    scene = _Scene(g, pc.Pointcloud)
    frames = list(range(len(g["P"])))
    plain = pc.PointcloudFromDepthMaps(scene, frames, list(g["pred"]), int(g["borders"]))
    assert plain.points.shape == g["points_plain"].shape and plain.points.dtype == np.float64
    assert np.abs(plain.points - g["points_plain"]).max() < 1e-9
    cons = pc.get_pointcloud(scene, frames, list(g["pred"]), True, borders=int(g["borders"]),
                             consistency_threshold=float(g["consistency_threshold"]),
                             n_neighbors=int(g["n_neighbors"]))
    assert cons.points.shape == g["points_consistency"].shape      # the same points survive
    assert np.abs(cons.points - g["points_consistency"]).max() < 1e-9


@pytest.mark.parametrize("name", ["plain", "consistency"])
@pytest.mark.parametrize("use_dm", [False, True])
def test_metrics_match_reference(g, name, use_dm):
    from raynet_amd import metrics, pointcloud as pc
    scene = _Scene(g, pc.Pointcloud)
    frames = list(range(len(g["P"])))
    tag = "%s_%s" % (name, "dm" if use_dm else "pc")
    pred = pc.Pointcloud(g["points_" + name].copy())
    acc, pts = metrics.Accuracy(None, truncate=0.3, borders=int(g["borders"]),
                                use_pc_from_depthmap=use_dm).compute(scene, frames, None, pred)
    assert acc.shape == (pred.points.shape[1], 1) and pts is pred.points
    assert np.abs(acc.ravel() - g["accuracy_" + tag]).max() < 2e-6        # fp32 scan vs f64 KD-tree
    comp, pts = metrics.Completeness(None, truncate=0.3, borders=int(g["borders"]),
                                     use_pc_from_depthmap=use_dm).compute(scene, frames, None, pred)
    assert np.abs(comp.ravel() - g["completeness_" + tag]).max() < 2e-6
    assert np.abs(np.asarray(pts) - g["completeness_points_" + tag]).max() < 1e-6
    err, _ = metrics.PerPixelMeanDepthError(int(g["borders"])).compute(scene, frames,
                                                                       list(g["pred"]), None)
    assert np.allclose(err, g["per_pixel_error"], rtol=1e-6, equal_nan=True)


def test_nearest_neighbours_at_scale_vs_oracle():
    """200k queries against 150k reference points (sizes that are not multiples of any
    tile): distances equal the float64 brute force, indices point at a nearest point."""
    from oracle import pointcloud_oracle as po
    from raynet_amd import pointcloud as pc
    rng = np.random.default_rng(0)
    ref = rng.standard_normal((3, 150001)).astype(np.float32)
    qry = rng.standard_normal((3, 200003)).astype(np.float32)
    d, idx = pc.Pointcloud(ref).nearest_neighbors(qry)
    sample = rng.choice(qry.shape[1], 3000, replace=False)
    truth = po.nearest_distances(ref, qry[:, sample])
    assert np.abs(d[sample, 0] - truth).max() < 1e-5
    picked = np.linalg.norm(ref[:, idx[sample, 0]].astype(np.float64) - qry[:, sample], axis=0)
    assert np.abs(picked - truth).max() < 1e-5
    # a point of the cloud is its own neighbour
    d0, i0 = pc.Pointcloud(ref).nearest_neighbors(ref[:, :1000])
    assert np.all(d0 == 0) and np.array_equal(i0.ravel(), np.arange(1000))


def test_end_to_end_from_forward_pass_depth_maps():
    """The forward pass's own depth maps -> point cloud -> accuracy against the planted
    sphere of the synthetic scene: most points lie within a voxel of the surface."""
    import torch  # noqa: F401
    from raynet_amd import pointcloud as pc
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    H, W, grid = 96, 128, (64, 64, 64)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
    gp = GenerationParameters(depth_planes=32, neighbors=4, grid_shape=np.array(grid, np.int32),
                              max_number_of_marched_voxels=192, padding=11, gamma_mrf=0.05)
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
    depths = list(fp.forward_pass(scene, (0, 5, 1)))
    masks = [np.ones((H, W), np.float32) for _ in range(5)]
    type(scene).get_depth_map = lambda self, i: masks[i]
    cloud = pc.PointcloudFromDepthMaps(scene, list(range(5)), depths, borders=8)
    pts = cloud.points
    assert pts.shape == (3, 5 * (H - 16) * (W - 16)) and np.isfinite(pts).all()
    # points come frame by frame, row-major over the cropped map: the central pixels of every
    # view look at the planted sphere (radius 0.5 around (0, 0, -0.1)) and must land on it
    Hc, Wc = H - 16, W - 16
    r = np.linalg.norm(pts - np.array([[0.0], [0.0], [-0.1]]), axis=0).reshape(5, Hc, Wc)
    centre = r[:, Hc // 2 - 4:Hc // 2 + 4, Wc // 2 - 4:Wc // 2 + 4]
    assert np.abs(np.median(centre, axis=(1, 2)) - 0.5).max() < 0.08
