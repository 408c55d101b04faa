"""SURVEY.md 8(f) row 3 on the CPU: the NumPy restatement (oracle/pointcloud_oracle.py)
against the outputs of the reference's own pointcloud.py / metrics.py
(tests/golden/ref_pointcloud.npz, made by tests/golden/gen_pointcloud_from_reference.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "ref_pointcloud.npz"))


def _cams(g):
    return [(g["P"][i], g["P_pinv"][i], g["center"][i]) for i in range(len(g["P"]))]


def test_points_match_reference(g):
    from oracle import pointcloud_oracle as po
    b = int(g["borders"])
    pts = np.hstack([po.points_per_image(g["P_pinv"][i], g["center"][i], g["pred"][i], g["gt"][i], b)
                     for i in range(len(g["P"]))])[:-1]
    assert pts.shape == g["points_plain"].shape
    assert np.abs(pts - g["points_plain"]).max() < 1e-9


def test_consistency_filter_matches_reference(g):
    from oracle import pointcloud_oracle as po
    cams = _cams(g)
    pts = np.hstack([po.consistent_points(i, cams, list(g["pred"]), list(g["gt"]), int(g["borders"]),
                                          float(g["consistency_threshold"]), int(g["n_neighbors"]))
                     for i in range(len(cams))])[:-1]
    assert pts.shape == g["points_consistency"].shape          # the same points survive
    assert np.abs(pts - g["points_consistency"]).max() < 1e-9


def test_accuracy_completeness_match_reference_kdtree(g):
    from oracle import pointcloud_oracle as po
    cams = _cams(g)
    b = int(g["borders"])
    gt_dm = np.hstack([po.points_per_image(g["P_pinv"][i], g["center"][i], g["gt"][i], g["gt"][i], b)
                       for i in range(len(cams))])[:-1]
    for name in ("plain", "consistency"):
        pred = g["points_" + name]
        for tag, gt_cloud in (("pc", g["gt_cloud"]), ("dm", gt_dm)):
            acc = np.minimum(po.nearest_distances(gt_cloud, pred), 0.3)
            comp = np.minimum(po.nearest_distances(pred, gt_cloud), 0.3)
            assert np.abs(acc - g["accuracy_%s_%s" % (name, tag)]).max() < 1e-9
            assert np.abs(comp - g["completeness_%s_%s" % (name, tag)]).max() < 1e-9


def test_per_pixel_error_matches_reference(g):
    from oracle import pointcloud_oracle as po
    err = [po.per_pixel_mean_error(g["gt"][i], g["pred"][i], int(g["borders"]))
           for i in range(len(g["P"]))]
    # the reference leaves the NaN of frame 1 in (np.load again, no clean-up): NaN mean
    assert np.allclose(err, g["per_pixel_error"], rtol=1e-6, equal_nan=True)
