"""the on-disk scene loaders (raynet_amd/common/scene.py RestrepoScene /
DTUScene <- raynet/common/scene.py:144-452).  The reference's own tests/test_scene.py:51-128
(temporary Restrepo dataset: 50 random views + scene_info.xml, neighbour assertions) restated,
the cameras of its mock scene_1 (tests/golden/restrepo_mock_scene_1) read through the loader,
and a temporary DTU-layout scan whose files are generated from known K, R, t."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest
from PIL import Image as PILImage

from conftest import GOLDEN


def _write_scene_info(tmp, bbox):
    root = ET.Element("bwm_info_for_boxm2")
    ET.SubElement(root, "bbox", dict(zip(("minx", "miny", "minz", "maxx", "maxy", "maxz"),
                                         [str(b) for b in bbox])))
    ET.SubElement(root, "resolution", {"val": "0.001"})
    ET.SubElement(root, "ntrees", {"ntrees_x": "48", "ntrees_y": "48", "ntrees_z": "48"})
    ET.ElementTree(root).write(os.path.join(tmp, "scene_info.xml"))


def _restrepo_dataset(tmp, n=50, H=20, W=30, with_gt=False):
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(tmp, "imgs"))
    os.makedirs(os.path.join(tmp, "cams_krt"))
    cams, images = [], []
    for i in range(n):
        D = (rng.random((H, W, 3)) * 255).astype(np.uint8)
        PILImage.fromarray(D).save(os.path.join(tmp, "imgs", "frame_%03d.png" % i))
        K, R, t = rng.random((3, 3)) * 200, rng.random((3, 3)) * 10, rng.random((1, 3))
        with open(os.path.join(tmp, "cams_krt", "camera_%03d.txt" % i), "w") as f:
            np.savetxt(f, K)
            f.write("\n")
            np.savetxt(f, R)
            f.write("\n")
            np.savetxt(f, t)
        cams.append((K, R, t))
        images.append(D)
    _write_scene_info(tmp, [-2.5, -2.5, -0.5, 2.5, 2.5, 0.5])
    if with_gt:
        os.makedirs(os.path.join(tmp, "gt"))
        np.save(os.path.join(tmp, "gt", "gt_depth_3.npy"), np.full((H, W), 2.5, np.float32))
    return cams, images


def test_restrepo_scene_like_the_reference_test(tmp_path):
    """tests/test_scene.py:108-128 of the reference."""
    from raynet_amd.common import scene
    cams, images = _restrepo_dataset(str(tmp_path), with_gt=True)
    s = scene.RestrepoScene(str(tmp_path))
    assert s.n_images == 50
    assert s.image_shape == (20, 30)
    bbox = s.bbox
    assert bbox.shape == (1, 6) and bbox.dtype == np.float32
    assert list(bbox[0]) == [-2.5, -2.5, -0.5, 2.5, 2.5, 0.5]
    assert list(s._get_neighbor_idxs(0, 4)) == [1, 2, 3, 4]
    assert list(s._get_neighbor_idxs(1, 4)) == [0, 2, 3, 4]
    assert list(s._get_neighbor_idxs(35, 4)) == [33, 34, 36, 37]
    assert list(s._get_neighbor_idxs(50, 4)) == [46, 47, 48, 49]
    # images come back scaled to [0, 1] float32, cameras as float32 K, R, t (image.py:16-21)
    im = s.get_image(7)
    assert im.image.dtype == np.float32 and im.image.shape == (20, 30, 3)
    assert np.array_equal(im.image, images[7].astype(np.float32) / np.float32(255.))
    K, R, t = cams[7]
    assert np.array_equal(im.camera.K, K.astype(np.float32))
    assert np.array_equal(im.camera.R, R.astype(np.float32))
    assert np.array_equal(im.camera.t, t.astype(np.float32).reshape(3, 1))
    assert im.camera.P.shape == (3, 4) and im.camera.P_pinv.shape == (4, 3)
    # reference first, then the neighbours
    views = s.get_image_with_neighbors(35, 4)
    assert views[0] is s.get_image(35) and views[1] is s.get_image(33)
    assert s.view_indices_with_neighbors(35, 4) == [35, 33, 34, 36, 37]
    # ground truth depth maps when the dataset ships them
    assert s.get_depthmap_file(3).endswith("gt_depth_3.npy") and s.get_depth_map(3)[0, 0] == 2.5
    assert s.get_depthmap_file(4) is None
    with pytest.raises(NotImplementedError):
        s.get_depth_map(4)


def test_restrepo_mock_scene_cameras_and_distance_neighbours(tmp_path):
    """The camera files + scene_info.xml of the reference's mock scene_1 through the loader
    (images of the right names are generated: the 34 MB of PNGs are not fixtures)."""
    import shutil
    from raynet_amd.common import scene
    src = os.path.join(GOLDEN, "restrepo_mock_scene_1")
    dst = str(tmp_path / "scene_1")
    shutil.copytree(src, dst)
    os.makedirs(os.path.join(dst, "imgs"))
    for c in sorted(os.listdir(os.path.join(dst, "cams_krt"))):
        PILImage.fromarray(np.zeros((9, 16, 3), np.uint8)).save(
            os.path.join(dst, "imgs", c.replace("_cam.txt", ".png")))
    s = scene.get_scene("restrepo", dst, select_neighbors_based_on="distance")
    assert s.n_images == 12 and s.image_shape == (9, 16)
    assert np.allclose(s.bbox.ravel(), [-5, -5, -0.7, 5, 5, 1.5])
    shim = scene.restrepo_cameras_scene(src, (9, 16))
    for i in range(12):
        assert np.array_equal(s.get_image(i).camera.P, shim.get_image(i).camera.P)
    # nearest camera centres (scene.py:59-79)
    centers = np.hstack([s.get_image(i).camera.center for i in range(12)])[:3]
    n3 = s._get_neighbor_idxs(0, 3)
    d = np.linalg.norm(centers - centers[:, :1], axis=0)
    assert sorted(n3) == sorted(np.argsort(d)[1:4].tolist())


def test_dtu_scene(tmp_path):
    """A two-frame scan in the DTU layout written from known K, R, t: the loader recovers
    them ([R t] = K^-1 P, scene.py:329-365), reads the bbox from the ObsMask .mat, keeps only
    the requested illumination, and turns the z-depth map into distances from the camera
    centre (scene.py:372-407)."""
    from scipy.io import savemat
    from raynet_amd.common import scene
    from raynet_amd.common.camera import Camera
    base = str(tmp_path)
    H, W = 6, 8
    cal = os.path.join(base, "SampleSet/MVS_Data/Calibration/cal18")
    os.makedirs(cal)
    os.makedirs(os.path.join(base, "SampleSet/MVS_Data/ObsMask"))
    os.makedirs(os.path.join(base, "Rectified/scan007"))
    os.makedirs(os.path.join(base, "Depth/scan007"))
    K = np.array([[50.0, 0, W / 2.0], [0, 50.0, H / 2.0], [0, 0, 1.0]])
    np.savetxt(os.path.join(cal, "intrinsic.txt"), K)
    cams = []
    for k in range(2):
        cam = Camera.look_at([1.0 + k, -2.0, 0.5], [0, 0, 0], 50.0, H, W)
        P = K.dot(np.hstack([cam.R, cam.t]))
        np.savetxt(os.path.join(cal, "pos_%03d.txt" % (k + 1)), P)
        cams.append(cam)
        for illum in ("max", "3_r5000"):
            PILImage.fromarray(np.full((H, W, 3), 40 * (k + 1), np.uint8)).save(
                os.path.join(base, "Rectified/scan007", "rect_%03d_%s.png" % (k + 1, illum)))
        z = np.full((H, W), 2.0 + k, np.float32)
        z[0, 0] = 0.0                                     # no ground truth there
        np.save(os.path.join(base, "Depth/scan007", "depth_%03d.npy" % (k + 1)), z)
    PILImage.fromarray(np.zeros((H, W, 3), np.uint8)).save(
        os.path.join(base, "Rectified/scan007", "rect_050_max.png"))   # frame > 49: dropped
    savemat(os.path.join(base, "SampleSet/MVS_Data/ObsMask", "ObsMask7_10.mat"),
            {"BB": np.array([[-1.0, -2.0, -3.0], [1.0, 2.0, 3.0]]), "ObsMask": np.ones((2, 2, 2))})
    s = scene.get_scene("dtu", base, 7, illumination="max")
    assert s.n_images == 2 and s.image_shape == (H, W)
    assert np.array_equal(s.bbox, np.array([[-1, -2, -3, 1, 2, 3]], np.float32))
    assert s.observation_mask.shape == (2, 2, 2)
    for k in range(2):
        im = s.get_image(k)
        assert np.allclose(im.image, 40 * (k + 1) / 255.0)
        assert np.allclose(im.camera.R, cams[k].R, atol=1e-5)
        assert np.allclose(im.camera.t, cams[k].t, atol=1e-4)
        assert np.allclose(im.camera.center, cams[k].center, atol=1e-4)
        D = s.get_depth_map(k)
        assert D.shape == (H, W) and D.dtype == np.float32 and D[0, 0] == 0
        # a pixel's distance to the centre = z-depth * |K^-1 (u, v, 1)|
        u, v = 5, 2
        expect = (2.0 + k) * np.linalg.norm(np.linalg.inv(K).dot([u, v, 1.0]))
        assert abs(D[v, u] - expect) < 1e-4
