#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8(f) row 3 (depth maps -> point cloud -> accuracy /
completeness) from the REFERENCE's own code.

Runs only in the build container (needs /root/reference).  What is run, and how:
  * raynet/pointcloud.py, raynet/metrics.py, raynet/utils/{geometry,checks}.py -- Python-2
    sources, converted with `python3 -m lib2to3` into a scratch directory under /tmp (never
    into this repo) and imported from there;
  * raynet/utils/fast_utils.pyx (imported by geometry.py) -- built with Cython + gcc from
    where it lies, into the same scratch directory.
The scene handed to them is a stand-in with the attributes these classes use
(`get_image(i).camera/.width/.height/.rays()`, `get_depth_map`, `get_depthmap_file`,
`get_pointcloud`, `image_shape`); `rays()` is the reference's own `project(P_pinv, pixels)`
over its pixel enumeration (common/image.py:242-258, whose module needs `imageio`, which is
not installed).  Outputs: tests/golden/ref_pointcloud.npz -- inputs and the reference's
outputs, nothing else.
"""
import importlib
import os
import shutil
import subprocess
import sys
import tempfile
from itertools import product

import numpy as np

REF = os.environ.get("RAYNET_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def load_reference():
    scratch = tempfile.mkdtemp(prefix="raynet_ref_pc_")
    pkg = os.path.join(scratch, "refpc")
    os.makedirs(os.path.join(pkg, "utils"))
    for d in ("", "utils"):
        open(os.path.join(pkg, d, "__init__.py"), "w").close()
    for rel in ("pointcloud.py", "metrics.py", "utils/geometry.py", "utils/checks.py"):
        shutil.copy(os.path.join(REF, "raynet", rel), os.path.join(pkg, rel))
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", pkg],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # the Cython helper geometry.py imports, compiled from where it lies
    pyx = os.path.join(REF, "raynet", "utils", "fast_utils.pyx")
    c_file = os.path.join(scratch, "fast_utils.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3", pyx, "-o", c_file],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import sysconfig
    so = os.path.join(pkg, "utils", "fast_utils" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I", sysconfig.get_paths()["include"],
                           "-I", np.get_include(), c_file, "-o", so], stderr=subprocess.DEVNULL)
    sys.path.insert(0, scratch)
    sys.path.insert(0, os.path.join(pkg, "utils"))     # geometry.py's Python-2 implicit-relative
    sys.path.insert(0, pkg)                            # imports that lib2to3 left absolute
    pc = importlib.import_module("refpc.pointcloud")
    metrics = importlib.import_module("refpc.metrics")
    geometry = importlib.import_module("refpc.utils.geometry")
    return pc, metrics, geometry, scratch


class _Image(object):
    def __init__(self, camera, H, W, project):
        self.camera = camera
        self._camera = camera
        self.height, self.width = H, W
        self._project = project

    def rays(self):          # common/image.py:242-258
        pixels = np.array([[u, v, 1.] for u, v in product(range(self.width), range(self.height))],
                          dtype=np.int32).T
        rays = self._project(self.camera.P_pinv, pixels)
        return self._camera.center, rays.T


class _Scene(object):
    def __init__(self, images, gt_maps, gt_files, gt_cloud, pc_mod):
        self._images, self._gt, self._gt_files = images, gt_maps, gt_files
        self._cloud, self._pc = gt_cloud, pc_mod
        self.image_shape = (images[0].height, images[0].width)

    def get_image(self, i):
        return self._images[i]

    def get_depth_map(self, i):
        return self._gt[i]

    def get_depthmap_file(self, i):
        return self._gt_files[i]

    def get_pointcloud(self):
        return self._pc.Pointcloud(self._cloud.copy())


def sphere_depth(cam, H, W, centre, radius, far):
    """Distance along every pixel's ray to a sphere (far where it misses); [H, W], float32."""
    c = cam.center.ravel()[:3]
    pix = np.array([[u, v, 1.] for u, v in product(range(W), range(H))], np.float64).T
    r = cam.P_pinv.dot(pix)
    r = (r / r[-1:])[:3].T - c
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    oc = c - centre
    b = r.dot(oc)
    disc = b * b - (oc.dot(oc) - radius * radius)
    t = np.where(disc > 0, -b - np.sqrt(np.maximum(disc, 0)), far)
    return t.reshape(W, H).T.astype(np.float32)


def main():
    from raynet_amd.synthetic import ring_cameras
    pc, metrics, geometry, scratch = load_reference()
    rng = np.random.default_rng(11)
    H, W, n_frames, borders = 20, 28, 4, 3
    cams = ring_cameras(n_frames, H, W, focal=1.5 * H, arc=np.pi / 2)
    images = [_Image(c, H, W, geometry.project) for c in cams]
    centre, radius = np.array([0.0, 0.0, -0.1]), 0.5
    gt = [sphere_depth(c, H, W, centre, radius, 0.0) for c in cams]          # 0 = no ground truth
    pred = []
    for g in gt:
        p = np.where(g > 0, g, 3.0).astype(np.float32)
        p += (rng.standard_normal(p.shape) * 0.02).astype(np.float32)
        p[rng.random(p.shape) < 0.08] += 0.9                                  # outliers
        pred.append(p)
    pred[1][5, 7] = np.nan            # the reference replaces NaNs by the map's minimum
    tmp = tempfile.mkdtemp(prefix="raynet_pc_maps_")
    pred_files, gt_files = [], []
    for i in range(n_frames):
        pf, gf = os.path.join(tmp, "depth_%03d.npy" % i), os.path.join(tmp, "gt_depth_%d.npy" % i)
        np.save(pf, pred[i])
        np.save(gf, gt[i])
        pred_files.append(pf)
        gt_files.append(gf)
    gt_cloud = centre[:, None] + radius * (lambda v: v / np.linalg.norm(v, axis=0))(
        rng.standard_normal((3, 900)))
    scene = _Scene(images, gt, gt_files, gt_cloud.astype(np.float32), pc)
    frames = list(range(n_frames))

    plain = pc.PointcloudFromDepthMaps(scene, frames, pred_files, borders)
    cons = pc.PointcloudFromDepthMapsWithConsistency(scene, frames, pred_files, borders,
                                                     consistency_threshold=0.25, n_neighbors=2)
    ff = metrics.FiltersFactory([])
    out = dict(H=H, W=W, borders=borders, consistency_threshold=0.25, n_neighbors=2,
               P=np.array([c.P for c in cams]), P_pinv=np.array([c.P_pinv for c in cams]),
               center=np.array([c.center for c in cams]), gt=np.array(gt), pred=np.array(pred),
               gt_cloud=gt_cloud.astype(np.float32),
               points_plain=plain.points, points_consistency=cons.points)
    out["per_pixel_error"] = metrics.PerPixelMeanDepthError(borders).compute(
        scene, frames, pred_files, None)[0]
    for name, cloud in (("plain", plain), ("consistency", cons)):
        for use_dm in (False, True):
            tag = "%s_%s" % (name, "dm" if use_dm else "pc")
            acc = metrics.Accuracy(ff, truncate=0.3, borders=borders, use_pc_from_depthmap=use_dm)
            d, pts = acc.compute(scene, frames, pred_files, pc.Pointcloud(cloud.points.copy()))
            out["accuracy_" + tag] = d.ravel()
            comp = metrics.Completeness(ff, truncate=0.3, borders=borders,
                                        use_pc_from_depthmap=use_dm)
            d, pts = comp.compute(scene, frames, pred_files, pc.Pointcloud(cloud.points.copy()))
            out["completeness_" + tag] = d.ravel()
            out["completeness_points_" + tag] = np.asarray(pts)
    np.savez_compressed(os.path.join(HERE, "ref_pointcloud.npz"), **out)
    for k, v in sorted(out.items()):
        print(k, getattr(v, "shape", v))
    shutil.rmtree(scratch, ignore_errors=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
