#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8(a) row a1 (`sample_in_bbox`) from the REFERENCE's own
NumPy sampling scheme.

Runs only in the build container (needs /root/reference).  What is run, and how:
  * raynet/common/camera.py, raynet/utils/{geometry,checks}.py -- Python-2 sources,
    converted with `python3 -m lib2to3` into a scratch directory under /tmp (never into
    this repo) and imported from there; raynet/utils/fast_utils.pyx (imported by
    geometry.py) is built with Cython + gcc from where it lies, into the same scratch
    directory;
  * `SamplingScheme` / `SamplingInBboxScheme` (raynet/common/sampling_schemes.py:10-186)
    and `Image.rays` (raynet/common/image.py:242-258): their modules cannot be imported
    here (the first imports TensorFlow through `..tf_implementations`, the second
    `imageio`; neither is installed, neither is used by the code below).  The class /
    method definitions are therefore cut out of the reference files with `ast` AT RUN TIME
    and executed unchanged in a namespace that holds what their bodies name (`np`,
    `product`, the reference's `project` and `ray_aabbox_intersection`).  No reference text
    is written anywhere.
What is called: `Image.rays()` (pixel enumeration u-major, `project(P_pinv, pixels)`) and
`SamplingInBboxScheme._sample_points_across_rays(center, rays - center, bbox.T)`, exactly as
`sample_points_across_rays` (:159-166) does, and the per-ray
`sample_points_across_ray` (:100-119, slab test of utils/geometry.py:77-147).

Cameras: the five first cameras of the reference's tests/restrepo_mock_dataset/scene_1
(K, R, t files, committed under tests/golden/restrepo_mock_scene_1) with that scene's
bounding box, and this repo's synthetic ring cameras (bench.py's scene).  Output:
tests/golden/ref_sampling_np.npz -- inputs (P, P_pinv, centre, bbox, image size, ray
indices) and the reference's outputs (points), nothing else.
"""
import ast
import importlib
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile
from itertools import product

import numpy as np

REF = os.environ.get("RAYNET_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def load_reference():
    scratch = tempfile.mkdtemp(prefix="raynet_ref_sampling_")
    pkg = os.path.join(scratch, "refsm")
    for d in ("", "utils", "common"):
        os.makedirs(os.path.join(pkg, d), exist_ok=True)
        open(os.path.join(pkg, d, "__init__.py"), "w").close()
    for rel in ("common/camera.py", "utils/geometry.py", "utils/checks.py"):
        shutil.copy(os.path.join(REF, "raynet", rel), os.path.join(pkg, rel))
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", pkg],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    pyx = os.path.join(REF, "raynet", "utils", "fast_utils.pyx")
    c_file = os.path.join(scratch, "fast_utils.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3", pyx, "-o", c_file],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    so = os.path.join(pkg, "utils", "fast_utils" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I", sysconfig.get_paths()["include"],
                           "-I", np.get_include(), c_file, "-o", so], stderr=subprocess.DEVNULL)
    sys.path.insert(0, scratch)
    sys.path.insert(0, os.path.join(pkg, "utils"))
    sys.path.insert(0, pkg)
    camera = importlib.import_module("refsm.common.camera")
    geometry = importlib.import_module("refsm.utils.geometry")

    def cut(path, wanted):
        """Definitions named in `wanted` (top-level classes, or `Class.method`) of a
        reference file, compiled from its own text."""
        src = open(os.path.join(REF, "raynet", path)).read()
        # the wanted bodies hold no Python-2-only syntax, other parts of the modules do
        # (print statements): parse definition by definition, by top-level line ranges
        lines = src.split("\n")
        out = []
        depth0 = [i for i, l in enumerate(lines) if l and not l[0].isspace() and not l.startswith("#")]
        for name in wanted:
            cls, _, meth = name.partition(".")
            start = next(i for i in depth0 if lines[i].startswith("class %s(" % cls))
            end = next((i for i in depth0 if i > start), len(lines))
            block = "\n".join(lines[start:end])
            tree = ast.parse(block)
            if meth:
                fn = next(n for n in tree.body[0].body
                          if isinstance(n, ast.FunctionDef) and n.name == meth)
                tree = ast.Module(body=[fn], type_ignores=[])
            out.append(tree)
        return out

    ns = dict(np=np, product=product, project=geometry.project,
              ray_aabbox_intersection=geometry.ray_aabbox_intersection, xrange=range)
    for tree in cut("common/sampling_schemes.py", ["SamplingScheme", "SamplingInBboxScheme"]):
        exec(compile(tree, "<raynet/common/sampling_schemes.py>", "exec"), ns)
    for tree in cut("common/image.py", ["Image.rays", "Image.ray"]):
        exec(compile(tree, "<raynet/common/image.py>", "exec"), ns)
    # utils/geometry.py:77 loops with xrange
    geometry.xrange = range
    return camera, geometry, ns, scratch


class _Holder(object):
    """What `Image.rays` / `Image.ray` read of an Image (common/image.py:210-258)."""

    def __init__(self, camera, H, W, ns):
        self.camera = self._camera = camera
        self.height, self.width = H, W
        self._ns = ns

    def rays(self):
        return self._ns["rays"](self)

    def ray(self, pixel):
        return self._ns["ray"](self, pixel)


class _Scene(object):
    def __init__(self, image, bbox):
        self._image, self.bbox = image, bbox

    def get_image(self, i):
        return self._image


class _GP(object):
    def __init__(self, D):
        self.sampling_type, self.depth_planes = "sample_in_bbox", D


def main():
    from raynet_amd.common.scene import parse_scene_info, read_krt
    from raynet_amd.synthetic import ring_cameras
    camera_mod, geometry, ns, scratch = load_reference()
    rng = np.random.default_rng(20180618)
    flat = {}
    try:
        cases = []
        base = os.path.join(HERE, "restrepo_mock_scene_1")
        bbox = np.asarray(parse_scene_info(os.path.join(base, "scene_info.xml")),
                          np.float32).reshape(1, 6)
        files = sorted(os.listdir(os.path.join(base, "cams_krt")))[:5]
        for k, f in enumerate(files):
            K, R, t = read_krt(os.path.join(base, "cams_krt", f))
            cases.append(("restrepo%d" % k, K, R, t, bbox, 72, 128, 16 if k % 2 else 32))
        # the reference's image size for the first camera
        K, R, t = read_krt(os.path.join(base, "cams_krt", files[0]))
        cases.append(("restrepo_full", K, R, t, bbox, 720, 1280, 32))
        ring_bbox = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
        for k, cam in enumerate(ring_cameras(5, 480, 640, focal=1.5 * 480)):
            cases.append(("ring%d" % k, cam.K, cam.R, cam.t, ring_bbox, 480, 640, 64))

        for name, K, R, t, bb, H, W, D in cases:
            cam = camera_mod.Camera(np.asarray(K, np.float64), np.asarray(R, np.float64),
                                    np.asarray(t, np.float64).reshape(3, 1))
            img = _Holder(cam, H, W, ns)
            scheme = ns["SamplingInBboxScheme"](_GP(D))
            # sampling_schemes.py:159-166, on a seeded subset of the rays (the per-ray list
            # comprehension of :144-152 takes ~20 us per ray)
            center, rays = img.rays()
            n = 250 if H * W > 250 else H * W
            ridx = np.sort(rng.choice(H * W, n, replace=False)).astype(np.int32)
            # always include the four corners and the centre pixel
            ridx[:5] = [0, H - 1, (W - 1) * H, W * H - 1, (W // 2) * H + H // 2]
            ridx = np.unique(ridx)
            directions = (rays - center)[:, ridx]
            pts = scheme._sample_points_across_rays(center, directions, bb.T)   # (4, n, D) f32
            assert pts.dtype == np.float32 and pts.shape == (4, len(ridx), D)
            # the per-ray entry point on a few of them (None when the ray misses the box)
            single = np.full((8, D, 4), np.nan, np.float32)
            hit = np.zeros((8,), np.int32)
            for j, r in enumerate(ridx[:8]):
                p = scheme.sample_points_across_ray(_Scene(img, bb), 0, int(r % H), int(r // H))
                if p is not None:
                    single[j] = p
                    hit[j] = 1
            flat[name + "/P"] = np.asarray(cam.P, np.float64)
            flat[name + "/P_pinv"] = np.asarray(cam.P_pinv, np.float64)
            flat[name + "/center"] = np.asarray(cam.center, np.float32).ravel()
            flat[name + "/bbox"] = bb.ravel()
            flat[name + "/HWD"] = np.array([H, W, D], np.int32)
            flat[name + "/ray_idxs"] = ridx
            assert np.all(pts[3] == 1.0)       # homogeneous coordinate: not stored
            flat[name + "/points"] = np.ascontiguousarray(pts[:3].transpose(1, 2, 0))   # (n, D, 3)
            flat[name + "/single_points"] = single[..., :3]
            flat[name + "/single_hit"] = hit
        out = os.path.join(HERE, "ref_sampling_np.npz")
        np.savez_compressed(out, **flat)
        print("wrote", out, os.path.getsize(out), "bytes;", len(cases), "cameras")
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
