"""Minimal scene model: just what the forward-pass drivers read.

The reference's raynet/common package (scene/dataset/image parsers, 1.7 kLoC) is
out of scope; this is the ~100-line shim of its array
conventions: Scene.{bbox, image_shape, n_images, get_image, get_image_with_neighbors,
voxel_grid} and Image.{image, camera}.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from .camera import Camera


def get_voxel_grid(bbox, grid_shape):
    """Voxel centres [3, gx, gy, gz] float32 (raynet/utils/generic_utils.py:90-110)."""
    bbox = np.asarray(bbox, dtype=np.float32).reshape(1, 6)
    xyz = [np.linspace(s, e, int(c), endpoint=False, dtype=np.float32)
           for s, e, c in zip(bbox[0, :3], bbox[0, 3:], grid_shape)]
    bin_size = np.array(
        [(a[1] - a[0]) if len(a) > 1 else np.float32(e - s)
         for a, s, e in zip(xyz, bbox[0, :3], bbox[0, 3:])], dtype=np.float32).reshape(3, 1, 1, 1)
    return (np.stack(np.meshgrid(*xyz, indexing="ij")) + bin_size / 2).astype(np.float32)


class Image(object):
    def __init__(self, image, camera):
        self.image = image          # (H, W, C) float array, may be None for feature-only scenes
        self.camera = camera
        self.height, self.width = image.shape[:2]


def adjacent_views(i, n_images, neighbors):
    """Indices of the `neighbors` views closest in index to i (a window centred on i,
    shifted inwards at the borders) -- the 'filesystem' rule of the reference
    (raynet/common/scene.py:41-57, utils/training_utils.py:9-60)."""
    assert neighbors < n_images
    lo = i - neighbors // 2
    lo = max(0, min(lo, n_images - 1 - neighbors))
    return [j for j in range(lo, lo + neighbors + 1) if j != i][:neighbors]


class Scene(object):
    def __init__(self, images, bbox):
        self._images = images
        self._bbox = np.asarray(bbox, dtype=np.float32).reshape(1, 6)
        self._voxel_grid = None

    bbox = property(lambda self: self._bbox)
    n_images = property(lambda self: len(self._images))

    @property
    def image_shape(self):
        im = self.get_image(0)
        return im.height, im.width

    def get_image(self, i):
        return self._images[i]

    def get_image_with_neighbors(self, i, neighbors=4):
        # reference first, then its neighbours (scene.py:110-115)
        return [self.get_image(i)] + [self.get_image(n)
                                      for n in adjacent_views(i, self.n_images, neighbors)]

    def view_indices_with_neighbors(self, i, neighbors=4):
        return [i] + adjacent_views(i, self.n_images, neighbors)

    def voxel_grid(self, grid_shape):
        if self._voxel_grid is None:
            self._voxel_grid = get_voxel_grid(self.bbox, grid_shape)
        return self._voxel_grid.astype(np.float32)


def parse_scene_info(path):
    """bbox (1, 6) from a Restrepo scene_info.xml (common/parse_input_data.py:13-39)."""
    bbox = ET.parse(path).getroot().find("bbox").attrib
    return np.array([[bbox["minx"], bbox["miny"], bbox["minz"]],
                     [bbox["maxx"], bbox["maxy"], bbox["maxz"]]], dtype=np.float32).reshape(1, -1)


def read_krt(path):
    """K (3x3), R (3x3), t (3x1) from a Restrepo cams_krt file (common/scene.py:219-242)."""
    with open(path) as f:
        rows = [ln.split() for ln in f if ln.strip()]
    K = np.array(rows[0:3]).astype(np.float32)
    R = np.array(rows[3:-1]).astype(np.float32)
    t = np.array(rows[-1]).astype(np.float32).reshape(-1, 1)
    return K, R, t


def restrepo_cameras_scene(basepath, image_shape, n_images=None, channels=3, seed=0,
                           scale=1.0):
    """Scene with the cameras + bbox of a Restrepo directory (cams_krt/*.txt,
    scene_info.xml) and synthetic noise images of `image_shape`; `scale` rescales the
    intrinsics when the images are smaller than the original 1280x720."""
    cams = sorted(os.listdir(os.path.join(basepath, "cams_krt")))
    if n_images is not None:
        cams = cams[:n_images]
    rng = np.random.default_rng(seed)
    images = []
    for c in cams:
        K, R, t = read_krt(os.path.join(basepath, "cams_krt", c))
        K = K.copy()
        K[:2] *= scale
        img = rng.random((image_shape[0], image_shape[1], channels)).astype(np.float32)
        images.append(Image(img, Camera(K, R, t)))
    return Scene(images, parse_scene_info(os.path.join(basepath, "scene_info.xml")))
