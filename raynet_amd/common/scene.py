"""Scene model: what the forward-pass drivers and the evaluation read.

`Scene` (in-memory images + cameras + bbox) carries the array conventions of the reference's
raynet/common/scene.py:22-141; `RestrepoScene` and `DTUScene` read the two on-disk dataset
layouts the reference supports (scene.py:144-452, parse_input_data.py:13-58), SURVEY.md 8(f)
row 4.  Host-side parsing only -- nothing here touches the GPU.
"""
import functools
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
        self.image = image          # (H, W, C) float array
        self.camera = camera
        self.height, self.width = image.shape[:2]

    @classmethod
    def from_file(cls, image_file, camera_poses):
        """common/image.py:23-47: pixels scaled to [0, 1] float32, camera from K, R, t."""
        from PIL import Image as PILImage
        data = np.asarray(PILImage.open(image_file))
        if data.ndim == 2:
            data = data[:, :, np.newaxis]
        data = data.astype(np.float32) / np.float32(255.)
        return cls(data, Camera(K=camera_poses["K"], R=camera_poses["R"], t=camera_poses["t"]))


def get_adjacent_frames_idxs(ref_idx, n_frames, n_adjacent, skip):
    """Neighbour views of `ref_idx` under the assumption that consecutive frames are close in
    space: the rule of raynet/utils/training_utils.py:9-60 (a window around the reference,
    pushed inwards at the ends of the sequence), including its unsigned index arithmetic.
    Checked case by case against the reference's own function
    (tests/golden/ref_adjacent_frames.json)."""
    if ref_idx > n_frames:
        raise ValueError("Ref index needs to be smaller than n_frames")
    step = skip + 1
    median = np.floor(n_adjacent / 2.0)
    if n_adjacent % 2 == 0:
        min_idx = max(0, ref_idx - median * step)
    else:
        min_idx = max(0, ref_idx - median * step - 1)
    max_idx = min(n_frames, ref_idx + median * step + 1)
    idxs = np.append(np.arange(min_idx, ref_idx, step=step, dtype=np.uint32),
                     np.arange(ref_idx + 1, max_idx, step=step, dtype=np.uint32))
    if len(idxs) != n_adjacent:
        if ref_idx == 0:
            idxs = np.arange(step, (n_adjacent + 1) * step, step=step)
        elif ref_idx == n_frames - 1:
            idxs = np.arange(ref_idx - n_adjacent * step, ref_idx, step=step)
        elif max(idxs) == n_frames - 1:
            for _ in range(n_adjacent - len(idxs)):
                idxs = np.insert(idxs, 0, min(idxs) - step)
        elif min(idxs) == 0:
            for _ in range(n_adjacent - len(idxs)):
                idxs = np.append(idxs, max(idxs) + step)
    return idxs


@functools.lru_cache(maxsize=4096)
def _adjacent_cached(i, n_images, neighbors):
    # a pure function of three integers, asked for every reference image of every pass
    return tuple(int(j) for j in get_adjacent_frames_idxs(i, n_images, neighbors, 0))


def adjacent_views(i, n_images, neighbors):
    """The 'filesystem' neighbour rule of the reference (common/scene.py:41-57) as a list."""
    assert neighbors < n_images
    return [int(j) for j in get_adjacent_frames_idxs(i, n_images, neighbors, 0)]


class Scene(object):
    """In-memory scene; the file-backed subclasses override n_images / get_image / bbox.
    select_neighbors_based_on: "filesystem" (consecutive frames are neighbours) or
    "distance" (nearest camera centres), scene.py:41-57."""

    def __init__(self, images=None, bbox=None, select_neighbors_based_on="filesystem"):
        self._images = images
        self._bbox = None if bbox is None else np.asarray(bbox, dtype=np.float32).reshape(1, 6)
        self._voxel_grid = None
        self._camera_neighbors = None
        self._select_neighbors_based_on = select_neighbors_based_on

    bbox = property(lambda self: self._bbox)
    n_images = property(lambda self: len(self._images))

    @staticmethod
    def _load_sorted_files(basepath, dir, condition=None):
        path = os.path.join(basepath, dir)
        return [os.path.join(path, f) for f in sorted(filter(condition, os.listdir(path)))]

    def _get_neighbor_idxs(self, i, neighbors):
        if self._select_neighbors_based_on == "distance":
            if self._camera_neighbors is None:       # scene.py:59-79
                a = np.hstack([self.get_image(k).camera.center for k in range(self.n_images)])
                distances = ((a.T[:, :, np.newaxis] - a[np.newaxis]) ** 2).sum(axis=1)
                self._camera_neighbors = distances.argsort()[:, 1:neighbors + 1]
            return [int(j) for j in self._camera_neighbors[i]]
        if self._select_neighbors_based_on == "filesystem":
            return list(_adjacent_cached(int(i), int(self.n_images), int(neighbors)))
        raise NotImplementedError()

    @property
    def observation_mask(self):
        return None

    @property
    def gt_depth_range(self):
        D = self.get_depth_map(0)
        return np.min(D[D != 0]), np.max(D)

    def get_images(self):
        return [self.get_image(i) for i in range(self.n_images)]

    def get_depth_map(self, i):
        raise NotImplementedError()

    def get_depthmap_file(self, i):
        return None

    def get_pointcloud(self):
        raise NotImplementedError()

    @property
    def image_shape(self):
        im = self.get_image(0)
        return im.height, im.width

    def get_image(self, i):
        return self._images[i]

    def get_image_with_neighbors(self, i, neighbors=4):
        # reference first, then its neighbours (scene.py:110-115)
        return [self.get_image(i)] + [self.get_image(n)
                                      for n in self._get_neighbor_idxs(i, neighbors)]

    def view_indices_with_neighbors(self, i, neighbors=4):
        return [i] + self._get_neighbor_idxs(i, neighbors)

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


class RestrepoScene(Scene):
    """A scene in the layout of Restrepo et al. (scene.py:144-254): imgs/, cams_krt/ (K, R, t
    blocks), scene_info.xml (bbox), optionally gt/gt_depth_%d.npy."""

    def __init__(self, basepath, select_neighbors_based_on="filesystem"):
        super(RestrepoScene, self).__init__(select_neighbors_based_on=select_neighbors_based_on)
        self._basepath = basepath
        self._image_paths = self._load_sorted_files(basepath, "imgs")
        self._cam_paths = self._load_sorted_files(basepath, "cams_krt")
        self._bbox_path = os.path.join(basepath, "scene_info.xml")
        self._cache = [None] * len(self._image_paths)

    n_images = property(lambda self: len(self._image_paths))

    @property
    def bbox(self):
        if self._bbox is None:
            self._bbox = parse_scene_info(self._bbox_path)
        return self._bbox

    def _read_camera_poses(self, i):
        K, R, t = read_krt(self._cam_paths[i])
        return {"K": K, "R": R, "t": t}

    def get_image(self, i):
        if self._cache[i] is None:
            self._cache[i] = Image.from_file(self._image_paths[i], self._read_camera_poses(i))
        return self._cache[i]

    def get_depthmap_file(self, i):
        f = os.path.join(self._basepath, "gt", "gt_depth_%d.npy" % (i,))
        return f if os.path.isfile(f) else None

    def get_depth_map(self, i):
        f = self.get_depthmap_file(i)
        if f is None:
            # the reference ray-casts the ground-truth meshes through an octree here
            # (scene.py:187-201); that training-data machinery is not part of this package
            raise NotImplementedError("no gt/gt_depth_%d.npy in %s" % (i, self._basepath))
        return np.load(f)


def parse_scene_info_dtu_dataset(scene_file):
    """bbox (1, 6) float32 from a DTU ObsMask .mat file (parse_input_data.py:42-58)."""
    from scipy.io import loadmat
    return loadmat(scene_file, squeeze_me=True)["BB"].astype(np.float32).reshape(1, -1)


class DTUScene(Scene):
    """A scan of the DTU MVS dataset (scene.py:257-452): Rectified/scanNNN images of one
    illumination, SampleSet/MVS_Data/Calibration/cal18 {intrinsic.txt, pos_*.txt},
    ObsMask%d_10.mat (bbox), Depth/scanNNN/*.npy (z-depth maps, converted to distances from
    the camera centre like the reference does)."""

    def __init__(self, basepath, scene_idx, illumination="max",
                 select_neighbors_based_on="filesystem"):
        super(DTUScene, self).__init__(select_neighbors_based_on=select_neighbors_based_on)
        self._basepath = basepath
        paths = self._load_sorted_files(basepath, os.path.join("Rectified", "scan%03d" % scene_idx),
                                        lambda f: illumination in f)
        # only frames 1..49 have depth maps (scene.py:277-285)
        self._image_paths = [
            ip for ip in paths
            if int(os.path.basename(ip).split(".")[0].split("_")[1]) <= 49]
        cal = "SampleSet/MVS_Data/Calibration/cal18"
        self._cam_paths = self._load_sorted_files(basepath, cal, lambda f: "pos" in f)
        self._cam_intrinsic_path = os.path.join(basepath, cal, "intrinsic.txt")
        self._bbox_path = os.path.join(basepath, "SampleSet/MVS_Data/ObsMask",
                                       "ObsMask%d_10.mat" % scene_idx)
        depth_dir = os.path.join("Depth", "scan%03d" % scene_idx)
        self._depth_map_paths = (
            self._load_sorted_files(basepath, depth_dir, lambda f: f.endswith("npy"))
            if os.path.isdir(os.path.join(basepath, depth_dir)) else [])
        self._cache = [None] * len(self._image_paths)
        self._cache_depth_maps = [None] * len(self._image_paths)

    n_images = property(lambda self: len(self._image_paths))

    @property
    def bbox(self):
        if self._bbox is None:
            self._bbox = parse_scene_info_dtu_dataset(self._bbox_path).astype(np.float32)
        return self._bbox

    @property
    def observation_mask(self):
        from scipy.io import loadmat
        return loadmat(self._bbox_path)["ObsMask"]

    def _read_camera_poses(self, i):
        """scene.py:329-365: K from intrinsic.txt, [R t] = K^-1 P from pos_XXX.txt."""
        with open(self._cam_intrinsic_path) as f:
            rows = [ln.strip().split(" ") for ln in f.readlines()]
        K = np.array(rows[0:3]).astype(np.float32)
        with open(self._cam_paths[i]) as f:
            rows = [ln.strip().split(" ") for ln in f.readlines()]
        P = np.array(rows[0:4]).astype(np.float32)
        Rt = np.dot(np.linalg.inv(K), P)
        return {"K": K, "R": Rt[:, :3], "t": Rt[:, -1].reshape(-1, 1)}

    def get_image(self, i):
        if self._cache[i] is None:
            self._cache[i] = Image.from_file(self._image_paths[i], self._read_camera_poses(i))
        return self._cache[i]

    def get_gt_depth_map(self, i):
        return np.load(self._depth_map_paths[i])

    def get_depth_map(self, i):
        """scene.py:372-407: per-pixel distance to the camera centre from the z-depth map."""
        if self._cache_depth_maps[i] is None:
            image = self.get_image(i)
            gt = self.get_gt_depth_map(i)
            H, W = image.height, image.width
            pixels = np.array([[u, v, 1.] for u in range(W) for v in range(H)], dtype=np.float32).T
            p_cc = np.dot(np.linalg.inv(image.camera.K), pixels) * gt.T.reshape(1, -1)
            p_cc = np.vstack([p_cc, np.ones(p_cc.shape[1], dtype=np.float32)])
            P = np.vstack([np.hstack([image.camera.R, image.camera.t]),
                           np.array([0., 0., 0., 1.])])
            target = np.dot(np.linalg.inv(P), p_cc).T
            target = target / target[:, -1:]
            D = np.sqrt(((target - image.camera.center.T) ** 2).sum(axis=-1)).reshape(W, H).T
            D = D * (gt != 0)
            self._cache_depth_maps[i] = D.astype(np.float32)
        return self._cache_depth_maps[i]


def get_scene(dataset_type, basepath, *args, **kwargs):
    """"restrepo" / "dtu" -> scene object (the switch of common/dataset.py:8-103)."""
    if dataset_type == "restrepo":
        return RestrepoScene(basepath, *args, **kwargs)
    if dataset_type == "dtu":
        return DTUScene(basepath, *args, **kwargs)
    raise NotImplementedError(dataset_type)
