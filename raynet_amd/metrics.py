"""Evaluation metrics of the reference (raynet/metrics.py:130-236) on MI355X: the same
classes and `compute(scene, frame_idxs, depthmaps, predicted_pointcloud)` signature; the
nearest-neighbour distances come from the HIP scan behind `Pointcloud.nearest_neighbors`."""
import numpy as np

from .pointcloud import Pointcloud, PointcloudFromDepthMaps  # noqa: F401


class FiltersFactory(object):
    """raynet/metrics.py:11-24."""

    def __init__(self, filters):
        self.filters = filters

    @property
    def has_filters(self):
        return len(self.filters) > 0

    def filter(self, X):
        for f in self.filters:
            X = f.filter(X)
        return X


class Metric(object):
    def compute(self, scene, frame_idxs, depthmaps, predicted_pointcloud):
        raise NotImplementedError()


class PerPixelMeanDepthError(Metric):
    """raynet/metrics.py:135-152."""

    def __init__(self, borders=40):
        self.borders = borders

    def compute(self, scene, frame_idxs, depthmaps, predicted_pointcloud):
        """-> (mean |ground truth - prediction| per frame over the pixels that HAVE ground
        truth, None); a frame's border of `borders` pixels is left out."""
        H, W = scene.image_shape
        b = self.borders
        inner = (slice(b, H - b), slice(b, W - b))

        def frame_error(frame, prediction):
            truth = scene.get_depth_map(frame)[inner]
            if isinstance(prediction, str):
                prediction = np.load(prediction)
            known = truth != 0
            return np.abs(truth[known] - np.asarray(prediction)[inner][known]).mean()

        errors = [frame_error(f, d) for f, d in zip(frame_idxs, depthmaps)]
        return np.array(errors, dtype=np.float64).reshape(len(frame_idxs)), None


class _CloudMetric(Metric):
    def __init__(self, filter_factory=None, truncate=float("inf"), borders=40,
                 use_pc_from_depthmap=False):
        self.filter_factory = filter_factory if filter_factory is not None else FiltersFactory([])
        self.truncate = truncate
        self.borders = borders
        self.use_pc_from_depthmap = use_pc_from_depthmap

    def _clouds(self, scene, frame_idxs, predicted_pointcloud):
        if self.use_pc_from_depthmap:
            # ground-truth cloud from the ground-truth depth maps (metrics.py:170-181)
            gt = [scene.get_depthmap_file(i) for i in frame_idxs]
            ground_truth_pc = PointcloudFromDepthMaps(scene, frame_idxs, gt, self.borders)
        else:
            ground_truth_pc = scene.get_pointcloud()
        if self.filter_factory.has_filters:
            ground_truth_pc.filter(self.filter_factory)
            predicted_pointcloud.filter(self.filter_factory)
        return ground_truth_pc


class Accuracy(_CloudMetric):
    """raynet/metrics.py:155-195: distance of every predicted point to the ground truth."""

    def compute(self, scene, frame_idxs, depthmaps, predicted_pointcloud):
        ground_truth_pc = self._clouds(scene, frame_idxs, predicted_pointcloud)
        ground_truth_pc.index()
        distances, indexes = ground_truth_pc.nearest_neighbors(predicted_pointcloud.points)
        return np.minimum(distances, self.truncate), predicted_pointcloud.points


class Completeness(_CloudMetric):
    """raynet/metrics.py:198-236: distance of every ground-truth point to the prediction."""

    def compute(self, scene, frame_idxs, depthmaps, predicted_pointcloud):
        ground_truth_pc = self._clouds(scene, frame_idxs, predicted_pointcloud)
        predicted_pointcloud.index()
        distances, indexes = predicted_pointcloud.nearest_neighbors(ground_truth_pc.points)
        return np.minimum(distances, self.truncate), ground_truth_pc.points
