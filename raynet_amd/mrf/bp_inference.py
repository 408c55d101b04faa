"""BP inference plugin -- mirror of raynet/mrf/bp_inference.py:14-439.

`get_bp_backend("hip", generation_params, bp_iterations=3, batch_size=...)` returns
an object with the reference's three methods.  The reference's other names
("numpy", "tf", "cuda") are not provided: this package ships one backend and no
CPU fallback, and asking for them raises NotImplementedError.
"""
import numpy as np

from .mrf_hip import belief_propagation as hip_bp
from .mrf_hip import compute_depth_distribution as hip_compute_depth_distribution


class BPInference(object):
    def __init__(self, generation_params, bp_iterations=3, gamma_prior=0.05):
        self._generation_params = generation_params
        self.bp_iterations = bp_iterations   # number of bp updates
        self.gamma_prior = gamma_prior       # prior probability of a voxel being occupied

    def update_bp_messages(self, S, ray_voxel_indices, ray_voxel_count,
                           ray_to_occupancy_pon=None):
        """-> (ray_to_occupancy_accumulated_pon [gx,gy,gz], ray_to_occupancy_pon [N,M])"""
        raise NotImplementedError

    def estimate_depth_probabilities_from_messages(self, S, ray_voxel_indices, ray_voxel_count,
                                                   ray_to_occupancy_accumulated_pon,
                                                   ray_to_occupancy_pon, S_new):
        """-> S_new [N, M]"""
        raise NotImplementedError

    def mrf_inference(self, S, ray_voxel_indices, ray_voxel_count, ray_to_occupancy_pon=None,
                      S_new=None):
        # bp_inference.py:122-147
        acc, msgs = self.update_bp_messages(S, ray_voxel_indices, ray_voxel_count,
                                            ray_to_occupancy_pon)
        S_new = self.estimate_depth_probabilities_from_messages(
            S, ray_voxel_indices, ray_voxel_count, acc, msgs, S_new)
        return acc, msgs, S_new


class HIPBPInference(BPInference):
    """Counterpart of CUDABPInference (bp_inference.py:340-409)."""

    def __init__(self, generation_params, batch_size=1, bp_iterations=3, gamma_prior=0.05):
        super(HIPBPInference, self).__init__(generation_params, bp_iterations, gamma_prior)
        self.batch_size = batch_size

    def update_bp_messages(self, S, ray_voxel_indices, ray_voxel_count, ray_to_occupancy_pon):
        # bp_inference.py:362-373
        assert S.shape[0] == ray_voxel_indices.shape[0]
        assert S.shape[0] == ray_voxel_count.shape[0]
        assert S.shape[0] == ray_to_occupancy_pon.shape[0]
        assert S.shape[1] == ray_voxel_indices.shape[1]
        assert S.shape[1] == ray_to_occupancy_pon.shape[1]
        assert len(ray_voxel_count.shape) == 1
        assert np.int32 == ray_voxel_indices.dtype
        assert np.int32 == ray_voxel_count.dtype
        assert np.float32 == S.dtype
        assert np.float32 == ray_to_occupancy_pon.dtype
        return hip_bp(S, ray_voxel_indices, ray_voxel_count, ray_to_occupancy_pon,
                      self._generation_params.grid_shape, gamma=self.gamma_prior,
                      bp_iterations=self.bp_iterations, batch_size=self.batch_size)

    def estimate_depth_probabilities_from_messages(self, S, ray_voxel_indices, ray_voxel_count,
                                                   ray_to_occupancy_accumulated_pon,
                                                   ray_to_occupancy_pon, S_new):
        return hip_compute_depth_distribution(
            S, ray_voxel_indices, ray_voxel_count, ray_to_occupancy_pon,
            ray_to_occupancy_accumulated_pon, S_new, self._generation_params.grid_shape,
            self.batch_size)


def get_bp_backend(name, generation_params, **kwargs):
    # bp_inference.py:412-439
    bp_iterations = kwargs["bp_iterations"] if "bp_iterations" in kwargs.keys() else 3
    if name == "hip":
        if kwargs and "batch_size" in kwargs.keys():
            return HIPBPInference(generation_params, kwargs["batch_size"],
                                  bp_iterations=bp_iterations)
        raise ValueError("Missing argument for HIP backend")
    raise NotImplementedError(
        "backend %r: raynet_amd provides the 'hip' backend only (no CPU fallback)" % (name,))
