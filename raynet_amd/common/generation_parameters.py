"""GenerationParameters -- the one config struct the hot path reads.

Mirror of raynet/common/generation_parameters.py:31-118 (same field names and
defaults; the training-only target_distribution_factory is carried but unused).
"""
import numpy as np


def get_sampling_type(name):
    # generation_parameters.py:20-28
    if "bbox" in name:
        return "sample_points_in_bbox"
    elif "range" in name:
        return "sample_points_in_range"
    elif "disparity" in name:
        return "sample_points_in_disparity"
    elif "voxel_space" in name:
        return "sample_points_in_voxel_space"


class GenerationParameters(object):
    def __init__(self, depth_planes=32, neighbors=4, patch_shape=(11, 11, 3),
                 grid_shape=np.array([64, 64, 32], dtype=np.int32),
                 max_number_of_marched_voxels=400, expand_patch=True,
                 target_distribution_factory=None, depth_range=None, step_depth=None,
                 padding=None, sampling_type=None, gamma_mrf=None):
        self.neighbors = neighbors
        self.patch_shape = patch_shape
        self.expand_patch = expand_patch
        self.depth_planes = depth_planes
        self.grid_shape = grid_shape
        self.depth_range = depth_range
        self.step_depth = step_depth
        self.padding = padding
        self.sampling_type = sampling_type
        self.target_distribution_factory = target_distribution_factory
        self.max_number_of_marched_voxels = max_number_of_marched_voxels
        self.gamma_mrf = gamma_mrf

    @classmethod
    def from_options(cls, argument_parser):
        """generation_parameters.py:64-118: build from an argparse Namespace."""
        args = vars(argument_parser)
        patch_shape = args["patch_shape"] if "patch_shape" in args else (None,) * 3
        padding = args["padding"] if args.get("padding") is not None else patch_shape[0]
        try:
            sampling_type = get_sampling_type(argument_parser.sampling_policy)
        except AttributeError:
            sampling_type = None
        return cls(
            patch_shape=patch_shape,
            depth_planes=args.get("depth_planes"),
            neighbors=args.get("neighbors"),
            target_distribution_factory=None,
            grid_shape=args.get("grid_shape"),
            max_number_of_marched_voxels=args.get("maximum_number_of_marched_voxels"),
            depth_range=args.get("depth_range"),
            step_depth=args.get("step_depth"),
            padding=padding,
            sampling_type=sampling_type,
            gamma_mrf=args.get("initial_gamma_prior"),
        )
