"""GenerationParameters -- the one config struct the hot path reads.

API mirror of raynet/common/generation_parameters.py:31-118: the same constructor keywords
and defaults, the same attribute names, `from_options(namespace)` for argparse callers.  The
training-only `target_distribution_factory` is carried as an opaque value and never used here.
"""
import numpy as np

# sampling policy name -> the sampling routine's name (generation_parameters.py:20-28);
# the first key contained in the policy name wins
_SAMPLING_TYPES = (("bbox", "sample_points_in_bbox"), ("range", "sample_points_in_range"),
                   ("disparity", "sample_points_in_disparity"),
                   ("voxel_space", "sample_points_in_voxel_space"))

# constructor keywords in the reference's order, with its defaults
_FIELDS = (("depth_planes", 32), ("neighbors", 4), ("patch_shape", (11, 11, 3)),
           ("grid_shape", None), ("max_number_of_marched_voxels", 400), ("expand_patch", True),
           ("target_distribution_factory", None), ("depth_range", None), ("step_depth", None),
           ("padding", None), ("sampling_type", None), ("gamma_mrf", None))

# Namespace attribute -> constructor keyword, where the two differ
_OPTION_NAMES = {"max_number_of_marched_voxels": "maximum_number_of_marched_voxels",
                 "gamma_mrf": "initial_gamma_prior"}


def get_sampling_type(name):
    for key, routine in _SAMPLING_TYPES:
        if key in name:
            return routine
    return None


class GenerationParameters(object):
    def __init__(self, *args, **kwargs):
        names = [f for f, _ in _FIELDS]
        if len(args) > len(names):
            raise TypeError("GenerationParameters takes at most %d arguments" % len(names))
        given = dict(zip(names, args))
        for k, v in kwargs.items():
            if k not in names:
                raise TypeError("unexpected keyword argument %r" % (k,))
            if k in given:
                raise TypeError("multiple values for argument %r" % (k,))
            given[k] = v
        for name, default in _FIELDS:
            setattr(self, name, given.get(name, default))
        if "grid_shape" not in given:       # (a fresh array per object, not a shared default)
            self.grid_shape = np.array([64, 64, 32], dtype=np.int32)

    @classmethod
    def from_options(cls, argument_parser):
        """From an argparse Namespace (generation_parameters.py:64-118): options the parser
        does not have become None; padding defaults to the patch width."""
        opts = vars(argument_parser)
        patch_shape = opts.get("patch_shape", (None,) * 3)
        values = {name: opts.get(_OPTION_NAMES.get(name, name)) for name, _ in _FIELDS}
        values["patch_shape"] = patch_shape
        values["padding"] = opts["padding"] if opts.get("padding") is not None else patch_shape[0]
        policy = opts.get("sampling_policy")
        values["sampling_type"] = get_sampling_type(policy) if policy is not None else None
        values["target_distribution_factory"] = None
        del values["expand_patch"]          # not an option: keeps the constructor's default
        return cls(**values)
