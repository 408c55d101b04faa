#!/usr/bin/env python3
"""The caller of the hot path: `raynet_forward` of the reference (raynet/scripts/forward_pass.py:29-146)
with the flags that reach the path -- same names, same defaults -- on MI355X.

    python -m raynet_amd.scripts.forward_pass DATASET_DIR OUT_DIR --dataset_type restrepo \\
        --forward_pass_factory raynet --depth_planes 32 --grid_shape 64,64,32 ...

Loads a scene (Restrepo or DTU layout), builds the MV-CNN twin (random weights unless
--weight_file is given: a `.npz` holding the reference's weight list in the reference's own
order, raynet/models.py:329-339 -- `numpy.savez(path, *model.get_weights())` on the Keras
side; HDF5 itself cannot be read in this image -- or a torch state_dict of the twin), runs `forward_pass(scene, (start, end, skip_every + 1))` and writes one `depth_%03d.npy`
((H, W) float32) per reference image, the wire format of scripts/forward_pass.py:136-142.
"""
import argparse
import os
import sys

import numpy as np


def _ints(x):
    return tuple(map(int, x.split(",")))


def build_parser():
    p = argparse.ArgumentParser(description=("Do a forward pass and estimate the per pixel depth "
                                             "for the images of a scene"))
    p.add_argument("dataset_directory", help="Directory containing the input data")
    p.add_argument("output_directory", help="Directory to save the output data")
    p.add_argument("--weight_file", help="MV-CNN weights: .npz in the reference's weight order "
                                         "(models.py:329-339) or a torch state_dict of the twin")
    p.add_argument("--schedule", choices=["resident", "reference"], default="resident",
                   help="raynet factory: per-ray columns resident in HBM (grouped by the free "
                        "memory) / the reference's literal recompute-everything schedule")
    p.add_argument("--scene_idx", default=1, type=int, help="DTU: the scan number")
    p.add_argument("--filter_out", action="store_true", help="Filter out rays with zero ground-truth")
    # scripts/arguments.py:146-223 (generation)
    p.add_argument("--patch_shape", type=_ints, default="11,11,3")
    p.add_argument("--padding", default=None, type=int)
    p.add_argument("--depth_planes", type=int, default=32)
    p.add_argument("--neighbors", type=int, default=4)
    p.add_argument("--grid_shape", type=_ints, default="256,256,128")
    p.add_argument("--maximum_number_of_marched_voxels", type=int, default=650)
    # :302-329 (dataset)
    p.add_argument("--select_neighbors_based_on", choices=["filesystem", "distance"],
                   default="filesystem")
    p.add_argument("--illumination_condition", default="max")
    p.add_argument("--dataset_type", choices=["restrepo", "dtu"], default="restrepo")
    # :335-358 (mrf, indexing)
    p.add_argument("--initial_gamma_prior", type=float, default=0.05)
    p.add_argument("--bp_iterations", type=int, default=3)
    p.add_argument("--start_end", type=_ints, default="0,5")
    p.add_argument("--skip_every", type=int, default=0)
    # :362-379 (forward pass factory)
    p.add_argument("--forward_pass_factory",
                   choices=["multi_view_cnn", "multi_view_cnn_voxel_space", "raynet"],
                   default="raynet")
    p.add_argument("--rays_batch", type=int, default=130000)
    p.add_argument("--network_architecture", choices=["simple_cnn"], default="simple_cnn")
    return p


def load_model(weight_file=None, architecture="simple_cnn", in_channels=3, device="cuda"):
    """The MV-CNN twin with the weights of `weight_file` (see --weight_file)."""
    import torch
    from raynet_amd.models import get_nn
    model = get_nn(architecture)(in_channels=in_channels).to(device)
    if weight_file:
        if str(weight_file).endswith(".npz"):
            model.load_reference_weights(weight_file)
        else:
            model.load_state_dict(torch.load(weight_file, map_location=device))
    return model


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.common.scene import get_scene
    from raynet_amd.forward_pass import get_forward_pass_factory

    if not os.path.exists(args.output_directory):
        os.makedirs(args.output_directory)
    if isinstance(args.patch_shape, str):
        args.patch_shape = _ints(args.patch_shape)
    if isinstance(args.grid_shape, str):
        args.grid_shape = _ints(args.grid_shape)
    if isinstance(args.start_end, str):
        args.start_end = _ints(args.start_end)
    args.grid_shape = np.array(args.grid_shape, dtype=np.int32)
    generation_params = GenerationParameters.from_options(args)

    if args.dataset_type == "dtu":
        scene = get_scene("dtu", args.dataset_directory, args.scene_idx,
                          illumination=args.illumination_condition,
                          select_neighbors_based_on=args.select_neighbors_based_on)
    else:
        scene = get_scene("restrepo", args.dataset_directory,
                          select_neighbors_based_on=args.select_neighbors_based_on)

    model = load_model(args.weight_file, args.network_architecture,
                       in_channels=scene.get_image(0).image.shape[2])

    cls = get_forward_pass_factory(args.forward_pass_factory)
    kwargs = dict(filter_out_rays=args.filter_out)
    if args.forward_pass_factory == "raynet":
        kwargs["bp_iterations"] = args.bp_iterations
        kwargs["schedule"] = args.schedule
    fp = cls(model, generation_params, "sample_in_bbox", scene.image_shape, args.rays_batch, **kwargs)

    start, end = args.start_end
    ref_idx = start
    for S in fp.forward_pass(scene, (start, end, args.skip_every + 1)):
        # (with a process group the rays are sharded and image k's map is assembled by ONE rank,
        # forward_pass.map_owner: the other ranks get None for it and leave its file alone)
        if S is not None:
            np.save(os.path.join(args.output_directory, "depth_%03d.npy" % (ref_idx,)), S)
        ref_idx += args.skip_every + 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
