"""raynet_amd -- RayNet's forward_pass hot path on MI355X (gfx950).

Only the path BASELINE.json names lives here: plane-sweep correlation, ray/voxel
traversal, planes->voxels mapping and the unrolled belief propagation, as
hand-written HIP kernels behind the reference's own factory API
(get_forward_pass_factory / get_bp_backend / perform_raynet_fp ...).  There is no
CPU fallback: every compute entry point needs libraynet_hip.so and a GPU.
"""
__version__ = "0.1.0"
