"""Loader of libraynet_hip.so (the C ABI declared in include/raynet_hip.h).

The product path fails loudly when the HIP library is missing or no GPU is
visible -- there is deliberately no CPU route in this package.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# RAYNET_HIP_LIB: another BUILD of the same library (a variant from build(extra_flags=..., out=...));
# its rn_version() says what it was built with
ENV_LIB = os.environ.get("RAYNET_HIP_LIB") or None     # read ONCE: the path and every test on it
LIB_PATH = ENV_LIB or os.path.join(CSRC, "libraynet_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "raynet_hip.h")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17",
    "-ffp-contract=off",        # index maps bit-exact w.r.t. the reference arithmetic
    "-munsafe-fp-atomics",      # hardware global_atomic_add_f32 for the accumulator scatter
    "-Wno-unused-value", "-fPIC", "-shared",
]

RN_OK = 0
STATUS = {0: "RN_OK", -1: "RN_ERR_INVALID", -2: "RN_ERR_HIP", -3: "RN_ERR_NO_DEVICE",
          -4: "RN_ERR_STATE"}


class RaynetHipError(RuntimeError):
    pass


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU).
    extra_flags / out: a VARIANT build (A/B runs, the exact-arithmetic build of the tests) into
    another path; the flags end up in rn_version() (RN_BUILD_EXTRA), so a variant can never
    pass for the shipped library (tests/test_abi.py)."""
    srcs = [os.path.join(CSRC, "raynet_hip.hip"), os.path.join(CSRC, "raynet_kernels.h"),
            os.path.join(CSRC, "raynet_prepare.inl"), os.path.join(CSRC, "raynet_mrf.inl"),
            os.path.join(CSRC, "raynet_train.inl"), os.path.join(CSRC, "raynet_eval.inl"), HEADER]
    extra = list(extra_flags) + os.environ.get("RAYNET_HIPCC_EXTRA", "").split()
    if out is None and ENV_LIB:
        # RAYNET_HIP_LIB names ANOTHER build of the library (a variant somebody made on purpose):
        # it is loaded as it is and never rebuilt with the default flags over its path -- an A/B
        # run would then test the default build without saying so.  The in-tree path
        # is the only implicit build target.
        if force or extra:
            raise RaynetHipError("RAYNET_HIP_LIB=%s is loaded as it is: force / extra flags (%s) "
                                 "need an explicit `out`" % (LIB_PATH, " ".join(extra)))
        if not os.path.exists(LIB_PATH):
            raise RaynetHipError("RAYNET_HIP_LIB=%s does not exist" % LIB_PATH)
        return LIB_PATH
    target = out or LIB_PATH
    if extra and out is None:
        raise RaynetHipError("a variant build (%s) needs its own output path" % " ".join(extra))
    if not force and os.path.exists(target) and \
            all(os.path.getmtime(target) >= os.path.getmtime(s) for s in srcs):
        return target
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + extra + [srcs[0], "-o", target]
    if extra:
        cmd.insert(1, '-DRN_BUILD_EXTRA="%s"' % " ".join(extra).replace('"', "'"))
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return target


class Config(ctypes.Structure):
    _fields_ = [
        ("M", ctypes.c_int32), ("D", ctypes.c_int32), ("N", ctypes.c_int32),
        ("F", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("padding", ctypes.c_int32), ("grid", ctypes.c_int32 * 3),
        ("bbox", ctypes.c_float * 6), ("device", ctypes.c_int32),
    ]


class Options(ctypes.Structure):
    # rn_options (include/raynet_hip.h)
    _fields_ = [("scatter_mode", ctypes.c_int32), ("box_level", ctypes.c_int32),
                ("box_pin", ctypes.c_int32), ("overlap", ctypes.c_int32),
                ("generic_sweep", ctypes.c_int32), ("sweep_rays_per_wave", ctypes.c_int32)]


class ScenePlan(ctypes.Structure):
    # rn_scene_plan (include/raynet_hip.h)
    _fields_ = [
        ("n_images", ctypes.c_int32), ("n", ctypes.c_int32), ("rows_per_image", ctypes.c_int64),
        ("ray_idxs", ctypes.c_void_p), ("order", ctypes.c_void_p),
        ("features_views", ctypes.c_void_p), ("cameras", ctypes.c_void_p),
        ("vox", ctypes.c_void_p), ("rvc", ctypes.c_void_p), ("Sr", ctypes.c_void_p),
        ("msgs", ctypes.c_void_p), ("ray_segments", ctypes.c_void_p),
        ("acc", ctypes.c_void_p * 2), ("acc_fixed", ctypes.c_void_p),
        ("depth", ctypes.c_void_p), ("prior", ctypes.c_float), ("row_layout", ctypes.c_int32),
        ("depth_image", ctypes.c_void_p), ("depth_image_stride", ctypes.c_int64),
        ("sweep_xcd_chunk", ctypes.c_int32),
    ]


RN_RUN_PREPARE, RN_RUN_SWEEP, RN_RUN_COMBINE, RN_RUN_DEPTH, RN_RUN_DEPTH_RANGE = 1, 2, 4, 8, 16

_P = ctypes.c_void_p
_I = ctypes.c_int32
_L = ctypes.c_int64
_F = ctypes.c_float

# name -> argtypes; mirrors include/raynet_hip.h one to one (checked by
# tests/test_abi.py against the header text)
SIGNATURES = {
    "rn_create": [ctypes.POINTER(Config), ctypes.POINTER(_P)],
    "rn_destroy": [_P],
    "rn_get_options": [_P, ctypes.POINTER(Options)],
    "rn_set_options": [_P, ctypes.POINTER(Options)],
    "rn_scene_run": [_P, ctypes.POINTER(ScenePlan), _I, _I, _I, _P],
    "rn_last_error": [_P],
    "rn_version": [],
    "rn_set_voxel_grid": [_P, _P, _P],
    "rn_fill_f32": [_P, _P, _L, _F, _P],
    "rn_fill_i32": [_P, _P, _L, _I, _P],
    "rn_sample_rays": [_P, _I, _P, _P, _P, _P, _P, _P],
    "rn_sample_points": [_P, _I, _P, _P, _P, _P, _P],
    "rn_compute_similarities": [_P, _I, _P, _P, _P, _P, _P, _P],
    "rn_voxel_traversal": [_P, _I, _P, _P, _P, _P, _P],
    "rn_planes_to_voxels": [_P, _I, _P, _P, _P, _P, _P, _P, _P],
    "rn_bp_sweep": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "rn_depth_estimation": [_P, _I, _P, _P, _P, _P, _P, _P, _P],
    "rn_mvcnn_similarities": [_P, _I, _P, _P, _P, _P, _P, _P, _P],
    "rn_mvcnn_depth": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "rn_mvcnn_voxel_space": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "rn_mvcnn_voxel_space_depth": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "rn_fused_bp_sweep": [_P, _I] + [_P] * 13,
    "rn_fused_depth": [_P, _I] + [_P] * 12,
    "rn_scene_prepare": [_P, _I, _P, ctypes.POINTER(_P), _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "rn_scene_prepare_all": [_P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "rn_scene_count_voxels": [_P, _I, _I, _P, _P, _P, _P],
    "rn_acc_copies": [_P],
    "rn_scatter_reset": [_P],
    "rn_slab_boxes_size": [_P, _L],
    "rn_scene_bind_slab_boxes": [_P, _P, _L, _P],
    "rn_scene_bind_scatter_items": [_P, _P, _L, _I, _P, _I],
    "rn_scatter_state": [_P, ctypes.POINTER(_I), ctypes.POINTER(ctypes.c_uint32),
                         ctypes.POINTER(ctypes.c_uint32)],
    "rn_scatter_settled": [_P],
    "rn_acc_size": [_P],
    "rn_acc_to_grid": [_P, _P, _P, _P],
    "rn_stitch_rows": [_P, _L, _P, _P, _P, _P],
    "rn_acc_from_grid": [_P, _P, _P, _P],
    "rn_scene_bp_sweep": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "rn_scene_bp_sweep_fixed": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "rn_acc_combine_fixed": [_P, _P, _F, _P, _P],
    "rn_acc_combine_fixed_range": [_P, _P, _L, _F, _P, _P],
    "rn_acc_combine": [_P, _P, _F, _P, _P],
    "rn_acc_reduce_local": [_P, _P, _P, _P],
    "rn_acc_add_prior": [_P, _P, _F, _P],
    "rn_scene_depth": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P],
    "rn_prof_begin": [_P, _I],
    "rn_prof_select": [_P, ctypes.c_uint32],
    "rn_prof_end": [_P, ctypes.POINTER(_I), _P, _P, _P],
    "rn_plane_weights": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "rn_train_bp_sweep": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "rn_train_depth": [_P, _I, _P, _P, _P, _P, _P, _P, _P],
    "rn_train_bp_sweep_bwd": [_P, _I] + [_P] * 11,
    "rn_train_depth_bwd": [_P, _I] + [_P] * 10,
    "rn_depthmap_points": [_P, _I, _I, _P, _P, _P, _P, _P],
    "rn_consistency_tau": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "rn_nearest_neighbors": [_P, _I, _P, _I, _P, _P, _P, _P],
    "rn_prof_offsets": [_P, _P],
    "rn_selftest_arith": [_P, _I, _P, _P, _P],
    "rn_selftest_quotient": [_P, _I, _P, _P, _P, _P],
    "rn_selftest_mapping": [_P, _I, _P, _P, _P, _P, _P],
    "rn_selftest_feature_offsets": [_P, _I, _P, _P, _P, _P, _P],
    "rn_timer_start": [_P, _P],
    "rn_timer_stop": [_P, _P, ctypes.POINTER(_F)],
}

_lib = None


def load():
    """dlopen the in-tree library; raises RaynetHipError if it was never built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RaynetHipError(
            "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(raynet_amd has no CPU fallback)" % LIB_PATH)
    # torch first: it brings its own libamdhip64, and a process must not end up with two HIP
    # runtimes (the library loaded before torch would bind /opt/rocm's copy and then see no
    # device once torch has initialised the other one)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.rn_destroy.restype = None
    lib.rn_last_error.restype = ctypes.c_char_p
    lib.rn_version.restype = ctypes.c_char_p
    lib.rn_acc_size.restype = ctypes.c_int64
    lib.rn_slab_boxes_size.restype = ctypes.c_int64
    _lib = lib
    return lib
