"""ctypes front-end of the CPU oracle (oracle/raynet_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.  The product package raynet_amd never imports it.

Each method restates one step of RayNet's forward_pass path; the C file cites
the reference file:line every function follows.  All arrays are host NumPy
arrays, C-contiguous, float32 / int32.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libraynet_oracle.so")


class _Config(ctypes.Structure):
    _fields_ = [
        ("M", ctypes.c_int32),
        ("D", ctypes.c_int32),
        ("N", ctypes.c_int32),
        ("F", ctypes.c_int32),
        ("H", ctypes.c_int32),
        ("W", ctypes.c_int32),
        ("padding", ctypes.c_int32),
        ("grid", ctypes.c_int32 * 3),
        ("bbox", ctypes.c_float * 6),
    ]


def build(force=False):
    """Compile the C restatement (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "raynet_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.rno_voxel_traversal.restype = ctypes.c_int
        _lib.rno_max_threads.restype = ctypes.c_int
        _lib.rno_depth_from_distribution.restype = ctypes.c_float
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def voxel_grid_centers(bbox, grid_shape):
    """Voxel centres laid out [gx][gy][gz][3] float32.

    Follows utils/generic_utils.py:90-110 (get_voxel_grid) and the transpose at
    forward_pass.py:573-575.
    """
    bbox = np.asarray(bbox, dtype=np.float32).reshape(-1)
    xyz = [
        np.linspace(s, e, int(c), endpoint=False, dtype=np.float32)
        for s, e, c in zip(bbox[:3], bbox[3:], grid_shape)
    ]
    bin_size = np.array(
        [(x[1] - x[0]) if len(x) > 1 else np.float32(e - s)
         for x, s, e in zip(xyz, bbox[:3], bbox[3:])],
        dtype=np.float32,
    ).reshape(3, 1, 1, 1)
    grid = np.stack(np.meshgrid(*xyz, indexing="ij")) + bin_size / 2
    return np.ascontiguousarray(grid.transpose(1, 2, 3, 0), dtype=np.float32)


class Oracle(object):
    """One configuration of the path (the reference bakes the same set of
    values into its kernels, cuda_implementations/raynet_fp.py:230-248)."""

    def __init__(self, M, D, N, F, H, W, padding, bbox, grid_shape, threads=1):
        self.lib = _load()
        self.cfg = _Config()
        self.cfg.M, self.cfg.D, self.cfg.N, self.cfg.F = int(M), int(D), int(N), int(F)
        self.cfg.H, self.cfg.W, self.cfg.padding = int(H), int(W), int(padding)
        bbox = np.asarray(bbox, dtype=np.float32).reshape(-1)
        for i in range(3):
            self.cfg.grid[i] = int(grid_shape[i])
        for i in range(6):
            self.cfg.bbox[i] = float(bbox[i])
        self.M, self.D, self.N, self.F = int(M), int(D), int(N), int(F)
        self.H, self.W, self.padding = int(H), int(W), int(padding)
        self.bbox = bbox
        self.grid_shape = tuple(int(g) for g in grid_shape)
        self.threads = int(threads)
        self._c = ctypes.byref(self.cfg)

    @staticmethod
    def max_threads():
        return int(_load().rno_max_threads())

    @staticmethod
    def set_robust_messages(on):
        """Process-wide: BP messages in the numerically robust form the HIP kernels use
        (direct suffix sum, log pos - log neg) instead of the reference's literal sequence;
        see raynet_oracle.c.  Off by default."""
        _load().rno_set_robust_messages(1 if on else 0)

    @staticmethod
    def set_private_accumulators(on):
        """Process-wide: a multi-threaded fused_bp sums every thread's messages in a private copy
        of the accumulator (merged at the end) instead of `omp atomic` adds into the caller's one
        array; see raynet_oracle.c.  Off by default."""
        _load().rno_set_private_accumulators(1 if on else 0)

    # -- a1 ---------------------------------------------------------------
    def sample(self, ray_idxs, P_inv, center):
        ray_idxs = _i32(ray_idxs)
        n = len(ray_idxs)
        P_inv, center = _f32(P_inv).ravel(), _f32(center).ravel()
        starts = np.zeros((n, 3), np.float32)
        ends = np.zeros((n, 3), np.float32)
        self.lib.rno_batch_sample(self._c, n, _p(ray_idxs), _p(P_inv), _p(center),
                                  _p(starts), _p(ends))
        return starts, ends

    # -- a2 ---------------------------------------------------------------
    def similarities(self, features, P, starts, ends):
        features, P = _f32(features), _f32(P).ravel()
        starts, ends = _f32(starts), _f32(ends)
        n = len(starts)
        S = np.zeros((n, self.D), np.float32)
        self.lib.rno_batch_similarities(self._c, n, _p(features), _p(P), _p(starts),
                                        _p(ends), _p(S), self.threads)
        return S

    def feature_indices(self, P, starts, ends):
        """[n, N, D, 2] (fx, fy) feature-map indices; integer parity surface."""
        P = _f32(P).ravel()
        starts, ends = _f32(starts), _f32(ends)
        n = len(starts)
        out = np.zeros((n, self.N, self.D, 2), np.int32)
        tmp = (ctypes.c_int * 2)()
        for r in range(n):
            for v in range(self.N):
                for k in range(self.D):
                    self.lib.rno_feature_index(self._c, _p(P), _p(starts[r]), _p(ends[r]),
                                               v, k, tmp)
                    out[r, v, k, 0], out[r, v, k, 1] = tmp[0], tmp[1]
        return out

    # -- a3 ---------------------------------------------------------------
    def voxel_traversal(self, bbox, grid_shape, voxels, ray_start, ray_end):
        """Same signature as ray_marching/ray_tracing.pyx:64 voxel_traversal."""
        bbox = _f32(bbox).ravel()
        grid_shape = _i32(grid_shape)
        assert voxels.dtype == np.int32 and voxels.flags["C_CONTIGUOUS"]
        return int(self.lib.rno_voxel_traversal(
            _p(bbox), _p(grid_shape), voxels.shape[0], _p(_f32(ray_start)),
            _p(_f32(ray_end)), _p(voxels)))

    def traversal(self, starts, ends):
        starts, ends = _f32(starts), _f32(ends)
        n = len(starts)
        rvi = np.zeros((n, self.M, 3), np.int32)
        rvc = np.zeros((n,), np.int32)
        self.lib.rno_batch_traversal(self._c, n, _p(starts), _p(ends), _p(rvi), _p(rvc))
        return rvi, rvc

    # -- a4 ---------------------------------------------------------------
    def planes_to_voxels(self, voxel_grid, rvi, rvc, starts, ends, S):
        voxel_grid = _f32(voxel_grid)
        rvi, rvc = _i32(rvi), _i32(rvc)
        starts, ends, S = _f32(starts), _f32(ends), _f32(S)
        n = len(rvc)
        S_new = np.zeros((n, self.M), np.float32)
        self.lib.rno_batch_planes_to_voxels(self._c, n, _p(voxel_grid), _p(rvi), _p(rvc),
                                            _p(starts), _p(ends), _p(S), _p(S_new))
        return S_new

    def plane_indices(self, voxel_grid, rvi, rvc, starts, ends):
        voxel_grid = _f32(voxel_grid)
        rvi, rvc = _i32(rvi), _i32(rvc)
        starts, ends = _f32(starts), _f32(ends)
        n = len(rvc)
        out = np.zeros((n, self.M), np.int32)
        for r in range(n):
            self.lib.rno_plane_indices(self._c, _p(voxel_grid), _p(rvi[r]), int(rvc[r]),
                                       _p(starts[r]), _p(ends[r]), _p(out[r]))
        return out

    # -- a5 / a6 ------------------------------------------------------------
    def prior(self, gamma):
        g = np.float32(np.log(gamma) - np.log(1 - gamma))
        return np.full(self.grid_shape, g, dtype=np.float32)

    def bp_sweep(self, S, rvi, rvc, acc_in, msgs, acc_out):
        """One sweep; msgs updated in place, acc_out accumulated in place."""
        S, rvi, rvc = _f32(S), _i32(rvi), _i32(rvc)
        assert msgs.dtype == np.float32 and msgs.flags["C_CONTIGUOUS"]
        assert acc_out.dtype == np.float32 and acc_out.flags["C_CONTIGUOUS"]
        acc_in = _f32(acc_in)
        self.lib.rno_batch_bp(self._c, len(rvc), _p(S), _p(rvi), _p(rvc), _p(acc_in),
                              _p(msgs), _p(acc_out), self.threads)
        return msgs

    def belief_propagation(self, S, rvi, rvc, msgs, gamma=0.05, bp_iterations=3,
                           callback=None):
        """mrf/mrf_np.py:243-330 schedule (messages zero-filled first)."""
        msgs.fill(0)
        acc_prev = self.prior(gamma)
        acc_new = self.prior(gamma)
        for it in range(bp_iterations):
            self.bp_sweep(S, rvi, rvc, acc_prev, msgs, acc_new)
            acc_prev[...] = acc_new
            acc_new[...] = self.prior(gamma)
            if callback is not None:
                callback(it, acc_prev, msgs)
        return acc_prev, msgs

    def depth_distribution(self, S, rvi, rvc, acc, msgs):
        S, rvi, rvc = _f32(S), _i32(rvi), _i32(rvc)
        acc, msgs = _f32(acc), _f32(msgs)
        S_new = np.zeros((len(rvc), self.M), np.float32)
        self.lib.rno_batch_depth(self._c, len(rvc), _p(S), _p(rvi), _p(rvc), _p(acc),
                                 _p(msgs), _p(S_new), self.threads)
        return S_new

    def depth_from_distribution(self, S_new, rvi, voxel_grid, center):
        S_new, rvi = _f32(S_new), _i32(rvi)
        voxel_grid, center = _f32(voxel_grid), _f32(center).ravel()
        out = np.zeros((len(S_new),), np.float32)
        for r in range(len(S_new)):
            out[r] = self.lib.rno_depth_from_distribution(
                self._c, _p(S_new[r]), _p(rvi[r]), _p(voxel_grid), _p(center))
        return out

    # -- a7 (fused K1 / K2) ---------------------------------------------------
    def fused_bp(self, ray_idxs, features, P, P_inv, center, voxel_grid, acc_in, msgs,
                 acc_out):
        ray_idxs = _i32(ray_idxs)
        n = len(ray_idxs)
        features, P = _f32(features), _f32(P).ravel()
        P_inv, center = _f32(P_inv).ravel(), _f32(center).ravel()
        voxel_grid, acc_in = _f32(voxel_grid), _f32(acc_in)
        assert msgs.dtype == np.float32 and msgs.flags["C_CONTIGUOUS"]
        assert acc_out.dtype == np.float32 and acc_out.flags["C_CONTIGUOUS"]
        rvi = np.zeros((n, self.M, 3), np.int32)
        rvc = np.zeros((n,), np.int32)
        Sv = np.zeros((n, self.M), np.float32)
        self.lib.rno_fused_bp(self._c, n, _p(ray_idxs), _p(features), _p(P), _p(P_inv),
                              _p(center), _p(voxel_grid), _p(rvi), _p(rvc), _p(Sv),
                              _p(acc_in), _p(msgs), _p(acc_out), self.threads)
        return rvi, rvc, Sv

    def fused_depth(self, ray_idxs, features, P, P_inv, center, voxel_grid, acc, msgs):
        ray_idxs = _i32(ray_idxs)
        n = len(ray_idxs)
        features, P = _f32(features), _f32(P).ravel()
        P_inv, center = _f32(P_inv).ravel(), _f32(center).ravel()
        voxel_grid, acc, msgs = _f32(voxel_grid), _f32(acc), _f32(msgs)
        rvi = np.zeros((n, self.M, 3), np.int32)
        rvc = np.zeros((n,), np.int32)
        Sv = np.zeros((n, self.M), np.float32)
        depth = np.zeros((n,), np.float32)
        self.lib.rno_fused_depth(self._c, n, _p(ray_idxs), _p(features), _p(P), _p(P_inv),
                                 _p(center), _p(voxel_grid), _p(rvi), _p(rvc), _p(Sv),
                                 _p(acc), _p(msgs), _p(depth), self.threads)
        return rvi, rvc, Sv, depth
