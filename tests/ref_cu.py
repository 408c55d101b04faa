"""Runs the REFERENCE's own kernels on the GPU -- test infrastructure.

oracle/build_ref_cu.py compiles the reference's .cu text (placeholders substituted the way
cuda_implementations/raynet_fp.py:230-248 does, otherwise unchanged) for gfx950 into
oracle/_ref/raynet_ref_<shape>_<fma|nofma>.co.  This module loads such a code object with
hipModuleLoad and launches its kernels with the reference's launch shape: one thread per ray,
`blocks = ceil(n / threads)` (raynet_fp.py:304-305; the reference's default of 2048 threads
per block is not launchable, 256 here -- the block size changes no result).

Arguments are torch CUDA tensors (or NumPy arrays, copied in), in the order of the kernel's
signature.  Nothing here is imported by the product package.
"""
import ctypes
import json
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(REPO, "oracle", "_ref")
MANIFEST = os.path.join(REF_DIR, "raynet_ref_cu.json")


def available():
    return os.path.exists(MANIFEST)


def manifest():
    with open(MANIFEST) as fh:
        return json.load(fh)


_hip = None


def _runtime():
    """The HIP runtime this process already runs on (torch's copy): a second one would not see
    torch's allocations."""
    global _hip
    if _hip is None:
        import torch
        torch.cuda.init()
        path = None
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        assert path, "torch has not loaded libamdhip64"
        _hip = ctypes.CDLL(path)
        _hip.hipModuleLoad.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p]
        _hip.hipModuleGetFunction.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p,
                                              ctypes.c_char_p]
        _hip.hipModuleLaunchKernel.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 7 + \
            [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]
        _hip.hipModuleUnload.argtypes = [ctypes.c_void_p]
    return _hip


class RefCu(object):
    """One code object = the reference's kernels with one shape baked in."""

    def __init__(self, shape, variant="nofma", threads=256):
        import torch
        self.torch = torch
        m = manifest()
        self.cfg = m["shapes"][shape]
        self.symbols = m["kernels"]
        self.shape, self.variant, self.threads = shape, variant, int(threads)
        self.M, self.D, self.N, self.F = (self.cfg[k] for k in "MDNF")
        path = os.path.join(REF_DIR, "raynet_ref_%s_%s.co" % (shape, variant))
        hip = _runtime()
        self._mod = ctypes.c_void_p()
        rc = hip.hipModuleLoad(ctypes.byref(self._mod), path.encode())
        assert rc == 0, "hipModuleLoad(%s) -> %d" % (path, rc)
        self._fn = {}

    def _function(self, name):
        if name not in self._fn:
            f = ctypes.c_void_p()
            rc = _runtime().hipModuleGetFunction(ctypes.byref(f), self._mod,
                                                 self.symbols[name].encode())
            assert rc == 0, "hipModuleGetFunction(%s) -> %d" % (name, rc)
            self._fn[name] = f
        return self._fn[name]

    def dev(self, a, dtype=None):
        torch = self.torch
        if not torch.is_tensor(a):
            a = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            a = a.to(dtype)
        a = a.to("cuda").contiguous()
        assert a.dtype in (torch.float32, torch.int32), a.dtype
        return a

    def launch(self, name, n, *tensors):
        """kernel(int n, pointers...) over ceil(n / threads) blocks on torch's current stream;
        returns after the launch (asynchronous)."""
        torch = self.torch
        if int(n) == 0:
            return
        nval = ctypes.c_int(int(n))
        vals = [nval] + [ctypes.c_void_p(t.data_ptr()) for t in tensors]
        params = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.pointer(v), ctypes.c_void_p)
                                                 for v in vals])
        blocks = (int(n) + self.threads - 1) // self.threads
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = _runtime().hipModuleLaunchKernel(self._function(name), blocks, 1, 1, self.threads, 1, 1,
                                              0, stream, params, None)
        assert rc == 0, "hipModuleLaunchKernel(%s) -> %d" % (name, rc)

    def timed(self, name, n, *tensors, repeats=3):
        """Best-of-`repeats` duration of one launch in ms (torch events on the launch stream)."""
        torch = self.torch
        best = float("inf")
        for _ in range(repeats):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self.launch(name, n, *tensors)
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b))
        return best

    # ---- the stand-alone kernels, one method each (arguments in the kernel's order) ----
    def sample_points(self, ray_idxs, P_inv, center):
        """sampling_schemes.cu:92-122 -> points [n, D, 4]"""
        r = self.dev(ray_idxs)
        pts = self.torch.zeros((len(r), self.D, 4), device="cuda")
        self.launch("batch_sample_points_in_bbox", len(r), r, self.dev(P_inv), self.dev(center), pts)
        return pts

    def similarities(self, features, P, starts, ends):
        """feature_similarities.cu:126-146 -> S [n, D]; S starts zero-filled (the kernel adds
        into it: forward_pass.py:320 zero-fills the global S of the unfused path)."""
        s, e = self.dev(starts), self.dev(ends)
        S = self.torch.zeros((len(s), self.D), device="cuda")
        self.launch("batch_compute_similarities", len(s), self.dev(features), self.dev(P), s, e, S)
        return S

    def mvcnn_similarities(self, ray_idxs, features, P, P_inv, center):
        """similarities.py:44-81 (a1 + a2) -> S [n, D]"""
        r = self.dev(ray_idxs)
        S = self.torch.zeros((len(r), self.D), device="cuda")
        self.launch("batch_multi_view_cnn_forward_pass", len(r), r, self.dev(features), self.dev(P),
                    self.dev(P_inv), self.dev(center), S)
        return S

    def traversal(self, starts, ends):
        """ray_tracing.cu:145-163 -> rvi [n, M, 3], rvc [n] (zero-filled first: Q11)"""
        s, e = self.dev(starts), self.dev(ends)
        rvi = self.torch.zeros((len(s), self.M, 3), dtype=self.torch.int32, device="cuda")
        rvc = self.torch.zeros((len(s),), dtype=self.torch.int32, device="cuda")
        self.launch("batch_voxel_traversal", len(s), s, e, rvi, rvc)
        return rvi, rvc

    def planes_to_voxels(self, voxel_grid, rvi, rvc, starts, ends, S):
        """planes_voxels_mapping.cu:94-119 -> S_voxel [n, M]"""
        rvi, rvc = self.dev(rvi, self.torch.int32), self.dev(rvc)
        out = self.torch.zeros((len(rvc), self.M), device="cuda")
        self.launch("batch_planes_voxels_mapping", len(rvc), self.dev(voxel_grid), rvi, rvc,
                    self.dev(starts), self.dev(ends), self.dev(S), out)
        return out

    def bp_sweep(self, S, rvi, rvc, acc_in, msgs, acc_out):
        """mrf_bp.cu:180-204; S is clipped in place, msgs in place (the reference aliases in and
        out, mrf_cuda.py:73-75), acc_out added into."""
        self.launch("batch_belief_propagation", len(rvc), S, rvi, rvc, acc_in, msgs, acc_out, msgs)

    def depth_estimation(self, S, rvi, rvc, acc, msgs):
        """mrf_bp.cu:206-229 -> S_new [n, M]"""
        out = self.torch.zeros((len(rvc), self.M), device="cuda")
        self.launch("batch_depth_estimation", len(rvc), S, rvi, rvc, acc, msgs, out)
        return out
