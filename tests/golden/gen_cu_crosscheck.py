#!/usr/bin/env python3
"""Cross-check vectors for the CUDA flavour of the path (a1, a2, a4 and the
fused K1/K2 composition), for which the reference has NO test and no CPU twin.

HONEST LABEL.  This is NOT a reference build and nothing from it lands in
oracle/_ref: it runs the device functions as HOST code behind stand-in macros
(rounds 1-4 believed the .cu files unbuildable without nvcc; they are not --
oracle/build_ref_cu.py compiles them unchanged with hipcc for gfx950, and
tests/golden/ref_cu_gfx950.npz + tests/test_reference_kernels.py are the pin;
these vectors remain as a CPU-side second opinion).  What this script does, in
the build container only:
  * reads the .cu files where they lie under /root/reference,
  * fills the $placeholders exactly as cuda_implementations/raynet_fp.py:230-248
    does (string.Template.substitute),
  * prepends a few empty macros so the *device functions* compile as host C++
    (g++), appends a thin extern "C" caller, builds into a /tmp scratch dir,
  * runs the device functions on seeded inputs and stores inputs + outputs.
The vectors are therefore "the reference's device-function text executed on a
CPU": supplementary evidence next to the GPU run of the real kernels.  No reference
text is written into this repository; only arrays are.

Host-execution caveats: bbox/grid literals become double
expressions; round() is half-away-from-zero; the thread-local S[D] is zeroed by
the caller (Q3); atomicAdd is serial.
"""
import ctypes
import os
import shutil
import subprocess
import tempfile
from string import Template

import numpy as np

REF = os.environ.get("RAYNET_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

SHIM = r"""
#include <cmath>
#include <cfloat>
#include <algorithm>
using std::min; using std::max; using std::abs;
#define __device__
#define __global__
#define __inline__ inline
static struct { int x, y, z; } threadIdx, blockIdx, blockDim;
static inline float atomicAdd(float *p, float v) { float o = *p; *p += v; return o; }
"""

CALLER = r"""
extern "C" {
void x_sample(int ray_idx, float *P_inv, float *cc, float *s, float *e) {
    sample_in_bbox(ray_idx, P_inv, cc, s, e);
}
void x_similarities(float *features, float *P, float *s, float *e, float *S) {
    compute_similarities_per_ray(features, P, s, e, S);
}
void x_traversal(float *s, float *e, int *rvi, int *rvc) { voxel_traversal(s, e, rvi, rvc); }
void x_mapping(float *grid, int *rvi, int *rvc, float *s, float *e, float *S, float *S_new) {
    planes_voxels_mapping(grid, rvi, rvc, s, e, S, S_new);
}
void x_bp(float *S, int *rvi, int *rvc, float *acc_in, float *msg_in, float *acc_out, float *msg_out) {
    belief_propagation(S, rvi, rvc, acc_in, msg_in, acc_out, msg_out);
}
void x_depth(float *S, int *rvi, int *rvc, float *acc, float *msg, float *S_new) {
    depth_estimation(S, rvi, rvc, acc, msg, S_new);
}
}
"""


def py2_float_repr(x):
    # raynet_fp.py substitutes str(np.float32) under Python 2; values used here
    # are exactly representable so the text is unambiguous.
    return repr(float(x))


def build(cfg, scratch):
    files = ["ray_tracing.cu", "utils.cu", "planes_voxels_mapping.cu",
             "feature_similarities.cu", "sampling_schemes.cu", "mrf_bp.cu"]  # raynet_fp.py:43-50
    src = ""
    for f in files:
        with open(os.path.join(REF, "raynet", "cuda_implementations", f)) as fh:
            src += fh.read()
    bbox, grid = cfg["bbox"], cfg["grid"]
    text = Template(src).substitute(
        max_voxels=cfg["M"], depth_planes=cfg["D"], n_views=cfg["N"], padding=cfg["padding"],
        features_dimensions=cfg["F"], width=cfg["W"], height=cfg["H"],
        grid_x=grid[0], grid_y=grid[1], grid_z=grid[2],
        bbox_min_x=py2_float_repr(bbox[0]), bbox_min_y=py2_float_repr(bbox[1]),
        bbox_min_z=py2_float_repr(bbox[2]), bbox_max_x=py2_float_repr(bbox[3]),
        bbox_max_y=py2_float_repr(bbox[4]), bbox_max_z=py2_float_repr(bbox[5]),
        sampling_scheme="sample_in_bbox")
    cpp = os.path.join(scratch, "dev_%s.cpp" % cfg["name"])
    so = os.path.join(scratch, "dev_%s.so" % cfg["name"])
    with open(cpp, "w") as fh:
        fh.write(SHIM + text + CALLER)
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-fno-fast-math",
                           "-fpermissive", "-w", "-shared", "-fPIC", cpp, "-o", so])
    return ctypes.CDLL(so)


def look_at_camera(pos, target, f, H, W):
    pos = np.asarray(pos, np.float64)
    z = np.asarray(target, np.float64) - pos
    z /= np.linalg.norm(z)
    up = np.array([0., 0., 1.])
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    t = -R.dot(pos).reshape(3, 1)
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]])
    P = K.dot(np.hstack([R, t]))                       # common/camera.py:52-58
    P_inv = np.linalg.pinv(P)                          # :60-65
    center = np.vstack([(-np.linalg.inv(R)).dot(t), [1]])   # :43-50
    return P.astype(np.float32), P_inv.astype(np.float32), center.astype(np.float32).ravel()


def voxel_grid(bbox, grid):
    xyz = [np.linspace(s, e, c, endpoint=False, dtype=np.float32)
           for s, e, c in zip(bbox[:3], bbox[3:], grid)]
    bin_size = np.array([a[1] - a[0] for a in xyz]).reshape(3, 1, 1, 1)
    g = np.stack(np.meshgrid(*xyz, indexing="ij")) + bin_size / 2   # generic_utils.py:90-110
    return np.ascontiguousarray(g.transpose(1, 2, 3, 0), dtype=np.float32)  # forward_pass.py:573-575


def p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def run(cfg, lib):
    M, D, N, F, H, W, pad = (cfg[k] for k in ("M", "D", "N", "F", "H", "W", "padding"))
    bbox = np.array(cfg["bbox"], np.float32)
    grid = np.array(cfg["grid"], np.int32)
    rng = np.random.default_rng(cfg["seed"])
    feats = (rng.standard_normal((N, H + pad + 1, W + pad + 1, F), dtype=np.float32) *
             np.float32(0.25))
    cams = []
    for v in range(N):
        a = 2 * np.pi * v / N * 0.35
        cams.append(look_at_camera([3 * np.cos(a), 3 * np.sin(a), 0.3 + 0.2 * v], [0, 0, 0],
                                   cfg["focal"], H, W))
    P = np.stack([c[0] for c in cams])
    P_inv, center = cams[0][1], cams[0][2]
    vg = voxel_grid(bbox, grid)
    ray_idxs = np.sort(rng.choice(H * W, size=cfg["n_rays"], replace=False)).astype(np.int32)
    n = len(ray_idxs)

    starts = np.zeros((n, 3), np.float32)
    ends = np.zeros((n, 3), np.float32)
    S = np.zeros((n, D), np.float32)
    rvi = np.zeros((n, M, 3), np.int32)
    rvc = np.zeros((n,), np.int32)
    Sv = np.zeros((n, M), np.float32)
    gamma = 0.05
    prior = np.float32(np.log(gamma) - np.log(1 - gamma))
    acc_in = np.full(tuple(grid), prior, np.float32)
    acc_out = np.full(tuple(grid), prior, np.float32)
    msgs = np.zeros((n, M), np.float32)
    S_new = np.zeros((n, M), np.float32)
    depth = np.zeros((n,), np.float32)
    Pf = np.ascontiguousarray(P.ravel())
    for r in range(n):
        lib.x_sample(int(ray_idxs[r]), p(P_inv), p(center), p(starts[r]), p(ends[r]))
        lib.x_similarities(p(feats), p(Pf), p(starts[r]), p(ends[r]), p(S[r]))   # S[r] is zero (Q3)
        lib.x_traversal(p(starts[r]), p(ends[r]), p(rvi[r]), p(rvc[r:r + 1]))
        lib.x_mapping(p(vg), p(rvi[r]), p(rvc[r:r + 1]), p(starts[r]), p(ends[r]), p(S[r]), p(Sv[r]))
    Sv_mapped = Sv.copy()
    ok = rvc >= 2          # count==1 gives +inf in mrf_bp.cu:157-165 (Q4); excluded
    for r in range(n):
        if ok[r]:
            Sr = Sv[r].copy()        # belief_propagation clips S in place (mrf_bp.cu:103-111)
            lib.x_bp(p(Sr), p(rvi[r]), p(rvc[r:r + 1]), p(acc_in), p(msgs[r]), p(acc_out), p(msgs[r]))
    for r in range(n):
        if ok[r]:
            Sr = Sv[r].copy()
            lib.x_depth(p(Sr), p(rvi[r]), p(rvc[r:r + 1]), p(acc_out), p(msgs[r]), p(S_new[r]))
            i = int(np.argmax(S_new[r]))                       # raynet_fp.py:199-204
            c = vg[tuple(rvi[r, i])]
            depth[r] = np.sqrt(((c - center[:3]) ** 2).sum(dtype=np.float32))   # :221-226
    out = dict(M=M, D=D, N=N, F=F, H=H, W=W, padding=pad, bbox=bbox, grid=grid,
               seed=cfg["seed"], gamma=np.float32(gamma), P=P, P_inv=P_inv, center=center,
               ray_idxs=ray_idxs, starts=starts, ends=ends, S=S, rvi=rvi.astype(np.int16),
               rvc=rvc, S_voxel=Sv_mapped, msgs=msgs, acc_out=acc_out, S_new=S_new,
               depth=depth, bp_valid=ok)
    return out


CONFIGS = [
    dict(name="small", M=48, D=16, N=3, F=8, H=24, W=32, padding=5, bbox=[-1, -1, -1, 1, 1, 1],
         grid=[16, 16, 16], focal=36.0, seed=11, n_rays=256),
    dict(name="wide", M=96, D=64, N=5, F=32, H=30, W=40, padding=11, bbox=[-1, -1, -1, 1, 1, 1],
         grid=[32, 32, 32], focal=45.0, seed=12, n_rays=96),
    dict(name="aniso", M=64, D=32, N=4, F=16, H=20, W=28, padding=11,
         bbox=[-1.5, -1, -0.5, 1.5, 1, 0.75], grid=[24, 16, 10], focal=30.0, seed=13, n_rays=128),
]


def main():
    scratch = tempfile.mkdtemp(prefix="raynet_cu_host_")
    try:
        flat = {}
        for cfg in CONFIGS:
            lib = build(cfg, scratch)
            res = run(cfg, lib)
            for k, v in res.items():
                flat["%s/%s" % (cfg["name"], k)] = np.asarray(v)
            print(cfg["name"], "rays", len(res["rvc"]), "mean count", res["rvc"].mean(),
                  "bp-valid", int(res["bp_valid"].sum()))
        out = os.path.join(HERE, "crosscheck_cu_host.npz")
        np.savez_compressed(out, **flat)
        print("wrote", out, os.path.getsize(out))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
