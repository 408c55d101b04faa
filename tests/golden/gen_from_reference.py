#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own runnable pieces.

Runs only in the build container (needs /root/reference); the outputs are small
.npz fixtures committed next to this script.  Nothing here is imported by the
tests -- they only read the .npz files.

What is run, and how:
  * raynet/ray_marching/ray_tracing.pyx -- built by oracle/build_ref.sh with
    Cython + gcc from where it lies; no source change.
  * raynet/mrf/mrf_np.py and raynet/planes_voxels_mapping/planes_voxels_mapping.py
    -- Python-2 sources; converted with `python3 -m lib2to3` into a scratch
    directory under /tmp (never into this repo) and imported from there.

NumPy-2 note (recorded because it changes dtypes): under NEP 50
`np.ones(f32) * (np.log(g) - np.log(1-g))` is float64, where the NumPy 1.x the
reference was written for gave float32.  The scratch copy's `np.log` is wrapped
to return a Python float for scalar arguments, which restores the float32
accumulator; nothing else is altered.
"""
import importlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = os.environ.get("RAYNET_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def load_reference_modules():
    subprocess.check_call(["bash", os.path.join(REPO, "oracle", "build_ref.sh")],
                          stderr=subprocess.DEVNULL)
    sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))
    ray_tracing = importlib.import_module("ray_tracing")

    scratch = tempfile.mkdtemp(prefix="raynet_ref_py3_")
    pkg = os.path.join(scratch, "refpy3")
    for sub in ("", "mrf", "planes_voxels_mapping", "utils"):
        os.makedirs(os.path.join(pkg, sub), exist_ok=True)
        open(os.path.join(pkg, sub, "__init__.py"), "w").close()
    for rel in ("mrf/mrf_np.py", "planes_voxels_mapping/planes_voxels_mapping.py",
                "utils/generic_utils.py"):
        shutil.copy(os.path.join(REF, "raynet", rel), os.path.join(pkg, rel))
    subprocess.check_call(
        [sys.executable, "-m", "lib2to3", "-w", "-n", pkg],
        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, scratch)
    mrf_np = importlib.import_module("refpy3.mrf.mrf_np")
    pvm = importlib.import_module("refpy3.planes_voxels_mapping.planes_voxels_mapping")

    class _NP1Log(object):
        """numpy proxy whose scalar log() returns a weak Python float (NumPy 1.x
        value-based casting for `f32_array * scalar`)."""

        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def log(x):
            r = np.log(x)
            return float(r) if np.ndim(r) == 0 else r

    mrf_np.np = _NP1Log()
    return ray_tracing, mrf_np, pvm, scratch


def face_points(rng, bbox, n):
    """Random points on the faces of bbox."""
    lo, hi = bbox[:3], bbox[3:]
    p = lo + rng.random((n, 3)) * (hi - lo)
    axis = rng.integers(0, 3, n)
    side = rng.integers(0, 2, n)
    p[np.arange(n), axis] = np.where(side == 0, lo[axis], hi[axis])
    return p.astype(np.float32)


def traverse_all(ray_tracing, bbox, grid, M, starts, ends):
    n = len(starts)
    rvi = np.zeros((n, M, 3), np.int32)
    rvc = np.zeros((n,), np.int32)
    for r in range(n):
        rvc[r] = ray_tracing.voxel_traversal(bbox, grid, rvi[r], starts[r], ends[r])
    return rvi, rvc


def gen_traversal(ray_tracing, out):
    rng = np.random.default_rng(20180618)
    cases = {}

    # the reference's own unit-test rays (tests/test_ray_marching.py:20-102)
    t2d_bbox = np.array([3, 3, 0, 6, 6, 1], np.float32)
    t2d_grid = np.array([3, 3, 1], np.int32)
    t2d_starts = np.array([[3., 4.1, .5], [4., 6., .5], [3., 3., .5], [6., 6., .5]], np.float32)
    t2d_ends = np.array([[6., 4.9, .5], [6., 5., .5], [6., 6., .5], [3., 3., .5]], np.float32)
    rvi, rvc = traverse_all(ray_tracing, t2d_bbox, t2d_grid, 10, t2d_starts, t2d_ends)
    assert list(rvc) == [3, 2, 5, 5], rvc  # the counts the reference test asserts
    cases["test2d"] = (t2d_bbox, t2d_grid, 10, t2d_starts, t2d_ends, rvi, rvc)

    b = np.array([0, 0, 0, 6, 6, 1], np.float32)
    g = np.array([6, 6, 1], np.int32)
    s = np.array([[0., 3.5, .5]], np.float32)
    e = np.array([[6., .5, .5]], np.float32)
    rvi, rvc = traverse_all(ray_tracing, b, g, 10, s, e)
    expected = np.array([[0, 3, 0], [0, 2, 0], [1, 2, 0], [2, 2, 0], [2, 1, 0], [3, 1, 0],
                         [4, 1, 0], [4, 0, 0], [5, 0, 0], [0, 0, 0]])  # test_ray_marching.py:66-77
    assert rvc[0] == 9 and np.all(rvi[0] == expected)
    cases["test2d_2"] = (b, g, 10, s, e, rvi, rvc)

    b = np.array([-3., -3., -0.5, 3., 3., 2.], np.float32)
    g = np.array([32, 32, 10], np.int32)
    s = np.array([[-1.40056884, -1.34645462, 2.]], np.float32)
    e = np.array([[-2.30040455, 3., -0.37297964]], np.float32)
    rvi, rvc = traverse_all(ray_tracing, b, g, 100, s, e)
    assert rvc[0] < 50
    cases["test3d"] = (b, g, 100, s, e, rvi, rvc)

    # seeded random chords, including axis-aligned and degenerate ones
    for name, bbox, grid, M, n in [
        ("rand_6x6x1", [0, 0, 0, 6, 6, 1], [6, 6, 1], 16, 200),
        ("rand_32x32x10", [-3, -3, -0.5, 3, 3, 2], [32, 32, 10], 100, 200),
        ("rand_16c", [-1, -1, -1, 1, 1, 1], [16, 16, 16], 48, 200),
        ("rand_restrepo32", [-5, -5, -0.7, 5, 5, 1.5], [32, 32, 32], 96, 200),
        ("rand_128c", [-1, -1, -1, 1, 1, 1], [128, 128, 128], 384, 96),
    ]:
        bbox = np.array(bbox, np.float32)
        grid = np.array(grid, np.int32)
        starts = face_points(rng, bbox, n)
        ends = face_points(rng, bbox, n)
        # a few special rays: axis aligned, zero-length, starting outside, grazing an edge
        starts[0] = [bbox[0], (bbox[1] + bbox[4]) / 2, (bbox[2] + bbox[5]) / 2]
        ends[0] = [bbox[3], (bbox[1] + bbox[4]) / 2, (bbox[2] + bbox[5]) / 2]
        starts[1] = ends[1]
        starts[2] = bbox[:3] - 1.0
        starts[3] = bbox[:3]
        ends[3] = bbox[3:]
        starts[4] = bbox[3:]
        ends[4] = bbox[:3]
        ends[5] = starts[5] + np.float32(1e-3)
        rvi, rvc = traverse_all(ray_tracing, bbox, grid, M, starts, ends)
        cases[name] = (bbox, grid, M, starts, ends, rvi, rvc)

    flat = {}
    for name, (bbox, grid, M, starts, ends, rvi, rvc) in cases.items():
        flat[name + "/bbox"] = bbox
        flat[name + "/grid"] = grid
        flat[name + "/M"] = np.int32(M)
        flat[name + "/starts"] = starts
        flat[name + "/ends"] = ends
        flat[name + "/rvi"] = rvi.astype(np.int16)  # small on disk; values < 2^15
        flat[name + "/rvc"] = rvc
    np.savez_compressed(out, **flat)
    print("wrote", out, {k: int(v[6].sum()) for k, v in cases.items()})
    return cases


def run_mrf(mrf_np, S, rvi, rvc, grid, gamma=0.05, iters=3):
    msgs = np.random.default_rng(1).random(S.shape).astype(np.float32)  # must be ignored
    accs = []

    def cb(S_, rvi_, rvc_, msgs_, acc_prev, it):
        accs.append(acc_prev.copy())

    devnull = open(os.devnull, "w")
    old = sys.stdout
    sys.stdout = devnull
    try:
        acc, msgs = mrf_np.belief_propagation(S, rvi, rvc, msgs, grid, gamma=gamma,
                                              bp_iterations=iters, progress_callback=cb)
        S_new = mrf_np.compute_depth_distribution(S, rvi, rvc, msgs, acc, np.zeros_like(S))
    finally:
        sys.stdout = old
    assert acc.dtype == np.float32, acc.dtype
    return np.stack(accs), msgs.copy(), S_new.copy()


def gen_mrf(ray_tracing, mrf_np, out):
    flat = {}
    bbox = np.array([0, 0, 0, 6, 6, 1], np.float32)
    grid = np.array([6, 6, 1], np.int32)

    def scene(name, M, rays, S_rows):
        n = len(rays)
        rvi = np.zeros((n, M, 3), np.int32)
        rvc = np.zeros((n,), np.int32)
        for r, (s, e) in enumerate(rays):
            rvc[r] = ray_tracing.voxel_traversal(bbox, grid, rvi[r], np.array(s, np.float32),
                                                 np.array(e, np.float32))
        S = np.array(S_rows, np.float32)
        accs, msgs, S_new = run_mrf(mrf_np, S, rvi, rvc, grid)
        flat[name + "/grid"] = grid
        flat[name + "/S"] = S
        flat[name + "/rvi"] = rvi
        flat[name + "/rvc"] = rvc
        flat[name + "/accs"] = accs
        flat[name + "/msgs"] = msgs
        flat[name + "/S_new"] = S_new

    # the six scenes of tests/test_mrf.py (:36-416), restated as data
    r1 = ([0., 3.5, .5], [6., .5, .5])
    s_peak10 = [0.075, 0.075, 0.075, 0.4, 0.075, 0.075, 0.075, 0.075, 0.075, 0.0]
    scene("single_ray", 10, [r1], [s_peak10])
    scene("two_rays", 10, [r1, ([6., 5.5, .5], [0., 2.5, .5])], [s_peak10, s_peak10])
    s_peak11 = s_peak10 + [0.0]
    s_two = [0.07, 0.07, 0.185, 0.07, 0.07, 0.07, 0.185, 0.07, 0.07, 0.07, 0.07]
    scene("two_rays_2", 11, [r1, ([6., 5.5, .5], [0., .5, .5])], [s_peak11, s_two])
    scene("three_rays", 11,
          [r1, ([0., 2.5, .5], [6., 2.5, .5]), ([6., 5.5, .5], [0., .5, .5])],
          [s_peak11, [0.45, 0.0875, 0.2, 0.0875, 0.0875, 0.0875, 0, 0, 0, 0, 0], s_two])
    conflict_S = np.zeros((2, 11), np.float32)
    conflict_S[0, 2] = 0.5
    conflict_S[0, 6] = 0.5
    conflict_S[1, 4] = 1.0
    scene("conflict", 11, [r1, ([0., 1.5, .5], [4.5, 6., .5])], conflict_S)

    # seeded random scenes on 3-D grids
    rng = np.random.default_rng(1234)
    for name, bb, g, M, n in [("rand32", [-1, -1, -1, 1, 1, 1], [32, 32, 32], 96, 48),
                              ("rand_aniso", [-3, -3, -0.5, 3, 3, 2], [24, 20, 8], 64, 48)]:
        bb = np.array(bb, np.float32)
        g = np.array(g, np.int32)
        starts = face_points(rng, bb, n)
        ends = face_points(rng, bb, n)
        ends[0] = starts[0]            # count <= 1 ray (skipped by mrf_np.py:300)
        rvi, rvc = traverse_all(ray_tracing, bb, g, M, starts, ends)
        S = np.zeros((n, M), np.float32)
        for r in range(n):
            c = rvc[r]
            if c > 0:
                v = rng.random(c) ** 4 + 1e-3
                if r % 5 == 0:
                    v[rng.integers(0, c)] += 5.0   # sharp peak
                S[r, :c] = (v / v.sum()).astype(np.float32)
        accs, msgs, S_new = run_mrf(mrf_np, S, rvi, rvc, g)
        flat[name + "/grid"] = g
        flat[name + "/S"] = S
        flat[name + "/rvi"] = rvi
        flat[name + "/rvc"] = rvc
        flat[name + "/accs"] = accs
        flat[name + "/msgs"] = msgs
        flat[name + "/S_new"] = S_new
    np.savez_compressed(out, **flat)
    print("wrote", out)


def gen_mapping(pvm, out):
    """NumPy `li` (np.interp) and `li_2` variants on voxels whose projections are
    non-decreasing along the ray (what a traversal produces and what the .cu
    walk assumes, planes_voxels_mapping.cu:68-76).  The centres lie on one line of
    a separable grid so that the same case can be fed to the HIP path, which keeps
    per-axis centre tables."""
    rng = np.random.default_rng(77)
    flat = {}
    for case, (C, D) in enumerate([(10, 5), (10, 5), (40, 32), (213, 64), (3, 2), (1, 16)]):
        start = (rng.random(3) - 1).astype(np.float32)
        end = (rng.random(3) + 1).astype(np.float32)
        ray = end - start
        t = np.sort(rng.random(C)) * 1.2 - 0.1          # some beyond [0,1] -> clipped
        # voxel centres on a separable grid line (x_i, y0, z0), like a real voxel grid:
        # the projection grows with x because ray_x > 0
        y0, z0 = rng.random(2) * 0.2
        rn = float((ray.astype(np.float64) ** 2).sum())
        x = start[0] + (t * rn - ray[1] * (y0 - start[1]) - ray[2] * (z0 - start[2])) / ray[0]
        voxels = np.stack([x, np.full(C, y0), np.full(C, z0)], axis=1).astype(np.float32)
        points = (start[:, None] + np.linspace(0, 1, D)[None] * ray[:, None])
        s = rng.random(D)
        s /= s.sum()
        s = s.astype(np.float32)
        li = pvm.single_ray_depth_to_voxels_li(voxels.T, points, s)
        li2 = pvm.single_ray_depth_to_voxels_li_2(voxels.T, points, s) if D > 1 and C > 0 else li
        flat["case%d/start" % case] = start
        flat["case%d/end" % case] = end
        flat["case%d/voxels" % case] = voxels
        flat["case%d/s" % case] = s
        flat["case%d/li" % case] = np.asarray(li, np.float64)
        flat["case%d/li_2" % case] = np.asarray(li2, np.float64)
    np.savez_compressed(out, **flat)
    print("wrote", out)


def main():
    ray_tracing, mrf_np, pvm, scratch = load_reference_modules()
    try:
        gen_traversal(ray_tracing, os.path.join(HERE, "ref_traversal.npz"))
        gen_mrf(ray_tracing, mrf_np, os.path.join(HERE, "ref_mrf_np.npz"))
        gen_mapping(pvm, os.path.join(HERE, "ref_mapping_np.npz"))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
