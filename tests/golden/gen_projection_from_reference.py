#!/usr/bin/env python3
"""Golden vectors for the GEOMETRY of SURVEY.md 8(a) row a2 (the plane sweep's projections)
from code the reference can run: `project` of raynet/utils/geometry.py:9-34.

The reference's own plane sweep (`compute_similarities_per_ray`, feature_similarities.cu:66-124)
exists as a PyCUDA kernel and as a TensorFlow graph only; neither can run in the build
container.  What it does before it touches a feature map is project every depth plane's point
of a ray into every view (feature_similarities.cu:10-32, `x = P point`, `x / x[2]`) and turn
the pixel into a feature index (`pixel_to_features`, :42-61).  The projection is the
reference's NumPy `project(P, points)` on the same points; this script calls THAT function
(imported through gen_sampling_from_reference.load_reference: lib2to3 copy in /tmp, nothing of
the reference is written to this repo) on

    points[k] = start + k (end - start) / (D - 1),  k = 0 .. D-1     (:84-98, in float32)

of the rays held by tests/golden/ref_sampling_np.npz -- `start` / `end` are the first and the
last of the reference's own sample points there -- for every view of the ray's camera group
(the five mock Restrepo cameras; the five ring cameras of bench.py's scene), with float64 and
with float32 matrices and points.  Output: tests/golden/ref_projection_np.npz -- the inputs
(view matrices, ray segments) and the reference's pixel coordinates, nothing else.
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_sampling_from_reference import load_reference   # noqa: E402

GROUPS = {"restrepo": ["restrepo%d" % k for k in range(5)], "ring": ["ring%d" % k for k in range(5)]}


def main():
    camera_mod, geometry, ns, scratch = load_reference()
    z = np.load(os.path.join(HERE, "ref_sampling_np.npz"))
    flat = {}
    try:
        for group, names in GROUPS.items():
            P64 = np.stack([z[n + "/P"] for n in names])                      # [V, 3, 4] float64
            flat[group + "/P"] = P64
            for ref, name in enumerate(names[:2]):                            # two reference views each
                H, W, D = (int(v) for v in z[name + "/HWD"])
                pts = z[name + "/points"]                                     # [n, D, 3] float32
                ok = np.flatnonzero(np.isfinite(pts).all((1, 2)))
                ok = ok[np.linspace(0, len(ok) - 1, min(len(ok), 100)).astype(np.int64)]   # 100 rays
                start, end = pts[ok, 0], pts[ok, -1]
                n = len(start)
                k = np.arange(D, dtype=np.float32)[None, :, None]
                # the kernel's plane points, in its float32 arithmetic (:84-98)
                plane = (start[:, None, :] + (k * (end - start)[:, None, :]) /
                         np.float32(D - 1)).astype(np.float32)
                hom32 = np.concatenate([plane, np.ones((n, D, 1), np.float32)], -1).reshape(-1, 4)
                pix64, pix32 = [], []
                for v in range(len(names)):
                    a = geometry.project(P64[v], hom32.astype(np.float64).T)          # [n D, 3]
                    b = geometry.project(P64[v].astype(np.float32), hom32.T)          # float32 in, out
                    assert b.dtype == np.float32 and np.all(a[:, 2] == 1.0)
                    pix64.append(a[:, :2].reshape(n, D, 2))
                    pix32.append(b[:, :2].reshape(n, D, 2))
                key = "%s/ref%d" % (group, ref)
                flat[key + "/HWD"] = np.array([H, W, D], np.int32)
                flat[key + "/start"], flat[key + "/end"] = start, end
                flat[key + "/pixels64"] = np.stack(pix64)                               # [V, n, D, 2]
                flat[key + "/pixels32"] = np.stack(pix32)
        out = os.path.join(HERE, "ref_projection_np.npz")
        np.savez_compressed(out, **flat)
        print("wrote", out, os.path.getsize(out), "bytes")
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
