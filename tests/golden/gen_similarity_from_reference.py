#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8(a) row a2 -- the VALUES of the plane sweep (F-long dot products
over the view pairs, `/ (N (N - 1) / 2)`, stable softmax: feature_similarities.cu:66-124) -- from a
RUN of the reference.  The one stage of the path whose values no reference run pins yet
(DESIGN.md section 7): the reference executes it only as a PyCUDA kernel (or inside a TF graph),
and neither runs in the build container.  This script is for the first box that can:

    python tests/golden/gen_similarity_from_reference.py        # needs /root/reference (or
                                                                # RAYNET_REFERENCE) and PyCUDA + an
                                                                # NVIDIA GPU, or Keras + TensorFlow 1.x
    python -m pytest tests/test_similarity_reference.py         # oracle (CPU) and HIP (-m gpu) vs it

Route 1, PyCUDA (preferred: the very kernel forward_pass.py runs): the reference's Python-2
package converted with lib2to3 into a scratch directory under /tmp (never into this repo),
`raynet.cuda_implementations.similarities.perform_multi_view_cnn_forward_pass(D, N, F, H, W,
padding, bbox, "sample_in_bbox")` (similarities.py:11-127) compiled from the .cu files where they
lie, called on the seeded inputs below: S [n][D].
Route 2, Keras / TensorFlow (tf_implementations/forward_pass_implementations.py:65-123,
`compute_similarities`): the pair dot products at the pixels the oracle's index map names -- the
dot products only (its out-of-image rule differs, SURVEY.md Q10: rays whose projections all fall
inside the image are kept), softmax restated in float64 from them.  UNTESTED in the build container
(no TensorFlow there): it records `route` in the fixture, and the test holds it to 1e-4 instead of 1e-5.

Neither importable: prints why and exits 0 -- nothing is written, tests/test_similarity_reference.py
stays skipped.  Output: tests/golden/ref_similarity.npz -- inputs (ray indices, features, P, P_pinv,
centre, bbox, sizes) and the reference's S, nothing else; nothing of the reference travels."""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = os.environ.get("RAYNET_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
OUT = os.path.join(HERE, "ref_similarity.npz")

H, W, VIEWS, D, F, PAD = 24, 32, 3, 16, 8, 5
BBOX = np.array([-1, -1, -1, 1, 1, 1], np.float32)


def inputs():
    """Seeded case: this repo's ring cameras, planted feature maps (CPU tensors)."""
    from raynet_amd.synthetic import make_synthetic_scene
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=VIEWS, F=F, padding=PAD, focal=1.5 * H,
                                       seed=77, device="cpu")
    views = scene.view_indices_with_neighbors(0, VIEWS - 1)
    feats = bank.stacked(views).numpy().astype(np.float32)                      # [N][Hf][Wf][F]
    P = np.array([scene.get_image(v).camera.P for v in views], np.float32)      # [N][3][4]
    P_inv = scene.get_image(0).camera.P_pinv.astype(np.float32)                 # [4][3]
    center = scene.get_image(0).camera.center.ravel().astype(np.float32)        # [4]
    rng = np.random.default_rng(5)
    ray_idxs = np.sort(rng.choice(H * W, 256, replace=False)).astype(np.int32)
    return dict(ray_idxs=ray_idxs, features=feats, P=P, P_inv=P_inv, center=center)


def converted_reference():
    """The reference package as Python 3, in a scratch directory (sys.path gets it)."""
    scratch = tempfile.mkdtemp(prefix="raynet_ref_similarity_")
    pkg = os.path.join(scratch, "raynet")
    shutil.copytree(os.path.join(REF, "raynet"), pkg,
                    ignore=shutil.ignore_patterns("*.pyc", "*.so", "*.c"))
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", pkg],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, scratch)
    return scratch


def route_pycuda(x):
    import pycuda.autoinit  # noqa: F401  (raises without an NVIDIA device)
    converted_reference()
    # (the package __init__ chain may import Keras; the module itself needs PyCUDA only)
    import importlib.util
    base = os.path.join(sys.path[0], "raynet", "cuda_implementations")
    for name in ("utils", "similarities"):
        spec = importlib.util.spec_from_file_location("raynet.cuda_implementations." + name,
                                                      os.path.join(base, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
    fp = mod.perform_multi_view_cnn_forward_pass(D, VIEWS, F, H, W, PAD, BBOX, "sample_in_bbox")
    S = np.zeros((len(x["ray_idxs"]), D), np.float32)
    from pycuda.gpuarray import to_gpu
    S_gpu = to_gpu(S)
    fp(x["ray_idxs"], x["features"].ravel(), x["P"].ravel(), x["P_inv"].ravel(), x["center"], S_gpu,
       threads=256)
    return S_gpu.get(), np.ones(len(S), bool)


def route_tf(x):
    import keras  # noqa: F401
    from keras import backend as K
    converted_reference()
    sys.path.insert(0, os.path.join(sys.path[0], "raynet", "tf_implementations"))
    import forward_pass_implementations as fpi
    # pixel coordinates per view from this repo's oracle index map (pinned to the reference's own
    # `project`, tests/test_projection_reference.py); rays with any projection outside stay out
    from oracle import oracle
    o = oracle.Oracle(M=8, D=D, N=VIEWS, F=F, H=H, W=W, padding=PAD, bbox=BBOX, grid_shape=(4, 4, 4))
    starts, ends = o.sample(x["ray_idxs"], x["P_inv"], x["center"])
    idx = o.feature_indices(x["P"], starts, ends)                               # [n][N][D] vector index
    Hf, Wf = H + PAD + 1, W + PAD + 1
    n = len(x["ray_idxs"])
    inside = np.all(idx > 0, axis=(1, 2))
    off = PAD - (PAD - 1) // 2
    pix = [K.constant(np.stack([idx[:, v, :] // Wf - off, idx[:, v, :] % Wf - off]).astype(np.int32),
                      dtype="int32") for v in range(VIEWS)]
    feats = [K.constant(x["features"][v]) for v in range(VIEWS)]
    total = 0
    for i in range(VIEWS):
        for j in range(i + 1, VIEWS):
            total = total + fpi.compute_similarities(i, j, feats, pix, n, PAD, D)
    raw = K.eval(total).astype(np.float64) / (VIEWS * (VIEWS - 1) / 2)
    e = np.exp(raw - raw.max(1, keepdims=True))
    return (e / e.sum(1, keepdims=True)).astype(np.float32), inside


def main():
    if not os.path.isdir(REF):
        print("no reference at %s: nothing generated" % REF)
        return 0
    x = inputs()
    S = route = None
    for name, fn in (("pycuda", route_pycuda), ("tf", route_tf)):
        try:
            S, keep = fn(x)
            route = name
            break
        except Exception as e:      # not installed / no device / API drift: try the next route
            print("route %s unavailable: %s: %s" % (name, type(e).__name__, e))
    if S is None:
        print("neither PyCUDA nor Keras/TensorFlow can run the reference here: nothing generated "
              "(tests/test_similarity_reference.py stays skipped)")
        return 0
    np.savez_compressed(OUT, S=S[keep], route=np.array(route), bbox=BBOX,
                        sizes=np.array([H, W, VIEWS, D, F, PAD], np.int32),
                        ray_idxs=x["ray_idxs"][keep], features=x["features"], P=x["P"],
                        P_inv=x["P_inv"], center=x["center"])
    print("wrote %s (%d rays, route %s)" % (OUT, int(keep.sum()), route))
    return 0


if __name__ == "__main__":
    sys.exit(main())
