"""Host cost of one pass: a tiny scene (24 x 32 rays, 32^3 voxels) whose GPU work is a few
microseconds per kernel, the step captured (capture="on"), so a pass's wall time is the
interpreter + one graph launch + one event wait.  With and without the identical-call fast path
of RayNetForwardPass._forward_pass_resident (fp._quick)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.forward_pass import get_forward_pass_factory
from raynet_amd.hip_implementations.options import PathOptions
from raynet_amd.synthetic import make_synthetic_scene
H, W = 24, 32
scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
gp = GenerationParameters(depth_planes=16, neighbors=4, grid_shape=np.array((32, 32, 32), np.int32),
                          max_number_of_marched_voxels=96, padding=11, gamma_mrf=0.05)
for capture in ("on", "off"):
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0,
                                            options=PathOptions(capture=capture))
    for _ in range(60):
        for _ in fp.forward_pass(scene, (0, 5, 1)):
            pass
    for quick in (True, False):
        ts = []
        for _ in range(400):
            if not quick:
                fp._quick = None
            t0 = time.perf_counter()
            for _ in fp.forward_pass(scene, (0, 5, 1)):
                pass
            ts.append(time.perf_counter() - t0)
        print("capture=%s captured=%s fast path %s: median %.1f us per pass (min %.1f)" % (
            capture, fp.captured, "on" if quick else "off", 1e6 * float(np.median(ts)), 1e6 * min(ts)))
