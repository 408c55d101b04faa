"""Where the HOST time of a step goes (config 2, one GPU): the interpreter work between the
last map of one pass and the first launch of the next is on the critical path
(profiles/r03_step_timeline*.txt: ~60 us per step).  perf_counter marks around the pieces of a
pass (monkeypatched wrappers; 300 passes), then cProfile for the call counts."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd.common.generation_parameters import GenerationParameters  # noqa: E402
from raynet_amd.forward_pass import get_forward_pass_factory               # noqa: E402
from raynet_amd.synthetic import make_synthetic_scene                      # noqa: E402

H, W, V, D_, M_, G_ = 480, 640, 5, 64, 384, 128
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=32, padding=11, focal=1.5 * H, seed=1234)
gp = GenerationParameters(depth_planes=D_, neighbors=4, grid_shape=np.array([G_] * 3, np.int32),
                          max_number_of_marched_voxels=M_, padding=11, gamma_mrf=0.05)
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
marks = {}


def timed(obj, name):
    f = getattr(obj, name)

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            t1 = time.perf_counter()
            marks.setdefault(name, []).append((t0, t1))
    setattr(obj, name, wrapper)


def run(n, log=None):
    for _ in range(n):
        t0 = time.perf_counter()
        for _ in fp.forward_pass(scene, (0, V, 1)):
            pass
        if log is not None:
            log.append((t0, time.perf_counter()))


run(5)
for name in ("_view_features", "_context", "_build_plan", "_epilogue_buffers", "_run_plan_path"):
    timed(fp, name)
timed(fp._ctx, "scene_run")
torch.cuda.synchronize()
log = []
run(300, log)
us = lambda x: 1e6 * float(np.median(x))
starts = [t0 for t0, _ in log]
print("pass (call to exhaustion)          %8.1f us" % us([b - a for a, b in log]))
print("end of a pass -> start of the next %8.1f us" % us([log[i + 1][0] - log[i][1] for i in range(len(log) - 1)]))
for name, m in marks.items():
    per = len(m) // len(log)
    first = m[0::per][:len(log)]
    print("%-20s x%d per pass: first starts %6.1f us after the call, one call %6.1f us, all %6.1f us" % (
        name, per, us([f[0] - s for f, s in zip(first, starts)]), us([b - a for a, b in m]),
        per * float(np.mean([b - a for a, b in m])) * 1e6))
