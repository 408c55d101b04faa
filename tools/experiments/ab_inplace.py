"""A/B of run-time switches on ONE driver object and ONE plan (the same buffers at the same
addresses): identical objects of one process differ by up to 1.4 % per step through where their
rows landed (tools/placement_probe.py), which drowns anything smaller in tools/ab_options.py.

    python tools/ab_inplace.py direct_maps spin_wait depth_head

Every argument is a switch flipped OFF/ON in turns on the same object (rounds alternate)."""
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

if os.environ.get("CONFIG", "config2") == "config4":
    H, W, V, D_, M_, G_ = 480, 640, 9, 128, 768, 256
else:
    H, W, V, D_, M_, G_ = 480, 640, 5, 64, 384, 128
STEPS, ROUNDS = int(os.environ.get("STEPS", "20")), int(os.environ.get("ROUNDS", "7"))
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=32, padding=11, focal=1.5 * H, seed=1234)
gp = GenerationParameters(depth_planes=D_, neighbors=min(4, V - 1) if V <= 5 else V - 1,
                          grid_shape=np.array([G_] * 3, np.int32),
                          max_number_of_marched_voxels=M_, padding=11, gamma_mrf=0.05)
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)


def run():
    for _ in fp.forward_pass(scene, (0, V, 1)):
        pass


for _ in range(3):
    run()
plan = fp._plan
image_ptr = plan["fast"].depth_image


def flip(name, on):
    if name == "direct_maps":
        plan["direct"] = on
        plan["fast"].depth_image = image_ptr if on else None
    else:
        setattr(fp.options, name, on)
        k = list(plan["key"])
        k[11] = fp.options.key()           # (the plan stays: these are read per pass)
        plan["key"] = tuple(k)


for name in sys.argv[1:]:
    t = {False: [], True: []}
    for rnd in range(ROUNDS):
        for on in (False, True):
            flip(name, on)
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(STEPS):
                run()
            torch.cuda.synchronize()
            t[on].append((time.perf_counter() - t0) / STEPS * 1e3)
    flip(name, True if name == "direct_maps" else getattr(type(fp.options)(), name))
    a, b = float(np.median(t[False])), float(np.median(t[True]))
    print("%-14s off %.3f (min %.3f)  on %.3f (min %.3f) ms/step  on vs off %+.2f %%" % (
        name, a, min(t[False]), b, min(t[True]), 100 * (b / a - 1)))
