"""A/B of PathOptions (and of library builds) inside ONE process on ONE box: the variants take
turns, several rounds each, and the medians are compared -- boxes of the pool differ by 3 %,
consecutive runs on one box by 0.5 %.

    python tools/ab_options.py base plan_path=0 "overlap=1,box_level=1"
    CONFIG=config4 STEPS=10 ROUNDS=5 python tools/ab_options.py base deterministic=1

Every argument is one variant: comma-separated field=value pairs of PathOptions ("base" = the
defaults).  LIB=<path to another libraynet_hip.so> runs everything on that build.  PROF=1 adds
the per-family kernel times (every launch bracketed: +0.05 ms per step)."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd import _lib                                              # noqa: E402
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.environ["LIB"]
from raynet_amd.common.generation_parameters import GenerationParameters  # noqa: E402
from raynet_amd.forward_pass import get_forward_pass_factory               # noqa: E402
from raynet_amd.hip_implementations.options import PathOptions             # noqa: E402
from raynet_amd.synthetic import make_synthetic_scene                      # noqa: E402

if os.environ.get("CONFIG", "config2") == "config4":
    H, W, V, D_, M_, G_ = 480, 640, 9, 128, 768, 256
else:
    H, W, V, D_, M_, G_ = 480, 640, 5, 64, 384, 128
STEPS, ROUNDS = int(os.environ.get("STEPS", "20")), int(os.environ.get("ROUNDS", "5"))
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=32, padding=11, focal=1.5 * H, seed=1234)
gp = GenerationParameters(depth_planes=D_, neighbors=min(4, V - 1) if V <= 5 else V - 1,
                          grid_shape=np.array([G_] * 3, np.int32),
                          max_number_of_marched_voxels=M_, padding=11, gamma_mrf=0.05)


def parse(spec):
    if spec == "base":
        return PathOptions()
    kw = {}
    for item in spec.split(","):
        k, v = item.split("=")
        field = PathOptions.__dataclass_fields__[k]
        env = [e for e, (f, _) in PathOptions.ENV.items() if f == k]
        kw[k] = PathOptions.ENV[env[0]][1](v) if env else field.type(v)
    return PathOptions(**kw)


variants = sys.argv[1:] or ["base"]
fps = {}
for spec in variants:
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0, options=parse(spec))
    for _ in range(3):
        for _ in fp.forward_pass(scene, (0, V, 1)):
            pass
    fps[spec] = fp
    # (each variant keeps its own plan: 7 GB at config 2, 25 GB at config 4)
times = {s: [] for s in variants}
fams = {s: {} for s in variants}
prof = os.environ.get("PROF") == "1"
for rnd in range(ROUNDS):
    for spec in variants:
        fp = fps[spec]
        fp._ctx.set_options(fp.options)
        torch.cuda.synchronize()
        if prof:
            fp._ctx.prof_begin(capacity=64 * STEPS)
        t0 = time.perf_counter()
        for _ in range(STEPS):
            for _ in fp.forward_pass(scene, (0, V, 1)):
                pass
        torch.cuda.synchronize()
        times[spec].append((time.perf_counter() - t0) / STEPS * 1e3)
        if prof:
            for name, _, ms in fp._ctx.prof_end():
                fams[spec][name] = fams[spec].get(name, 0.0) + ms / (STEPS * ROUNDS)
base = float(np.median(times[variants[0]]))
for spec in variants:
    t = times[spec]
    line = "%-40s median %.3f ms/step  (min %.3f max %.3f)  %+.2f %%" % (
        spec, float(np.median(t)), min(t), max(t), 100.0 * (float(np.median(t)) / base - 1))
    if prof:
        line += "   " + " ".join("%s=%.3f" % kv for kv in sorted(fams[spec].items()))
    print(line)
