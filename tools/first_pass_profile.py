#!/usr/bin/env python3
"""Where the FIRST pass over a new scene spends its time beyond a warm pass (bench.py's
`first_pass_ms`; the reference's caller makes one pass per scene, scripts/forward_pass.py:120-142):
cProfile of the host side of three first passes (new Scene objects, same driver), top entries by
cumulative time, next to the wall clock of first and warm passes.

    python tools/first_pass_profile.py > gpurun_out/r06_first_pass_profile.txt
"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bench
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.common.scene import Scene
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import _FeatureOnlyImage, make_synthetic_scene, ring_cameras
    cfg = bench.CONFIGS["config2"]
    H, W, V = cfg["H"], cfg["W"], cfg["views"]
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=cfg["F"], padding=cfg["padding"],
                                       focal=1.5 * H, seed=1234)
    gp = GenerationParameters(depth_planes=cfg["D"], neighbors=4, grid_shape=np.array(cfg["grid"], np.int32),
                              max_number_of_marched_voxels=cfg["M"], padding=cfg["padding"], gamma_mrf=0.05)
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)

    def step(sc):
        for _ in fp.forward_pass(sc, (0, V, 1)):
            pass
    for _ in range(4):
        step(scene)
    torch.cuda.synchronize()

    def fresh():
        return Scene([_FeatureOnlyImage(H, W, c) for c in ring_cameras(V, H, W, focal=1.5 * H)], scene.bbox)
    firsts, warms = [], []
    for _ in range(4):
        sc = fresh()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(sc)
        torch.cuda.synchronize()
        firsts.append((time.perf_counter() - t0) * 1e3)
        for _ in range(2):
            step(sc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(sc)
        torch.cuda.synchronize()
        warms.append((time.perf_counter() - t0) * 1e3)
    print("first passes (ms):", [round(x, 2) for x in firsts], " warm passes (ms):", [round(x, 2) for x in warms])
    pr = cProfile.Profile()
    scenes = [fresh() for _ in range(3)]
    torch.cuda.synchronize()
    pr.enable()
    for sc in scenes:
        step(sc)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print("host profile of 3 first passes (cumulative seconds for all three):")
    print(s.getvalue())


if __name__ == "__main__":
    main()
