#!/usr/bin/env python3
"""What a caller holding its feature maps in HOST memory would see (the reference's caller does:
Keras hands NumPy arrays to PyCUDA, forward_pass.py:604-627).  The library's boundary takes DEVICE
pointers (include/raynet_hip.h: rn_scene_plan.features), so the bench line's `value` has the maps
resident; this script times the upload of config 2's five maps next to a warm step and prints the
PCIe-inclusive rate for DESIGN.md section 5.  (The depth maps' way back to the host IS inside
the step.)

    gpurun -- python tools/pcie_inclusive.py
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bench
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene
    cfg = bench.CONFIGS["config2"]
    H, W, V, D, M, F, pad = (cfg[k] for k in ("H", "W", "views", "D", "M", "F", "padding"))
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=F, padding=pad, focal=1.5 * H, seed=1234)
    gp = GenerationParameters(depth_planes=D, neighbors=4, grid_shape=np.array(cfg["grid"], np.int32),
                              max_number_of_marched_voxels=M, padding=pad, gamma_mrf=0.05)
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)

    def step():
        for _ in fp.forward_pass(scene, (0, V, 1)):
            pass
    for _ in range(16):           # (the scatter's adaptive shape settles within 12 passes)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / 20 * 1e3
    dev = [bank.view_features(scene, v) for v in range(V)]
    host = [d.cpu().numpy().copy() for d in dev]                  # pageable, as NumPy hands them over
    pinned = [torch.from_numpy(h).pin_memory() for h in host]
    nbytes = sum(h.nbytes for h in host)

    def timed(fn, reps=5):
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) * 1e3)
        return best
    up_pageable = timed(lambda: [d.copy_(torch.from_numpy(h)) for d, h in zip(dev, host)])
    up_pinned = timed(lambda: [d.copy_(p, non_blocking=True) for d, p in zip(dev, pinned)])
    rays = V * H * W
    rep = {"config": "config2", "feature_maps_MB": round(nbytes / 1e6, 1), "step_ms": round(step_ms, 3),
           "upload_ms_pageable": round(up_pageable, 3), "upload_ms_pinned": round(up_pinned, 3),
           "upload_GBps_pageable": round(nbytes / up_pageable / 1e6, 1),
           "upload_GBps_pinned": round(nbytes / up_pinned / 1e6, 1),
           "rays_per_s_resident": round(rays / step_ms * 1e3),
           "rays_per_s_with_pageable_upload": round(rays / (step_ms + up_pageable) * 1e3),
           "rays_per_s_with_pinned_upload": round(rays / (step_ms + up_pinned) * 1e3)}
    print(json.dumps(rep, indent=1))
    out = os.path.join(REPO, "gpurun_out", "r05_pcie_inclusive.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rep, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
