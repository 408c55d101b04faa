#!/usr/bin/env python3
"""Where the accumulator scatter's time goes at a configuration (round 6, VERDICT r5 item 3: config
4's scatter at 0.28 of the roofline against config 2's 0.52): per tile shape of the LDS-box scatter
(PathOptions.box_level pinned: 0 = 128 rays x 32 steps, 1 = 256 x 16, 2 = the slab scatter) and for
the adaptive default, the kernel families' ms per step, the scatter's state (chunks, overflowed
chunks) and its algorithmic GB/s.

    python tools/scatter_probe.py --config config4 > gpurun_out/r06_scatter_probe_config4.json
"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4")
    ap.add_argument("--levels", default="auto,0,1,2")
    ap.add_argument("--extra", default="", help="comma list of name=value PathOptions overrides")
    args = ap.parse_args()
    import torch
    import bench
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import make_synthetic_scene
    cfg = bench.CONFIGS[args.config]
    H, W, V = cfg["H"], cfg["W"], cfg["views"]
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=cfg["F"], padding=cfg["padding"],
                                       focal=1.5 * H, seed=1234)
    gp = GenerationParameters(depth_planes=cfg["D"], neighbors=min(4, V - 1) if V <= 5 else V - 1,
                              grid_shape=np.array(cfg["grid"], np.int32),
                              max_number_of_marched_voxels=cfg["M"], padding=cfg["padding"], gamma_mrf=0.05)
    extra = {}
    for kv in filter(None, args.extra.split(",")):
        k, v = kv.split("=")
        extra[k] = int(v) if v.lstrip("-").isdigit() else v
    rep = {"config": args.config, "runs": []}
    for lv in args.levels.split(","):
        opt = PathOptions(**extra) if lv == "auto" else PathOptions(box_level=int(lv), box_pin=True, **extra)
        fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0, options=opt)

        def step():
            for _ in fp.forward_pass(scene, (0, V, 1)):
                pass
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ctx = fp._ctx
        ctx.prof_begin(capacity=8192)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        fam = {}
        for name, _, ms in ctx.prof_end():
            fam[name] = fam.get(name, 0.0) + ms / 3
        visits = float(sum(float(fp.voxel_count[r].sum().item()) for r in fp.voxel_count))
        rays = float(sum(int(fp.voxel_count[r].numel()) for r in fp.voxel_count))
        st = ctx.scatter_state()
        import time
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ms_step = (time.perf_counter() - t0) / 5 * 1e3
        run = {"box_level": lv, "ms_per_step": round(ms_step, 3), "kernel_ms_per_step": {k: round(v, 3) for k, v in sorted(fam.items())},
               "scatter_state": st,
               "scatter_algorithmic_GBps": round(3 * (8 * visits + 4 * rays) / (fam["scatter"] * 1e-3) / 1e9, 1),
               "mean_voxels_per_ray": round(visits / rays, 1)}
        rep["runs"].append(run)
        print(json.dumps(run), file=sys.stderr, flush=True)
        del fp
        torch.cuda.empty_cache()
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
