#!/usr/bin/env python3
"""K10 (the multi_view_cnn factory's kernel: sampling + plane sweep + softmax + arg-max) on the
reference's getting-started shape (1280 x 720 rays, 5 views, F = 32, mock Restrepo cameras) for
D = 16 / 32 / 64 / 48 planes, with the cooperative sweep's rays per wavefront chosen by D (round 6:
4 / 2 / 1) and forced to 1 (round 5's kernel): ms per image, rays/s, plane samples/s.  VERDICT r5
item 2: "rays/s per plane at D = 32 within 15 % of D = 64's".  Also the full path (k_sweep_map with
the folded first BP iteration) at the reference's CLI defaults, both ways.

    python tools/sweep_planes_ab.py > gpurun_out/r06_sweep_planes_ab.json
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
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.common.scene import restrepo_cameras_scene
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.hip_implementations.options import PathOptions
    from raynet_amd.synthetic import FeatureBank
    H, W, F, pad, nb = 720, 1280, 32, 11, 4
    scene = restrepo_cameras_scene(os.path.join(REPO, "tests", "golden", "restrepo_mock_scene_1"),
                                   (H, W), n_images=8, channels=1)
    g = torch.Generator(device="cuda").manual_seed(7)
    bank = FeatureBank([torch.randn((H + pad + 1, W + pad + 1, F), generator=g, device="cuda") * 0.25
                        for _ in range(scene.n_images)])
    n = H * W
    views = scene.view_indices_with_neighbors(0, nb)
    images = [scene.get_image(v) for v in views]
    feats = bank.stacked(views)
    rep = {"shape": "1280x720 rays, 5 views, F=32, mock Restrepo cameras, reference image 0", "k10": []}
    for D in (16, 32, 48, 64):
        ctx = get_context(1, D, nb + 1, F, H, W, pad, scene.bbox.ravel(), (1, 1, 1))
        P = ctx.dev(np.array([im.camera.P for im in images], np.float32))
        Pi = ctx.dev(images[0].camera.P_pinv.astype(np.float32))
        cc = ctx.dev(images[0].camera.center.ravel().astype(np.float32))
        ridx = ctx.dev(np.arange(n, dtype=np.int32))
        S = torch.zeros((n, D), device="cuda")
        pts = torch.zeros((n, D, 4), device="cuda")
        depth = torch.zeros((n,), device="cuda")
        keep = {}
        for mode in (0, 1):
            ctx.set_options(PathOptions(sweep_rays_per_wave=mode))
            ctx.mvcnn_depth(ridx, feats, P, Pi, cc, S, pts, depth)
            torch.cuda.synchronize()
            best = float("inf")
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ctx.mvcnn_depth(ridx, feats, P, Pi, cc, S, pts, depth)
                b.record()
                b.synchronize()
                best = min(best, a.elapsed_time(b))
            keep[mode] = (S.clone(), depth.clone())
            rep["k10"].append({"D": D, "rays_per_wave": "by D" if mode == 0 else 1, "ms": round(best, 4),
                               "rays_per_s": round(n / best * 1e3, 1),
                               "plane_samples_per_s": round(n * D / best * 1e3, 1)})
            print(json.dumps(rep["k10"][-1]), file=sys.stderr, flush=True)
        assert torch.equal(keep[0][0], keep[1][0]) and torch.equal(keep[0][1], keep[1][1])
        ctx.set_options(PathOptions())
        del S, pts, depth
    del feats
    torch.cuda.empty_cache()
    gp = GenerationParameters(depth_planes=32, neighbors=nb, grid_shape=np.array((256, 256, 128), np.int32),
                              max_number_of_marched_voxels=650, padding=pad, gamma_mrf=0.05)
    rep["cli_defaults"] = []
    for mode in (0, 1):
        fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0,
                                                options=PathOptions(sweep_rays_per_wave=mode))

        def step():
            for _ in fp.forward_pass(scene, (0, 5, 1)):
                pass
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        c = fp._ctx
        c.prof_begin(capacity=4096)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        fam = {}
        for name, _, kms in c.prof_end():
            fam[name] = fam.get(name, 0.0) + kms / 3
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        rep["cli_defaults"].append({"rays_per_wave": "by D" if mode == 0 else 1, "ms_per_step": round(ms, 3),
                                    "kernel_ms_per_step": {k: round(v, 3) for k, v in sorted(fam.items())}})
        print(json.dumps(rep["cli_defaults"][-1]), file=sys.stderr, flush=True)
        del fp
        torch.cuda.empty_cache()
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
