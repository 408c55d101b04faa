#!/usr/bin/env python3
"""What a bounding box whose corners are NOT float32 values does (round 5).

The reference bakes the box into its kernels as decimal text: `Template.substitute` writes
str(np.float32(-0.7)) = "-0.7", which the C compiler reads as the DOUBLE -0.7
(sampling_schemes.cu:65-77, ray_tracing.cu:18-30), while a non-templated kernel holds the float32
-0.699999988...  The mock Restrepo scene of BASELINE.json configs[0] has such a box
([-5, -5, -0.7, 5, 5, 1.5]).  This script runs the reference's OWN kernels (oracle/_ref, shape
"config1") and the library on every ray of the twelve mock cameras and counts what differs:
ray end points (a1), voxel lists on the SAME end points (a3: CUDA flavour vs the Cython flavour
the library follows), voxel lists end to end, plane-sweep columns end to end.

    gpurun -- python tools/ref_cu_bbox_census.py  ->  gpurun_out/r05_ref_cu_bbox_census.json
"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch
    import ref_cu
    from raynet_amd.common.scene import restrepo_cameras_scene
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.ray_marching.ray_tracing_hip import batch_voxel_traversal
    shape = ref_cu.manifest()["shapes"]["config1"]
    M, D, N, F, H, W, pad = (shape[k] for k in ("M", "D", "N", "F", "H", "W", "padding"))
    scene = restrepo_cameras_scene(os.path.join(REPO, "tests", "golden", "restrepo_mock_scene_1"), (H, W),
                                   scale=W / 1280.0)
    bbox = scene.bbox.ravel().astype(np.float32)
    assert np.allclose(bbox, shape["bbox"])
    ctx = get_context(M, D, N, F, H, W, pad, bbox, shape["grid"])
    vtr = batch_voxel_traversal(M, bbox, np.array(shape["grid"], np.int32))
    r = ref_cu.RefCu("config1", "nofma")
    rng = np.random.default_rng(3)
    feats = torch.from_numpy((rng.standard_normal((N, H + pad + 1, W + pad + 1, F), dtype=np.float32) * 0.25)).cuda()
    ridx = torch.arange(H * W, dtype=torch.int32, device="cuda")
    n = H * W
    tot = dict(rays=0, live_rays=0, start_differs=0, end_differs=0, max_endpoint_diff=0.0,
               lists_differ_same_endpoints=0,
               sweep_rays_gt_1e5=0, sweep_max=0.0)
    for cam_i in range(scene.n_images):
        cam = scene.get_image(cam_i).camera
        P_inv = ctx.dev(cam.P_pinv.astype(np.float32))
        cc = ctx.dev(cam.center.ravel().astype(np.float32))
        pts = r.sample_points(ridx, P_inv.reshape(-1), cc)              # the reference's a1
        s_ref = pts[:, 0, :3].contiguous()
        # its ray_end is not an output; the last point is start + (D-1)(end-start)/(D-1): use the
        # library's end where the starts agree, and compare the last points instead
        s = torch.zeros((n, 3), device="cuda")
        e = torch.zeros((n, 3), device="cuda")
        ctx.sample_rays(ridx, P_inv, cc, s, e)
        # (IEEE operations on the host: torch divides by a scalar through its reciprocal)
        sn, en = s.cpu().numpy(), e.cpu().numpy()
        last_hip = torch.from_numpy((sn + np.float32(D - 1) * (en - sn) / np.float32(D - 1)).astype(np.float32)).cuda()
        ds = (s != s_ref).any(1)
        de = (last_hip != pts[:, -1, :3]).any(1)
        tot["max_endpoint_diff"] = max(tot["max_endpoint_diff"], float((s - s_ref).abs().max()),
                                       float((last_hip - pts[:, -1, :3]).abs().max()))
        # a3 on the SAME end points (the library's): CUDA flavour vs the library (Cython flavour)
        rvi_r, rvc_r = r.traversal(s, e)
        rvi_h = torch.zeros((n, M, 3), dtype=torch.int32, device="cuda")
        rvc_h = torch.zeros((n,), dtype=torch.int32, device="cuda")
        vtr(s, e, rvi_h, rvc_h)
        same = ((rvi_r != rvi_h).any(2).any(1) | (rvc_r != rvc_h))
        # end to end: the reference's a3 on the reference's end points needs its ray_end; the fused
        # similarity kernel (a1 + a2) gives the end-to-end effect on the column instead
        views = scene.view_indices_with_neighbors(cam_i, N - 1)
        P = ctx.dev(np.array([scene.get_image(v).camera.P for v in views], np.float32))
        S_ref = r.mvcnn_similarities(ridx, feats, P.reshape(-1), P_inv.reshape(-1), cc)
        S_hip = torch.zeros((n, D), device="cuda")
        ctx.mvcnn_similarities(ridx, feats, P, P_inv, cc, S_hip)
        dS = (S_ref - S_hip).abs().max(1).values
        live = rvc_h > 0
        tot["rays"] += n
        tot["live_rays"] += int(live.sum())
        tot["start_differs"] += int(ds.sum())
        tot["end_differs"] += int(de.sum())
        tot["lists_differ_same_endpoints"] += int(same.sum())
        tot["sweep_rays_gt_1e5"] += int((dS > 1e-5).sum())
        tot["sweep_max"] = max(tot["sweep_max"], float(dS.max()))
    tot["box"] = [float(b) for b in bbox]
    tot["what"] = ("rays of the 12 mock Restrepo cameras at 36 x 64; start/end_differs: rays whose first / "
                   "last plane point from the reference's batch_sample_points_in_bbox is not bit-equal to "
                   "the library's; lists_differ_same_endpoints: reference batch_voxel_traversal (CUDA "
                   "flavour, double-promoted box literals) vs the library (Cython flavour) on identical "
                   "end points; sweep_*: batch_multi_view_cnn_forward_pass (a1 + a2) vs the library's")
    out = os.path.join(REPO, "gpurun_out", "r05_ref_cu_bbox_census.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(tot, open(out, "w"), indent=1)
    print(json.dumps(tot, indent=1))


if __name__ == "__main__":
    main()
