#!/usr/bin/env python3
"""The reference's own kernels (oracle/_ref/*.co, see oracle/build_ref_cu.py) on the bench scene:

 1. how many rays the contracted build (`fma`, what nvcc / PyCUDA's default would run) moves
    against the -ffp-contract=off build (the oracle's and this library's convention) -- the census
    VERDICT r4 asked for, over ALL rays of one reference image of config 2;
 2. the library's two plane sweeps against both builds on the same rays;
 3. one timed launch of every reference kernel at config 2's size on this GPU: the same-GPU
    reference-DESIGN baseline (thread per ray, everything in global / scratch memory) that
    BASELINE.md says was never run.  Recorded under profiles/, not part of bench.py's line.

Run on the GPU box: `gpurun -- python tools/ref_cu_report.py [--config config2] > ...`.
"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config2", choices=["config2", "config4"])
    ap.add_argument("--image", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "r05_ref_cu_report.json"))
    args = ap.parse_args()
    import torch
    import ref_cu
    from oracle import oracle
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.synthetic import make_synthetic_scene

    shape = ref_cu.manifest()["shapes"][args.config]
    M, D, N, F, H, W, pad = (shape[k] for k in ("M", "D", "N", "F", "H", "W", "padding"))
    grid, bbox = shape["grid"], np.asarray(shape["bbox"], np.float32)
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=N, F=F, padding=pad, focal=1.5 * H, seed=1234)   # bench.py's scene
    views = scene.view_indices_with_neighbors(args.image, N - 1)
    feats = bank.stacked(views)
    P = np.array([scene.get_image(v).camera.P for v in views], np.float32)
    cam = scene.get_image(args.image).camera
    P_inv, center = cam.P_pinv.astype(np.float32), cam.center.ravel().astype(np.float32)
    ridx = np.arange(H * W, dtype=np.int32)
    o = oracle.Oracle(M=M, D=D, N=N, F=F, H=H, W=W, padding=pad, bbox=bbox, grid_shape=grid,
                      threads=oracle.Oracle.max_threads())
    starts, ends = o.sample(ridx, P_inv, center)
    vg = oracle.voxel_grid_centers(bbox, grid)
    n = len(ridx)
    rep = {"config": args.config, "image": args.image, "rays": n, "flags": ref_cu.manifest()["flags"]}

    mods = {v: ref_cu.RefCu(args.config, v) for v in ("nofma", "fma")}
    r = mods["nofma"]
    P_d, s_d, e_d = r.dev(P.reshape(-1)), r.dev(starts), r.dev(ends)
    S = {v: m.similarities(feats, P_d, s_d, e_d) for v, m in mods.items()}
    d = (S["fma"] - S["nofma"]).abs().max(1).values
    hits = (torch.from_numpy(np.asarray(starts != ends)).any(1)).to("cuda")
    rep["contraction"] = {
        "rays_moved_gt_1e-5": int((d > 1e-5).sum()), "rays_moved_gt_1e-4": int((d > 1e-4).sum()),
        "rays_moved_gt_1e-3": int((d > 1e-3).sum()), "largest_move": float(d.max()),
        "median_move": float(d.median()), "fraction_gt_1e-5": float((d > 1e-5).float().mean()),
        "note": "max over the D planes of |S_fma - S_nofma| per ray; a ray moves by more than 1e-5 "
                "when one of its N*D projections rounds to another pixel under contraction"}
    # the library's sweeps against both builds
    ctx = get_context(M, D, N, F, H, W, pad, bbox, grid)
    rep["hip_vs_reference"] = {}
    for name, generic in (("cooperative", False), ("generic", True)):
        os.environ["RAYNET_HIP_GENERIC_SWEEP"] = "1" if generic else "0"
        from raynet_amd.hip_implementations.options import PathOptions
        ctx.set_options(PathOptions.from_env())
        Sh = torch.zeros((n, D), device="cuda")
        ctx.compute_similarities(feats, P_d, s_d, e_d, Sh)
        e_no = (Sh - S["nofma"]).abs().max(1).values
        e_f = (Sh - S["fma"]).abs().max(1).values
        rep["hip_vs_reference"][name] = {
            "max_abs_vs_nofma_build": float(e_no.max()),
            "rays_gt_1e-5_vs_nofma_build": int((e_no > 1e-5).sum()),
            "rays_gt_1e-5_vs_fma_build": int((e_f > 1e-5).sum()),
            "max_abs_vs_fma_build_on_unmoved_rays": float(e_f[d <= 1e-5].max())}
    os.environ.pop("RAYNET_HIP_GENERIC_SWEEP", None)
    So = o.similarities(feats.cpu().numpy(), P, starts[:4096], ends[:4096])
    rep["oracle_vs_nofma_build_first_4096_rays"] = float(np.abs(So - S["nofma"][:4096].cpu().numpy()).max())

    # ---- timings of the reference kernels, one image of the configuration -----------------
    t = {}
    Sz = torch.zeros((n, D), device="cuda")
    t["batch_compute_similarities"] = r.timed("batch_compute_similarities", n, feats, P_d, s_d, e_d, Sz)
    rvi = torch.zeros((n, M, 3), dtype=torch.int32, device="cuda")
    rvc = torch.zeros((n,), dtype=torch.int32, device="cuda")
    t["batch_voxel_traversal"] = r.timed("batch_voxel_traversal", n, s_d, e_d, rvi, rvc)
    vg_d = r.dev(vg)
    Sv = torch.zeros((n, M), device="cuda")
    t["batch_planes_voxels_mapping"] = r.timed("batch_planes_voxels_mapping", n, vg_d, rvi, rvc, s_d, e_d,
                                               S["nofma"], Sv)
    # rays with one voxel write +inf: give them none, as the NumPy path does
    rvc_bp = torch.where(rvc >= 2, rvc, torch.zeros_like(rvc))
    prior = float(np.float32(np.log(0.05) - np.log(0.95)))
    acc_in = torch.full(tuple(grid), prior, device="cuda")
    acc_out = torch.full(tuple(grid), prior, device="cuda")
    msgs = torch.zeros((n, M), device="cuda")
    t["batch_belief_propagation"] = r.timed("batch_belief_propagation", n, Sv.clone(), rvi, rvc_bp, acc_in,
                                            msgs, acc_out, msgs, repeats=2)
    S_new = torch.zeros((n, M), device="cuda")
    t["batch_depth_estimation"] = r.timed("batch_depth_estimation", n, Sv.clone(), rvi, rvc_bp, acc_out,
                                          msgs, S_new)
    pts = torch.zeros((n, D, 4), device="cuda")
    t["batch_sample_points_in_bbox"] = r.timed("batch_sample_points_in_bbox", n, r.dev(ridx), r.dev(P_inv.ravel()),
                                               r.dev(center), pts)
    # the fused K1 as the reference launches it (raynet_fp.py:106-149); its thread-local S[D] is
    # read uninitialised, so only its duration means anything
    ridx_d = r.dev(ridx)
    rvi.zero_(); rvc.zero_(); Sv.zero_(); msgs.zero_()
    t["batch_raynet_fp (fused K1)"] = r.timed(
        "batch_raynet_fp", n, ridx_d, feats, P_d, r.dev(P_inv.ravel()), r.dev(center), vg_d, rvi, rvc, Sv,
        acc_in, msgs, acc_out, msgs, repeats=2)
    depth = torch.zeros((n,), device="cuda")
    t["batch_complete_depth_estimation (fused K2)"] = r.timed(
        "batch_complete_depth_estimation", n, ridx_d, feats, P_d, r.dev(P_inv.ravel()), r.dev(center), vg_d,
        rvi, rvc, Sv, acc_out, msgs, depth, repeats=2)
    rep["reference_kernels_ms_per_image"] = {k: round(v, 3) for k, v in t.items()}
    V = N
    k1, k2 = t["batch_raynet_fp (fused K1)"], t["batch_complete_depth_estimation (fused K2)"]
    step = V * (3 * k1 + k2)
    rep["reference_design_step"] = {
        "schedule": "forward_pass.py:593-748: per reference image 3 x fused K1 + 1 x fused K2, "
                    "everything recomputed in every sweep; kernels only (no CNN, no memmap, no H2D)",
        "ms_per_step": round(step, 2), "rays_per_s": round(V * n / step * 1e3, 1),
        "mean_voxels_per_ray": float(rvc.float().mean())}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(rep, fh, indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
