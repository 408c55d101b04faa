#!/usr/bin/env python3
"""The reference's OWN kernels through the reference's schedule at BASELINE config 2's full size.

The unfused kernels of oracle/_ref/raynet_ref_config2_nofma.co (batch_compute_similarities,
batch_voxel_traversal, batch_planes_voxels_mapping, batch_belief_propagation: the reference's .cu
text compiled unchanged, oracle/build_ref_cu.py) run the schedule of forward_pass.py:579-748 --
3 BP iterations over the 5 reference images, accumulator swapped and refilled with the prior after
each -- with the decisions of SURVEY.md section 9 that are about the DRIVER (messages persist, Q1;
rays with fewer than 2 voxels send nothing, Q4).  Two questions:

  1. does the kernel's literal message arithmetic (`cumsum1 - cumsum2`, mrf_bp.cu:157) stay finite
     at this size?  (DESIGN.md section 6 says no, from the oracle's restatement of it: here it is
     the reference's own code that answers);
  2. where it is finite, how far is the library's accumulator from it?

    gpurun -- python tools/ref_cu_fullsize.py  ->  gpurun_out/r05_ref_cu_fullsize_config2.json
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
    from oracle import oracle
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.synthetic import make_synthetic_scene
    shape = ref_cu.manifest()["shapes"]["config2"]
    M, D, N, F, H, W, pad = (shape[k] for k in ("M", "D", "N", "F", "H", "W", "padding"))
    grid, bbox = tuple(shape["grid"]), np.asarray(shape["bbox"], np.float32)
    V = N
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=F, padding=pad, focal=1.5 * H, seed=1234)
    ctx = get_context(M, D, N, F, H, W, pad, bbox, grid)
    r = ref_cu.RefCu("config2", "nofma")
    vg = r.dev(oracle.voxel_grid_centers(bbox, grid))
    n = H * W
    ridx = torch.arange(n, dtype=torch.int32, device="cuda")
    per = []
    for img in range(V):
        views = scene.view_indices_with_neighbors(img, N - 1)
        feats = bank.stacked(views)
        P = ctx.dev(np.array([scene.get_image(v).camera.P for v in views], np.float32))
        cam = scene.get_image(img).camera
        s = torch.zeros((n, 3), device="cuda")
        e = torch.zeros((n, 3), device="cuda")
        ctx.sample_rays(ridx, ctx.dev(cam.P_pinv.astype(np.float32)), ctx.dev(cam.center.ravel().astype(np.float32)), s, e)
        S = r.similarities(feats, P.reshape(-1), s, e)
        rvi, rvc = r.traversal(s, e)
        Sv = r.planes_to_voxels(vg, rvi, rvc, s, e, S)
        rvc_bp = torch.where(rvc >= 2, rvc, torch.zeros_like(rvc))          # Q4
        per.append(dict(rvi=rvi, rvc=rvc_bp, Sv=Sv, msgs=torch.zeros((n, M), device="cuda")))
        del S, feats
    prior = float(np.float32(np.log(0.05) - np.log(0.95)))
    acc = torch.full(grid, prior, device="cuda")
    rep = {"config": "config2: 5 views x 480x640 rays, 64 planes, 128^3, M=384, 3 BP iterations",
           "kernels": "oracle/_ref/raynet_ref_config2_nofma.co (the reference's .cu text, unchanged)",
           "iterations": []}
    for it in range(3):
        out = torch.full(grid, prior, device="cuda")
        for st in per:
            r.bp_sweep(st["Sv"].clone(), st["rvi"], st["rvc"], acc, st["msgs"], out)     # S is clipped in place
        acc = out
        torch.cuda.synchronize()
        valid = [torch.arange(M, device="cuda")[None, :] < st["rvc"][:, None] for st in per]
        bad_m = int(sum(int((~torch.isfinite(st["msgs"]) & v).sum()) for st, v in zip(per, valid)))
        rep["iterations"].append({"iteration": it + 1, "non_finite_messages": bad_m,
                                  "non_finite_accumulator_voxels": int((~torch.isfinite(acc)).sum()),
                                  "max_abs_finite_accumulator": float(acc[torch.isfinite(acc)].abs().max())})
        print(rep["iterations"][-1], flush=True)
    acc_ref = acc.cpu().numpy()
    msgs_ref0 = per[0]["msgs"].cpu().numpy()
    rvc0 = per[0]["rvc"].cpu().numpy()
    # the reference's depth maps: batch_depth_estimation, first arg-max, distance of that voxel's
    # centre to the camera (raynet_fp.py:193-226)
    depth_ref = []
    vgf = vg.reshape(-1, 3)
    for img, st in enumerate(per):
        S_new = r.depth_estimation(st["Sv"].clone(), st["rvi"], st["rvc"], acc, st["msgs"])
        mx = S_new.max(1, keepdim=True).values
        first = (S_new == mx).to(torch.int32).argmax(1)                      # first maximum
        v = st["rvi"][torch.arange(n, device="cuda"), first].long()
        centre = vgf[(v[:, 0] * grid[1] + v[:, 1]) * grid[2] + v[:, 2]]
        cc = ctx.dev(scene.get_image(img).camera.center.ravel().astype(np.float32))[:3]
        dd = torch.sqrt(((centre - cc) ** 2).sum(1))
        dd = torch.where(torch.isfinite(S_new).all(1), dd, torch.full_like(dd, float("nan")))
        depth_ref.append(dd.cpu().numpy().reshape(W, H).T)
        del S_new
    depth_ref = np.stack(depth_ref)
    # the oracle's restatement of the kernel's LITERAL arithmetic on the same columns and lists:
    # is it the kernel?  (one image's first sweep would do; all three iterations cost ~25 s of CPU)
    o = oracle.Oracle(M=M, D=D, N=N, F=F, H=H, W=W, padding=pad, bbox=bbox, grid_shape=grid,
                      threads=oracle.Oracle.max_threads())
    oracle.Oracle.set_robust_messages(False)
    host = [dict(rvi=st["rvi"].cpu().numpy(), rvc=st["rvc"].cpu().numpy(), Sv=st["Sv"].cpu().numpy(),
                 msgs=np.zeros((n, M), np.float32)) for st in per]
    acc_o = o.prior(0.05)
    for it in range(3):
        out_o = o.prior(0.05)
        for st in host:
            o.bp_sweep(st["Sv"], st["rvi"], st["rvc"], acc_o, st["msgs"], out_o)
        acc_o = out_o
    both = np.isfinite(acc_o) & np.isfinite(acc_ref)
    rep["oracle_literal_form_vs_the_kernel"] = {
        "non_finite_voxels_oracle": int((~np.isfinite(acc_o)).sum()),
        "non_finite_voxels_kernel": int((~np.isfinite(acc_ref)).sum()),
        "same_voxels_non_finite": bool(np.array_equal(np.isfinite(acc_o), np.isfinite(acc_ref))),
        "non_finite_in_both": int((~np.isfinite(acc_o) & ~np.isfinite(acc_ref)).sum()),
        "max_abs_accumulator_diff_where_both_finite": float(np.abs(acc_o - acc_ref)[both].max()),
        "voxels_beyond_1e-2": int((np.abs(acc_o - acc_ref)[both] > 1e-2).sum()),
        "non_finite_messages_oracle": int(sum((~np.isfinite(st["msgs"])).sum() for st in host))}
    print(rep["oracle_literal_form_vs_the_kernel"], flush=True)
    del per, host
    torch.cuda.empty_cache()
    # the library on the same scene
    gp = GenerationParameters(depth_planes=D, neighbors=N - 1, grid_shape=np.array(grid, np.int32),
                              max_number_of_marched_voxels=M, padding=pad, gamma_mrf=0.05)
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
    depth = np.stack(list(fp.forward_pass(scene, (0, V, 1))))
    acc_hip = fp.accumulator.cpu().numpy()
    fin = np.isfinite(acc_ref)
    d = np.abs(acc_hip - acc_ref)[fin]
    rows = fp.messages[0].cpu().numpy()
    m_hip = np.zeros_like(rows)
    m_hip[fp.ray_index[0].cpu().numpy().astype(np.int64)] = rows
    ok = np.isfinite(msgs_ref0) & (np.arange(M)[None, :] < rvc0[:, None])
    dm = np.abs(m_hip - msgs_ref0)[ok]
    dz = np.abs(depth - depth_ref)
    rep["depth_maps"] = {"pixels": int(depth.size), "reference_pixels_non_finite": int(np.isnan(depth_ref).sum()),
                         "pixels_beyond_1e-4": int((dz > 1e-4).sum()),
                         "fraction_beyond_1e-4": float((dz > 1e-4).mean())}
    rep["library"] = {"accumulator_finite_everywhere": bool(np.isfinite(acc_hip).all()),
                      "depth_maps_finite": bool(np.isfinite(depth).all()),
                      "max_abs_accumulator_diff_on_the_reference_s_finite_voxels": float(d.max()),
                      "voxels_beyond_1e-2": int((d > 1e-2).sum()), "voxels_compared": int(fin.sum()),
                      "image0_messages_max_abs_diff_on_finite_entries": float(dm.max()),
                      "image0_messages_beyond_1e-3": int((dm > 1e-3).sum()), "image0_messages_compared": int(ok.sum()),
                      "note": "the library sums the suffix directly (log pos - log neg), the kernel forms "
                              "cumsum1 - cumsum2 in fp32: next to a saturated voxel the latter cancels, so "
                              "the two legitimately differ there (DESIGN.md section 6; both are held to the "
                              "float64 value of a sweep in tests/test_reference_kernels.py)"}
    out = os.path.join(REPO, "gpurun_out", "r05_ref_cu_fullsize_config2.json")
    json.dump(rep, open(out, "w"), indent=1)
    print(json.dumps(rep["depth_maps"]))
    print(json.dumps(rep["library"], indent=1))


if __name__ == "__main__":
    main()
