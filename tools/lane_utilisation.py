#!/usr/bin/env python3
"""VERDICT r5 item 5 ("spend the 29 % idle lanes"): how full the 64-lane chunks of the per-voxel
phases really are at a configuration -- from the traversal's own voxel counts (rn_scene_count_voxels,
bit-exact integers) -- and what perfect packing of the rays' last chunks could buy, priced with the
measured shares of the kernels that walk voxels (profiles/r05_sweep_phase_budget_config2.json for the
plane sweep's tail, the bench line's family times for k_bp / k_depth).

    python tools/lane_utilisation.py --config config2 --bench profiles/r06_a_bench.json
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
    ap.add_argument("--config", default="config2")
    ap.add_argument("--bench", default=os.path.join(REPO, "profiles", "r06_a_bench.json"))
    args = ap.parse_args()
    import torch
    import bench
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.synthetic import make_synthetic_scene
    cfg = bench.CONFIGS[args.config]
    H, W, V, D, M, F, pad = (cfg[k] for k in ("H", "W", "views", "D", "M", "F", "padding"))
    N = V if V > 5 else 5
    scene, _ = make_synthetic_scene(H=H, W=W, n_views=V, F=F, padding=pad, focal=1.5 * H, seed=1234)
    ctx = get_context(M, D, N, F, H, W, pad, scene.bbox.ravel(), cfg["grid"])
    rays = torch.arange(H * W, dtype=torch.int32, device="cuda")
    cam = np.zeros((V, 12 * N + 16), np.float32)
    for r in range(V):
        views = scene.view_indices_with_neighbors(r, N - 1)
        images = [scene.get_image(v) for v in views]
        cam[r, :12 * N] = np.array([im.camera.P for im in images], np.float32).ravel()
        cam[r, 12 * N:12 * N + 12] = images[0].camera.P_pinv.astype(np.float32).ravel()
        cam[r, 12 * N + 12:12 * N + 15] = images[0].camera.center.ravel()[:3]
    counts = ctx.count_voxels(rays, ctx.dev(cam)).cpu().numpy().astype(np.int64).ravel()
    counts = np.minimum(counts, M)
    live = counts[counts > 1]
    chunks = (live + 63) // 64
    slots = 64 * chunks
    hist = np.bincount(chunks, minlength=M // 64 + 2)
    util = live.sum() / slots.sum()
    # perfect packing of last chunks: every ray keeps its full chunks, the partial ones are filled
    # pairwise at best (two tails per wavefront chunk): slots >= 64 * (full chunks + ceil(tails / 2))
    tails = live % 64
    packed_slots = 64 * ((live // 64).sum() + np.ceil((tails > 0).sum() / 2.0))
    rep = {"config": args.config, "rays": int(len(counts)), "rays_with_voxels": int(len(live)),
           "mean_voxels_per_live_ray": round(float(live.mean()), 2),
           "chunks_per_live_ray": round(float(chunks.mean()), 3),
           "rays_by_chunk_count": {str(i): int(h) for i, h in enumerate(hist) if h},
           "lane_utilisation": round(float(util), 4),
           "lane_utilisation_if_every_two_tails_shared_a_chunk": round(float(live.sum() / packed_slots), 4),
           "chunk_instructions_saved_by_that_packing": round(float(1 - packed_slots / slots.sum()), 4)}
    try:
        line = json.loads(open(args.bench).read().strip().splitlines()[-1])
        k = line["kernels"]
        sweep = k["sweep_map"]["total_ms_per_step"]
        # the sweep's per-voxel phases (mapping 185 + clip 29 + first BP iteration 163 of 921 VALU
        # instructions per ray: profiles/r05_sweep_phase_budget_config2.json), the only VALU-bound ones
        tail = sweep * (185 + 29 + 163) / 921.0
        save = rep["chunk_instructions_saved_by_that_packing"]
        rep["priced"] = {
            "bench_line": os.path.basename(args.bench), "ms_per_step": line["ms_per_step"],
            "sweep_tail_ms (VALU-bound: chunks = instructions)": round(tail, 3),
            "upper_bound_saved_in_sweep_tail_ms": round(tail * save, 3),
            "k_bp_ms": k["bp"]["total_ms_per_step"], "k_depth_ms": k["depth"]["total_ms_per_step"],
            "k_bp_k_depth_note": "memory-bound (rows + gathers take the kernels' whole time with no "
                                 "arithmetic at all: profiles/r02_exp_bp_ablation.txt): idle lanes of a "
                                 "last chunk issue no memory requests -- packing them saves VALU slots "
                                 "these kernels do not wait for",
            "upper_bound_fraction_of_step": round(tail * save / line["ms_per_step"], 4)}
    except Exception as e:
        rep["priced"] = {"error": repr(e)}
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
