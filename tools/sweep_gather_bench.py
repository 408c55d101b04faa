#!/usr/bin/env python3
"""Drives tools/sweep_gather_bench.hip: the plane sweep's feature gathers ALONE, in the product's
order and in candidate orders (lane = plane; 16 x 16-ray tiles walking the planes in bands), on the
real offsets of one reference image of the bench scene.  See the .hip file for the question.

    python tools/sweep_gather_bench.py --config config4 [--image 0] [--variants 0,1,2,...]
    rocprofv3 --pmc TCC_MISS_sum TCC_REQ_sum -d out -- python tools/sweep_gather_bench.py ... --once
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
HERE = os.path.dirname(os.path.abspath(__file__))

VARIANTS = {0: "ray order, 8 lanes per vector (the product's)", 1: "ray order, lane = plane",
            2: "16x16 tile, bands of 16 planes, 16 waves", 3: "tile, bands of 8, 16 waves",
            4: "tile, bands of 32, 16 waves", 5: "tile, bands of 16, 8 waves",
            6: "tile, bands of 64, 16 waves (control: ray order inside a synchronised tile)",
            20: "FOOTPRINT-STAGED tile: bands of 8, row segments by LDS-DMA, vector reads from LDS",
            21: "footprint-staged tile, bands of 4",
            22: "footprint-staged tile, bands of 16",
            23: "footprint-staged tile, bands of 8, 8 waves",
            10: "ray order, all 32 loads of a chunk in flight, default cache policy",
            11: "... nt", 12: "... sc0", 13: "... sc1", 14: "... sc0 sc1", 15: "... sc0 sc1 nt"}


def build():
    so = os.path.join(HERE, "libsweep_gather_bench.so")
    src = os.path.join(HERE, "sweep_gather_bench.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-shared", src, "-o", so])
    return so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4", choices=["config2", "config4"])
    ap.add_argument("--image", type=int, default=0)
    ap.add_argument("--variants", default="0,1,2,3,4,5,6")
    ap.add_argument("--lds", type=int, default=131072, help="dynamic LDS of the tile kernels (occupancy cap)")
    ap.add_argument("--ray-lds", type=int, default=0, help="dynamic LDS of the ray-order kernel (occupancy cap)")
    ap.add_argument("--stage", type=int, default=155000, help="LDS bytes the footprint-staged tile may fill")
    ap.add_argument("--chunk", type=int, default=512, help="workgroups side by side on one XCD (ray order)")
    ap.add_argument("--tile-chunk", type=int, default=8, help="tiles side by side on one XCD")
    ap.add_argument("--once", action="store_true", help="one launch per variant (for counter passes)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import bench
    from raynet_amd.forward_pass import sweep_direction, tile_order
    from raynet_amd.hip_implementations import get_context
    from raynet_amd.synthetic import make_synthetic_scene
    cfg = bench.CONFIGS[args.config]
    H, W, V, D, M, F, pad = (cfg[k] for k in ("H", "W", "views", "D", "M", "F", "padding"))
    N = V if V > 5 else 5
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=F, padding=pad, focal=1.5 * H, seed=1234)
    ctx = get_context(M, D, N, F, H, W, pad, scene.bbox.ravel(), cfg["grid"])
    views = scene.view_indices_with_neighbors(args.image, N - 1)
    images = [scene.get_image(v) for v in views]
    along = sweep_direction(H, W, images) == "rows"
    rays = tile_order(torch.arange(H * W, dtype=torch.int32, device="cuda"), H, W, 16, 16, along_rows=along)
    n = len(rays)
    P = np.ascontiguousarray(np.array([im.camera.P for im in images], np.float32))
    cam0 = images[0].camera
    P_inv, center = cam0.P_pinv.astype(np.float32), cam0.center.ravel().astype(np.float32)
    s = torch.zeros((n, 3), device="cuda")
    e = torch.zeros((n, 3), device="cuda")
    ctx.sample_rays(rays, ctx.dev(P_inv), ctx.dev(center), s, e)
    both = torch.zeros((n, N, D, 2), dtype=torch.int32, device="cuda")
    ctx.selftest_feature_offsets(ctx.dev(P), s, e, both)
    offs = both[..., 1].contiguous()
    del both
    cam = np.zeros((1, 12 * N + 16), np.float32)
    cam[0, :12 * N], cam[0, 12 * N:12 * N + 12], cam[0, 12 * N + 12:] = P.ravel(), P_inv.ravel(), center
    live = ctx.count_voxels(rays, ctx.dev(cam))[0].contiguous()
    maps = [bank.view_features(scene, v) for v in views]
    ptrs = (ctypes.c_void_p * N)(*[m.data_ptr() for m in maps])
    lib = ctypes.CDLL(build())
    lib.sgb_run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
    lib.sgb_set_lds_variant.argtypes = [ctypes.c_int, ctypes.c_void_p]
    stats = torch.zeros((4,), dtype=torch.int32, device="cuda")
    lib.sgb_set_lds_variant(W + pad + 1, stats.data_ptr())
    out = torch.zeros((max(n * 64, ((n + 255) // 256) * 1024),), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    n_live = int((live > 1).sum())
    gathered = n_live * (N - 1) * D * 128
    unique = sum(int(torch.unique(offs[live > 1][:, v]).numel()) for v in range(1, N)) * 128
    rep = {"config": args.config, "image": args.image, "rays": n, "live_rays": n_live, "views": N, "planes": D,
           "gathered_GB": round(gathered / 1e9, 3), "unique_GB": round(unique / 1e9, 4),
           "lds_bytes_tile_kernels": args.lds, "variants": {}}
    for v in [int(x) for x in args.variants.split(",")]:
        chunk = args.chunk if (v < 2 or 10 <= v < 20) else args.tile_chunk
        lds_v = args.stage if v >= 20 else (args.lds if 2 <= v < 10 else args.ray_lds)

        def run():
            rc = lib.sgb_run(v, n, N, D, offs.data_ptr(), live.data_ptr(), ptrs, out.data_ptr(), chunk,
                             lds_v, stream)
            assert rc == 0, rc
        if args.once:
            run()
            torch.cuda.synchronize()
            continue
        run()
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            run()
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b))
        rep["variants"][str(v)] = {"what": VARIANTS[v], "ms": round(best, 3),
                                   "L1_side_TBps": round(gathered / best / 1e9, 2)}
        if v >= 20:
            stats.zero_()
            run()
            torch.cuda.synchronize()
            st = stats.cpu().numpy().astype(np.int64)
            rep["variants"][str(v)].update(
                bands=int(st[0]), bands_gathering_from_global=int(st[1]), staged_pixels=int(st[2]) * 16,
                staged_GB=round(int(st[2]) * 16 * 128 / 1e9, 3), stage_bytes=lds_v,
                checksum=float(out[:((n + 255) // 256) * 1024].double().sum()))
            # what the staged reads must add up to: every live ray's vectors, straight from the maps
            expect = 0.0
            for vv in range(1, N):
                vs = maps[vv].reshape(-1, F).double().sum(1)
                expect += float(vs[offs[live > 1][:, vv].reshape(-1).long()].sum())
            rep["variants"][str(v)]["checksum_expected"] = expect
        print("variant %d  %-75s %8.3f ms   %6.2f TB/s of gathered vectors" % (v, VARIANTS[v], best,
                                                                                gathered / best / 1e9), flush=True)
    if not args.once:
        print(json.dumps(rep))
        if args.out:
            with open(args.out, "w") as fh:
                json.dump(rep, fh, indent=1)


if __name__ == "__main__":
    main()
