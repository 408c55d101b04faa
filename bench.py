#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: rays/sec through RayNet's forward_pass hot path
at 5 views x 64 depth planes x 128^3 voxels, on N MI355X of one node.

One "step" = one complete pass of the hot path over the synthetic scene
(SURVEY.md 8d): for each of the V=5 reference images (480x640 rays each) the K1
prefix (ray sampling, plane sweep + softmax, voxel traversal, planes->voxels
mapping), then 3 BP sweeps over all images with the accumulator hand-over after each
(plus one RCCL all-reduce per iteration when N>1), then the depth sweep, through the
public RayNetForwardPass.forward_pass generator (depth maps are copied back to the
host like the reference's `.get()`).  Feature maps are resident in HBM when the timed
region starts (the MV-CNN is outside the path).  The timed steps are passes over a cached PLAN
of the scene (what depends on cameras, image range and sharding only: `config.reused_between_
steps` lists it); no result of a pass is reused -- every step recomputes traversal, plane sweep,
mapping, the BP iterations and the depth sweep.  The cost of the FIRST pass over a new scene
(plan construction included) is reported as `first_pass_ms`.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus 8 --steps 5 --warmup 1

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     dominant kernel family: algorithmic bytes of its
               launches in the timed region / their hipEvent durations (per launch, on
               the launch stream, via rn_prof_begin/rn_prof_end), vs the 8 TB/s HBM peak;
               `traffic` (HBM-side bytes) and `valu_insts_per_launch` from three rocprofv3
               --pmc passes of this script spawned after the timed region (--pmc live, N = 1),
               `bound` = "valu" when the VALU-issue time exceeds the HBM time
  kernels      every family's share of a step, from up to 3 untimed steps before the timed
               region with every launch bracketed (the timed region brackets the dominant
               family's launches only, unless --events all): the sum of its launches'
               durations, and its share of the timeline (concurrent launches on the second
               stream split the time they overlap; the dominant family is chosen by this)
  cpu_baseline two legs on bounded ray samples of the same scene (rays of all reference
               images), on this box's host cores: "port" = the C oracle's fused K1/K2 path
               (the reference's algorithm, OpenMP over rays, all cores) -- also the object's
               top-level value; "numpy_restatement" = the reference's own NumPy
               implementation of the MRF stages (mrf/mrf_np.py semantics restated in
               oracle/cpu_reference.py, per-ray Python loop, one process)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

CONFIGS = {
    # BASELINE.json configs[1] / [2]: the configuration the metric is quoted on
    "config2": dict(H=480, W=640, views=5, D=64, M=384, grid=(128, 128, 128), F=32, padding=11,
                    workload="synthetic 5-view scene, 480x640 rays/view, 64 depth planes, "
                             "128^3 voxels, M=384, 3 BP iterations + depth sweep"),
    # BASELINE.json configs[3] (DTU-style), for reference runs
    "config4": dict(H=480, W=640, views=9, D=128, M=768, grid=(256, 256, 256), F=32, padding=11,
                    workload="synthetic 9-view scene, 640x480, 128 depth planes, 256^3 voxels"),
    "small": dict(H=120, W=160, views=5, D=64, M=384, grid=(128, 128, 128), F=32, padding=11,
                  workload="reduced 120x160 (debug only)"),
    # one eighth of config 2's rays per image at the same pixel density: what one rank of an
    # 8-GPU run has to do (debug only; no collectives)
    "eighth": dict(H=480, W=80, views=5, D=64, M=384, grid=(128, 128, 128), F=32, padding=11,
                   workload="480x80 strip of config 2 (debug only)"),
    "quarter": dict(H=480, W=160, views=5, D=64, M=384, grid=(128, 128, 128), F=32, padding=11,
                    workload="480x160 strip of config 2 (debug only)"),
    "half": dict(H=480, W=320, views=5, D=64, M=384, grid=(128, 128, 128), F=32, padding=11,
                 workload="480x320 strip of config 2 (debug only)"),
}

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


VALU_CLOCK_HZ = 2.4e9     # nominal shader clock; 256 CUs x 4 SIMDs
N_SIMDS = 1024
MEASURED_COPY_GBS = 6290.0    # MI355X_MICROARCH.md: measured device-to-device copy rate


def valu_cycles_per_inst(family, config):
    """Cycles one SIMD needs to issue one wave64 VALU instruction of this kernel: the MEASURED cost
    of each instruction class (tools/valu_issue_bench.hip: plain fp32 / integer add, mul, fma,
    logic ~2.3 - 2.9; packed, DPP, min / max / med3, shifts, conversions, compares, 24-bit mad ~4.1;
    transcendentals ~8.1) weighted with the kernel's static instruction histogram
    (tools/valu_mix.py -> profiles/valu_issue.json).  -> (cycles, source)"""
    path = os.path.join(REPO, "profiles", "valu_issue.json")
    try:
        table = json.load(open(path))["kernels"]
        for key in ("%s/%s" % (family, config), family + "/steady", family + "/box128x32", family):
            if key in table:
                return float(table[key]["mean_cycles"]), "profiles/valu_issue.json [%s]" % key
    except Exception:
        pass
    return 4.0, "default (no profiles/valu_issue.json entry)"


def algorithmic_bytes(kernel, n_rays, voxels, cfg, images=1, folded=False, strict=False):
    """Algorithmic HBM bytes of ONE launch.  voxels = sum of the
    per-ray voxel counts of the rays in the launch; images = reference images it covers.
    folded: the plane sweep also writes BP iteration 0's messages (4 B per voxel).
    strict: SURVEY.md 8(d)'s own figure for the plane sweep -- the N feature maps once per
    reference image and nothing else (a fused design need not materialise lists or columns)."""
    N, F = cfg["views"], cfg["F"]
    Hf, Wf = cfg["H"] + cfg["padding"] + 1, cfg["W"] + cfg["padding"] + 1
    if kernel == "traverse":      # write packed voxel list + count + ray segment, read ray index
        return 4 * voxels + 40 * n_rays
    if kernel == "sweep_map":     # N feature maps once; read voxel list + segment, write column
        if strict:
            return images * 4 * N * F * Hf * Wf
        return images * 4 * N * F * Hf * Wf + (12 if folded else 8) * voxels + 40 * n_rays
    if kernel == "bp":            # Sr, voxel list, msg in, acc gather, msg out
        return 20 * voxels + 4 * n_rays
    if kernel == "scatter":       # msg, voxel list (the HBM-side streams); the atomic RMW of the
        # accumulator (8 B per visit in SURVEY.md 8(d)'s count) never leaves LDS / L2 -- the LDS box
        # sums ~11 visits per voxel, the 8.4 MB accumulator lives in the L2 -- and is reported
        # on its own (cache_resident_bytes), not as HBM traffic
        return 8 * voxels + 4 * n_rays
    if kernel == "depth":         # Sr, voxel list, msg, acc gather; depth out
        return 16 * voxels + 8 * n_rays
    return 0


def cache_resident_bytes(kernel, voxels):
    """Bytes of SURVEY.md 8(d)'s per-visit model that stay in LDS / L2 by construction (not HBM
    traffic): the scatter's read-modify-write of the accumulator, 8 B per visit."""
    return 8 * voxels if kernel == "scatter" else 0


KERNEL_OF_FAMILY = {"sweep_map": "k_sweep_map", "bp": "k_bp", "scatter": "k_scatter_box",
                    "depth": "k_depth", "traverse": "k_traverse"}


def live_pmc(config, family, budget_s=180.0):
    """HBM-side traffic, VALU instructions + the cycles the VALUs were busy with them, and the L2's
    requests / misses of the dominant kernel, per launch, measured NOW: four rocprofv3 --pmc passes
    (FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU + SQ_ACTIVE_INST_VALU / TCC_REQ + TCC_MISS: each
    counter set in its own run, kernel-trace only, as MI355X_MICROARCH.md prescribes) of this very
    script with --steps 1, spawned from here.  -> dict or None (no rocprofv3, a pass failed, out
    of time)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    kernel = KERNEL_OF_FAMILY.get(family)
    if not os.path.exists(rocprof) or kernel is None:
        return None
    out = {}
    t_start = time.perf_counter()
    env = dict(os.environ, TMPDIR="/tmp")
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"],
                     ["TCC_REQ_sum", "TCC_MISS_sum"]):
        left = budget_s - (time.perf_counter() - t_start)
        if left < 20:
            return None
        d = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        try:
            cmd = [rocprof, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv",
                                                  "-d", d, "-o", "p", "--", sys.executable,
                                                  os.path.abspath(__file__), "--steps", "1",
                                                  "--warmup", "1", "--no-cpu-baseline", "--no-config4",
                                                  "--no-reference-shapes",     # (their sweeps would
                                                  # be averaged into the counters per launch)
                                                  "--pmc", "off", "--config", config]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=min(left, 60.0))
            if r.returncode != 0:
                if counters[0].startswith("TCC"):
                    continue
                return None
            tot, n, everything = {}, {}, {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        c = row["Counter_Name"]
                        # (every kernel of the child's passes, the library's and torch's few alike:
                        # the whole path's VALU time per pass, see `path_valu` below)
                        everything[c] = everything.get(c, 0.0) + float(row["Counter_Value"])
                        if kernel in row.get("Kernel_Name", ""):
                            tot[c] = tot.get(c, 0.0) + float(row["Counter_Value"])
                            n[c] = n.get(c, 0) + 1
            for c in counters:
                if not n.get(c):
                    if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
                        return None
                    continue            # (the newer, optional counters)
                out[c] = tot[c] / n[c]
                out["launches"] = n[c]
                if c.startswith("SQ_") and family == "sweep_map":
                    # one launch of the plane sweep per pass: all kernels' total per PASS
                    out["path_" + c] = everything[c] / n[c]
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out["seconds"] = time.perf_counter() - t_start
    return out


def timeline_shares(launches, starts):
    """launches: (family, n_rays, duration ms) per launch; starts: its start (ms, one clock).
    -> {family: ms}: an instant with k launches in flight gives each of them 1 / k of it.
    Without start times: the plain sums of the durations."""
    out = {}
    if len(starts) != len(launches) or not launches:
        for name, _, ms in launches:
            out[name] = out.get(name, 0.0) + ms
        return out
    edges = sorted({t for (_, _, ms), st in zip(launches, starts) for t in (st, st + ms)})
    iv = sorted((st, st + ms, name) for (name, _, ms), st in zip(launches, starts))
    active, nxt = [], 0
    for a, b in zip(edges[:-1], edges[1:]):
        while nxt < len(iv) and iv[nxt][0] <= a:
            active.append(iv[nxt])
            nxt += 1
        active = [x for x in active if x[1] > a]
        for x in active:
            out[x[2]] = out.get(x[2], 0.0) + (b - a) / len(active)
    return out


def reference_shapes(torch, steps=5):
    """The two workloads the reference itself names, as side figures of the default run:

    k10_getting_started  the ONLY timing the reference publishes (docs/getting-started.md:105-157):
        `raynet_forward --forward_pass_factory multi_view_cnn` on 1280 x 720 images, 5 views,
        D = 32, F = 32 -- "Per-pixel depth estimation" 64.7 - 67.2 ms per reference image on a
        TITAN X (Pascal), i.e. kernel K10 (similarities.py:168-230) over 921,600 rays in batches.
        Here: the mock Restrepo scene's cameras (1280 x 720, tests/golden), random feature maps,
        the same closure (perform_multi_view_cnn_forward_pass_with_depth_estimation) over all rays
        of a reference image in one launch (event-timed), and the MultiViewCNNForwardPass driver's
        own loop (wall clock, map copied to the host, rays_batch = the reference's 130,000).
    cli_defaults  the full path (RayNetForwardPass) at the reference's command-line defaults
        (scripts/arguments.py:154, 215, 221: D = 32, grid 256 x 256 x 128, M = 650, 4 neighbours),
        five reference images of the same cameras: 921,600 rays per image.
    Different hardware, synthetic features: the published figure is quoted for orientation only."""
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.common.scene import restrepo_cameras_scene
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.hip_implementations.similarities import \
        perform_multi_view_cnn_forward_pass_with_depth_estimation
    from raynet_amd.synthetic import FeatureBank
    H, W, D, F, pad, nb = 720, 1280, 32, 32, 11, 4
    scene = restrepo_cameras_scene(os.path.join(REPO, "tests", "golden", "restrepo_mock_scene_1"),
                                   (H, W), n_images=8, channels=1)
    g = torch.Generator(device="cuda").manual_seed(7)
    bank = FeatureBank([torch.randn((H + pad + 1, W + pad + 1, F), generator=g, device="cuda") * 0.25
                        for _ in range(scene.n_images)])
    out = {}
    # ---- K10 ---------------------------------------------------------------------------------
    k10 = perform_multi_view_cnn_forward_pass_with_depth_estimation(D, nb + 1, F, H, W, pad,
                                                                    scene.bbox.ravel(), "sample_in_bbox")
    ctx = k10.context
    n = H * W
    ridx = ctx.dev(np.arange(n, dtype=np.int32))
    S = torch.zeros((n, D), device="cuda")
    pts = torch.zeros((n, D, 4), device="cuda")
    depth = torch.zeros((n,), device="cuda")
    per_image = []
    for ref in range(3):
        views = scene.view_indices_with_neighbors(ref, nb)
        images = [scene.get_image(v) for v in views]
        feats = bank.stacked(views)
        P = ctx.dev(np.array([im.camera.P for im in images], np.float32))
        Pi = ctx.dev(images[0].camera.P_pinv.astype(np.float32))
        cc = ctx.dev(images[0].camera.center.ravel().astype(np.float32))
        k10(ridx, feats, P, Pi, cc, S, pts, depth)
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            k10(ridx, feats, P, Pi, cc, S, pts, depth)
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b))
        per_image.append(best)
        del feats
    gp = GenerationParameters(depth_planes=D, neighbors=nb, grid_shape=np.array((256, 256, 128), np.int32),
                              max_number_of_marched_voxels=650, padding=pad, gamma_mrf=0.05)
    drv = get_forward_pass_factory("multi_view_cnn")(bank, gp, "sample_in_bbox", (H, W), 130000)
    for _ in drv.forward_pass(scene, (0, 1, 1)):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in drv.forward_pass(scene, (0, 3, 1)):
        pass
    torch.cuda.synchronize()
    drv_ms = (time.perf_counter() - t0) / 3 * 1e3
    ms = float(np.mean(per_image))
    out["k10_getting_started"] = {
        "workload": "multi_view_cnn factory (kernel K10: sampling + plane sweep + softmax + arg-max + "
                    "distance), 1280x720 = 921,600 rays per reference image, 5 views, D = 32, F = 32, "
                    "mock Restrepo cameras, random features",
        "kernel_ms_per_image": round(ms, 3), "kernel_ms_per_image_each": [round(x, 3) for x in per_image],
        "rays_per_s": round(n / ms * 1e3, 1),
        "plane_samples_per_s": round(n * D / ms * 1e3, 1),
        "driver_ms_per_image": round(drv_ms, 3),
        "driver_what": "MultiViewCNNForwardPass.forward_pass per reference image: 8 launches of "
                       "130,000 rays + the map's copy to the host, features resident",
        "published": {"ms_per_image": [64.7, 67.2], "rays_per_s": 14.0e6,
                      "hardware": "1x TITAN X (Pascal), 2018",
                      "source": "docs/getting-started.md:119-157 ('Per-pixel depth estimation')",
                      "note": "other hardware, real features; the only timing the reference publishes"},
        "vs_published": round(65.9 / ms, 1)}
    del drv, S, pts, depth
    torch.cuda.empty_cache()
    # ---- the full path at the CLI defaults ---------------------------------------------------
    fpd = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
    rng_images = (0, 5, 1)

    def step():
        for _ in fpd.forward_pass(scene, rng_images):
            pass
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    c = fpd._ctx
    c.prof_begin(capacity=4096)
    step()
    torch.cuda.synchronize()
    fam = {}
    for name, _, kms in c.prof_end():
        fam[name] = fam.get(name, 0.0) + kms
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    counts = [fpd.voxel_count[r] for r in fpd.voxel_count]
    mean_vox = float(sum(float(x.sum().item()) for x in counts) / max(1, sum(int(x.numel()) for x in counts)))
    out["cli_defaults"] = {
        "workload": "RayNetForwardPass at the reference's CLI defaults: D = 32, grid 256x256x128, "
                    "M = 650, 4 neighbours, 5 reference images of 1280x720 rays (mock Restrepo "
                    "cameras), 3 BP iterations + depth sweep",
        "ms_per_step": round(ms, 3), "rays_per_step": 5 * n, "rays_per_s": round(5 * n / ms * 1e3, 1),
        "mean_voxels_per_ray": round(mean_vox, 1),
        "kernel_ms_per_step": {k: round(v, 3) for k, v in sorted(fam.items())}}
    del fpd
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="config2", choices=sorted(CONFIGS))
    ap.add_argument("--schedule", default="resident", choices=["resident", "reference"])
    ap.add_argument("--rays-batch", type=int, default=0, help="0 = one launch per image shard")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--events", default="dominant", choices=["dominant", "all"],
                    help="launches bracketed by HIP events inside the timed region: those of the "
                         "dominant kernel family (the roofline block), or all of them")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-config4", dest="also_config4", action="store_false",
                    help="skip the side measurement of BASELINE.json configs[3] (other_configs)")
    ap.add_argument("--no-reference-shapes", dest="reference_shapes", action="store_false",
                    help="skip the side measurements on the reference's own workloads (other_configs: "
                         "K10 at the getting-started shape, the full path at its CLI defaults)")
    ap.add_argument("--pmc", default="live", choices=["live", "table", "off"],
                    help="roofline.traffic / valu_insts of the dominant kernel: measured now by "
                         "rocprofv3 --pmc passes of this script spawned from this run (live; N = 1 "
                         "only, falls back to the table), read from profiles/pmc_traffic.json "
                         "(table), or left out (off)")
    args = ap.parse_args()
    t_wall = {"start": time.perf_counter()}

    import torch
    import torch.distributed as dist
    from raynet_amd import _lib
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.forward_pass import get_forward_pass_factory
    from raynet_amd.synthetic import make_synthetic_scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # one rank per GPU; RAYNET_DIST_BACKEND=gloo lets several ranks share one GPU for
    # functional tests of the sharded path on a single-GPU box (RCCL refuses that)
    backend = os.environ.get("RAYNET_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    _lib.load()     # fails loudly if the HIP library was not built

    cfg = CONFIGS[args.config]
    H, W, V = cfg["H"], cfg["W"], cfg["views"]
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=cfg["F"], padding=cfg["padding"],
                                       focal=1.5 * H, seed=1234)
    gp = GenerationParameters(depth_planes=cfg["D"], neighbors=min(4, V - 1) if V <= 5 else V - 1,
                              grid_shape=np.array(cfg["grid"], np.int32),
                              max_number_of_marched_voxels=cfg["M"], padding=cfg["padding"],
                              gamma_mrf=0.05)
    cfg["views_per_ray"] = gp.neighbors + 1
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W),
                                            args.rays_batch, schedule=args.schedule)
    images_range = (0, V, 1)
    t_wall["scene"] = time.perf_counter()

    def step():
        out = None
        for out in fp.forward_pass(scene, images_range):
            pass
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- the FIRST pass (untimed region; reported next to the steady state) ----------------
    # The reference's caller makes ONE pass per scene (scripts/forward_pass.py:120-142): for it the
    # cost of a scene is the first pass -- plan construction (ray lists, voxel counts for the shard
    # cuts, buffers, tables) plus a pass whose scatter still probes its tile shape -- not a warm
    # replay.  `cold_process_first_pass_ms`: the very first pass of this process (HIP module
    # load, context, 7 GB of first-touch allocations on top).  `first_pass_ms`: the same driver
    # object, in the now warm process, handed a NEW scene object (new cameras: every per-scene
    # structure is rebuilt; the buffers come back from the allocator's cache) -- what a caller
    # looping over scenes pays per scene.  Feature maps are resident in both, as in the timed steps.
    fence()
    t0 = time.perf_counter()
    step()
    fence()
    cold_process_first_pass_ms = (time.perf_counter() - t0) * 1e3
    step()          # (the plan's second pass builds the scatter's work list: the process is warm now)
    from raynet_amd.common.scene import Scene
    from raynet_amd.synthetic import _FeatureOnlyImage, ring_cameras
    first_passes = []
    for _ in range(3):      # three scenes in a row, as a caller looping over scenes would: the median
        scene = Scene([_FeatureOnlyImage(H, W, c) for c in ring_cameras(V, H, W, focal=1.5 * H)],
                      scene.bbox)
        fence()
        t0 = time.perf_counter()
        step()
        fence()
        first_passes.append((time.perf_counter() - t0) * 1e3)
    first_pass_ms = sorted(first_passes)[1]
    # (the new plan's second and third pass: work list, the scatter's tile shape settles)
    for _ in range(max(2, args.warmup - 3)):
        step()
    ctx = fp._ctx
    # N > 1: every exchange of the breakdown steps (the all-reduce per BP iteration, the depth
    # rows' way to their owners) is bracketed by a pair of events on its stream (fp.trace)
    if world > 1:
        fp.trace = []
    # Per-kernel breakdown: untimed steps with every launch bracketed by an event pair.  The
    # timed region then brackets the launches of the dominant family only (--events all: every
    # launch, as the breakdown steps do): an event pair costs the stream a few microseconds,
    # ~20 of them per step are 1 % of a one-GPU step and 5 % of an eight-GPU one.
    fence()
    breakdown_steps = max(1, min(3, args.steps))
    ctx.prof_begin(capacity=64 * V * breakdown_steps + 64)
    for _ in range(breakdown_steps):
        step()
    fence()
    launches_all = ctx.prof_end()
    exchange_ms = {}
    if fp.trace is not None:
        open_ev = {}
        for name, begin, ev in fp.trace:
            if begin:
                open_ev[name] = ev
            else:
                exchange_ms[name] = exchange_ms.get(name, 0.0) + open_ev.pop(name).elapsed_time(ev)
        exchange_ms = {k: v / breakdown_steps for k, v in exchange_ms.items()}
        fp.trace = None
    # the step is recorded into ONE HIP graph once the scatter's adaptive tile shape has settled
    # (a dozen scatter launches after the plan was built): the remaining untimed passes until
    # then -- the same number on every rank, the criterion counts launches
    extra_warmup = 0
    capturable = fp.options.capture == "on" or (world > 1 and backend == "nccl" and
                                                  fp.options.capture == "auto")
    # (... the same number on EVERY rank: a rank whose scatter stepped to another tile shape settles
    # -- and captures -- a pass or two later than the others, and a rank that warms up longer than its
    # peers would leave them waiting in the collectives of a step they never run.  The ranks agree
    # after every pass: untimed, one 4-byte all-reduce.)
    while capturable and extra_warmup < 8:
        done = bool(fp.captured)
        if world > 1:
            flag = torch.tensor([1 if done else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(int(flag.item()))
        if done:
            break
        step()
        extra_warmup += 1
    # Which family dominates: by its SHARE of the timeline, not by the sum of its launches'
    # durations -- with the second stream on (config 4: the scatter of one half of the rows next
    # to k_bp of the other) concurrent launches would each be charged the whole overlap.  An
    # instant with k launches in flight gives each of them 1 / k of it.
    by_family = timeline_shares(launches_all, list(getattr(ctx, "prof_starts", [])))
    dominant = max((k for k in by_family if k != "acc"), key=lambda k: by_family[k], default=None)
    only = [dominant] if (args.events == "dominant" and dominant) else None
    # A captured step cannot be bracketed from outside (and the HIP runtime torch ships does not
    # honour external event-record nodes inside a graph: tried, hipEventElapsedTime refuses
    # them).  One GPU: the step is not captured (PathOptions.capture "auto": nothing to gain, the
    # host runs ahead of a 6.7 ms step) and the timed region brackets the dominant family's
    # launches with plain event pairs, as before.  N > 1: the timed region is graph replays;
    # the per-launch durations of every family -- the roofline block's too -- are those of the
    # untimed, eager breakdown steps just before it (roofline.duration_source says which).
    graph_mode = bool(fp.captured)
    if not graph_mode:
        ctx.prof_begin(capacity=64 * V * max(args.steps, 1) + 64, only=only)
    # the interpreter's cyclic garbage collector is a property of the host process, not of the
    # path: a generation-2 collection of a process that has imported torch pauses it for ~40 ms,
    # once every ~20 passes (tools/step_jitter.py) -- five steps' worth landing in whichever
    # timed region happens to contain it.  Collected before, off inside, back on after.
    import gc
    gc.collect()
    gc.disable()
    # The W untimed warm-up steps go HERE, back to back with the timed ones: the collection above (and
    # the host work before it) leaves the GPU idle for tens of milliseconds, and the first passes
    # after an idle spell run 5 - 15 % slower (step_ms.each of a run without them: 7.6, 7.0, 6.7, 6.6
    # ... 6.4 ms -- the clocks come back over ~10 passes).  The passes further up (first-pass
    # figures, the plan's settling, the breakdown) are untimed as well; these are the contract's W.
    for _ in range(max(0, args.warmup)):
        step()
    fence()         # (barrier + synchronize on both sides of the K timed steps)
    t0 = time.perf_counter()
    step_ends = []
    for _ in range(args.steps):
        step()
        step_ends.append(time.perf_counter())     # (a pass returns when its last map is on the host)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    step_ms = [round((b - a) * 1e3, 3) for a, b in zip([t0] + step_ends[:-1], step_ends)]
    launches = [l for l in launches_all if only is None or l[0] in only] if graph_mode \
        else ctx.prof_end()
    ranks_report = None
    if world > 1:
        # what every rank saw: its own wall time for the K steps, its kernel families' sums and
        # its exchanges (breakdown steps, eager), its share of the rays and of the voxel visits
        mine = dict(ms_per_step=elapsed / args.steps * 1e3,
                    exchange_ms_per_step=exchange_ms.get("exchange", 0.0),
                    gather_ms_per_step=exchange_ms.get("gather", 0.0),
                    kernel_ms_per_step=sum(ms for _, _, ms in launches_all) / breakdown_steps,
                    rows=int(sum(len(fp.ray_index[r]) for r in fp.ray_index)))
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        keys = ("ms_per_step", "exchange_ms_per_step", "gather_ms_per_step", "kernel_ms_per_step")
        ranks_report = {k: [round(g[k], 4) for g in gathered] for k in keys}
        ranks_report["rows"] = [g["rows"] for g in gathered]
        ranks_report["ms_per_step_min_max"] = [round(min(ranks_report["ms_per_step"]), 4),
                                               round(max(ranks_report["ms_per_step"]), 4)]
        if fp.shard_balance is not None:
            bal = np.array(fp.shard_balance, dtype=np.float64).sum(0)
            ranks_report["voxel_share_over_mean"] = np.round(bal / bal.mean(), 4).tolist()
        ranks_report["what"] = (
            "per rank, in rank order.  ms_per_step: the rank's own clock over the timed steps (the "
            "line's ms_per_step is the maximum).  exchange / gather: event-bracketed collectives of "
            "the untimed breakdown steps, eager schedule -- all-reduce of the partial accumulators "
            "(3 per step) / the depth rows' way to the maps' owners; a rank that arrives early "
            "waits inside them, so they hold the imbalance too.  kernel: sum of the rank's "
            "launches.  rows, voxel_share_over_mean: the shard.")

    rays_per_step = V * H * W
    value = rays_per_step * args.steps / elapsed
    t_wall["timed"] = time.perf_counter()

    # ---- per-kernel accounting (this rank's launches) ------------------------------
    counts = {r: fp.voxel_count[r] for r in fp.voxel_count}
    vox_by_n = {}
    for r, c in counts.items():
        vox_by_n.setdefault(int(c.numel()), []).append(float(c.sum().item()))
    mean_vox = float(sum(float(c.sum().item()) for c in counts.values()) /
                     max(1, sum(int(c.numel()) for c in counts.values())))
    cfg_acc = dict(cfg, views=gp.neighbors + 1)
    # rays that miss the bounding box take no sweep (no voxels: result-neutral) but count in
    # rays_per_step, as SURVEY.md 8(d)'s V R / T has it
    rays_missing = "%d of %d" % (int(sum(int((c == 0).sum().item()) for c in counts.values())),
                                 int(sum(int(c.numel()) for c in counts.values())))

    # the plane sweep wrote BP iteration 0's messages itself when a step shows one k_bp launch
    # fewer than BP iterations (rn_scene_run folds it when its LDS rows fit)
    n_bp = sum(1 for name, _, _ in launches_all if name == "bp") / breakdown_steps
    folded = n_bp < fp.bp_iterations - 0.5

    def account(recorded):
        fam = {}
        for name, n_rays, ms in recorded:
            f = fam.setdefault(name, dict(ms=0.0, launches=0, bytes=0.0, strict=0.0, resident=0.0))
            f["ms"] += ms
            f["launches"] += 1
            if n_rays:
                # launches are per image shard: voxels of a launch = mean over images with n rays
                vs = vox_by_n.get(n_rays)
                vox = float(np.mean(vs)) if vs else mean_vox * n_rays
                per_image = max(1, min(int(c.numel()) for c in counts.values())) if counts else n_rays
                images = max(1, n_rays // per_image)
                f["bytes"] += algorithmic_bytes(name, n_rays, vox, cfg_acc, images=images, folded=folded)
                f["strict"] += algorithmic_bytes(name, n_rays, vox, cfg_acc, images=images,
                                                 folded=folded, strict=True)
                f["resident"] += cache_resident_bytes(name, vox)
        return fam

    fam = account(launches_all)             # the breakdown steps: every family
    timed = account(launches)               # the timed region: the dominant family (or all)
    roofline = live = None
    if dominant and dominant in timed:
        d = timed[dominant]
        achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        # HBM-side bytes per launch come from PMC counters, which need their own rocprofv3
        # passes (tools/pmc_passes.sh): the figure is read from the summary committed for
        # this configuration, never measured inside this run -- traffic_source says so
        traffic = traffic_source = valu_insts = valu_active_quads = l2_req = l2_miss = None
        tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
        live = None
        if world == 1 and args.pmc == "live":
            # frees the GPU's memory first: the profiled child builds its own 7 - 25 GB plan
            fp._plan = None
            torch.cuda.empty_cache()
            live = live_pmc(args.config, dominant)
        if live is not None:
            # 2 x FETCH_SIZE + WRITE_SIZE, KiB: the gfx950 correction of MI355X_MICROARCH.md (HBM)
            traffic = int((2 * live["FETCH_SIZE"] + live["WRITE_SIZE"]) * 1024)
            valu_insts = live["SQ_INSTS_VALU"]
            valu_active_quads = live.get("SQ_ACTIVE_INST_VALU")
            l2_req, l2_miss = live.get("TCC_REQ_sum"), live.get("TCC_MISS_sum")
            traffic_source = ("measured in this run: four rocprofv3 --pmc passes (FETCH_SIZE; "
                              "WRITE_SIZE; SQ_INSTS_VALU + SQ_ACTIVE_INST_VALU; TCC_REQ + TCC_MISS; "
                              "kernel-trace only) of `bench.py --steps 1` "
                              "spawned by this process, mean of %d launches, %.0f s" % (
                                  live["launches"], live["seconds"]))
        elif world == 1 and args.pmc != "off" and os.path.exists(tpath):   # N=1 launch sizes
            try:
                tj = json.load(open(tpath)).get(args.config)
                if tj is not None and dominant in tj["kernels"]:
                    traffic = tj["kernels"][dominant]["traffic_bytes"]
                    valu_insts = tj["kernels"][dominant].get("valu_insts")
                    traffic_source = "profiles/pmc_traffic.json (%s): separate rocprofv3 --pmc " \
                                     "passes of this command, not this run" % tj["profile"]
            except Exception:
                traffic = valu_insts = None
        avg_ms = d["ms"] / d["launches"]
        hbm_ms = d["bytes"] / d["launches"] / (HBM_PEAK_GBS * 1e9) * 1e3
        # what binds the kernel: the time its algorithmic bytes need at the HBM peak, or the time
        # its VALU instructions need to ISSUE (4 cycles each on one of 1024 SIMDs) -- counters
        # from the same separate PMC passes as `traffic`
        cyc, cyc_source = valu_cycles_per_inst(dominant, args.config)
        if valu_insts and valu_active_quads:
            # measured, not modelled: SQ_ACTIVE_INST_VALU counts the quad-cycles (4 shader cycles,
            # MI355X_MICROARCH.md) the SIMDs spent on the kernel's VALU instructions as executed
            cyc = 4.0 * valu_active_quads / valu_insts
            cyc_source = "4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU of this run's counter pass"
        valu_issue_ms = valu_insts * cyc / (N_SIMDS * VALU_CLOCK_HZ) * 1e3 if valu_insts else None
        strict = d["strict"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        # what binds: the larger of (a) the time the kernel's memory-side bytes need -- the
        # algorithmic ones at the HBM peak, or what the counters saw leave the L2 (FETCH / WRITE:
        # HBM and Infinity Cache alike) at the measured copy rate -- and (b) its VALU-issue time
        traffic_ms = traffic / (MEASURED_COPY_GBS * 1e9) * 1e3 if traffic else None
        mem_ms = max(hbm_ms, traffic_ms or 0.0)
        roofline = dict(bound="valu" if (valu_issue_ms or 0.0) > mem_ms else "hbm", kernel=dominant,
                        achieved=round(achieved, 1),
                        peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                        traffic=traffic, traffic_source=traffic_source,
                        avg_launch_ms=round(avg_ms, 4), launches=d["launches"],
                        algorithmic_bytes_per_launch=int(d["bytes"] / d["launches"]),
                        hbm_time_ms=round(hbm_ms, 4),
                        l2_side_traffic_time_ms=round(traffic_ms, 4) if traffic_ms else None,
                        valu_cycles_per_inst=cyc, valu_cycles_source=cyc_source,
                        valu_insts_per_launch=valu_insts,
                        valu_issue_ms=round(valu_issue_ms, 4) if valu_issue_ms else None,
                        valu_frac=round(valu_issue_ms / avg_ms, 4) if valu_issue_ms else None,
                        # the gathers' side (profiles/r05_exp_sweep_gather_orders.txt: the sweep's
                        # loads ALONE take ~85 % of the kernel's time): 128-byte line requests the
                        # L1s sent to the L2, and how many of them missed there
                        l2_requests_per_launch=int(l2_req) if l2_req else None,
                        l2_misses_per_launch=int(l2_miss) if l2_miss else None,
                        l2_to_l1_TBps=round(l2_req * 128 / (avg_ms * 1e-3) / 1e12, 2) if l2_req else None,
                        strict_algorithmic=dict(
                            what="SURVEY.md 8(d): the N feature maps once per reference image, no "
                                 "lists / columns / messages",
                            bytes_per_launch=int(d["strict"] / d["launches"]),
                            achieved=round(strict, 1), frac=round(strict / HBM_PEAK_GBS, 4)),
                        writes_first_messages=bool(folded) if dominant == "sweep_map" else None,
                        duration_source="HIP event pairs around the kernel's launches in the timed "
                                        "region" if not graph_mode else
                                        "HIP event pairs around the kernel's launches in the %d untimed "
                                        "eager steps before the timed region (which replays a captured "
                                        "graph)" % breakdown_steps)
    kernels = {k: dict(total_ms_per_step=round(v["ms"] / breakdown_steps, 3),
                       timeline_share_ms_per_step=round(by_family.get(k, 0.0) / breakdown_steps, 3),
                       launches_per_step=v["launches"] / breakdown_steps,
                       algorithmic_GBps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
                       if v["ms"] > 0 and v["bytes"] else None,
                       **({"cache_resident_GBps": round(v["resident"] / (v["ms"] * 1e-3) / 1e9, 1),
                           "cache_resident_what": "accumulator read-modify-write of SURVEY.md 8(d)'s "
                                                  "count (8 B per visit): LDS box + L2, not HBM"}
                          if v["resident"] and v["ms"] > 0 else {}))
               for k, v in sorted(fam.items())}

    t_wall["counters"] = time.perf_counter()
    # ---- BASELINE.json configs[3] on the side (N = 1, default run only) ------------------------
    # The headline is config 2; config 4 (9 views, 128 planes, 256^3, M = 768) gets ONE figure in
    # the same line so that a driver-run record of it exists: a fresh driver, 5 untimed passes
    # (plan, work list, the scatter's tile shape), 5 timed ones.  Outside the timed region above.
    other_configs = None
    if world == 1 and args.config == "config2" and args.also_config4 and args.schedule == "resident":
        try:
            c4 = CONFIGS["config4"]
            scene4, bank4 = make_synthetic_scene(H=c4["H"], W=c4["W"], n_views=c4["views"], F=c4["F"],
                                                 padding=c4["padding"], focal=1.5 * c4["H"], seed=1234)
            gp4 = GenerationParameters(depth_planes=c4["D"], neighbors=c4["views"] - 1,
                                       grid_shape=np.array(c4["grid"], np.int32),
                                       max_number_of_marched_voxels=c4["M"], padding=c4["padding"],
                                       gamma_mrf=0.05)
            # config 2's 7 GB back to the allocator first (the plan is also held by the driver's
            # record of its last call)
            fp._plan = None
            fp._quick = None
            torch.cuda.empty_cache()
            fp4 = get_forward_pass_factory("raynet")(bank4, gp4, "sample_in_bbox",
                                                     (c4["H"], c4["W"]), 0)

            def step4():
                for _ in fp4.forward_pass(scene4, (0, c4["views"], 1)):
                    pass
            for _ in range(5):
                step4()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(5):
                step4()
            torch.cuda.synchronize()
            ms4 = (time.perf_counter() - t4) / 5 * 1e3
            rays4 = c4["views"] * c4["H"] * c4["W"]
            other_configs = {"config4": {"workload": c4["workload"], "ms_per_step": round(ms4, 3),
                                         "rays_per_s": round(rays4 / ms4 * 1e3, 1), "steps": 5,
                                         "rays_per_step": rays4}}
            del fp4, scene4, bank4
            torch.cuda.empty_cache()
        except Exception as e:              # (never at the cost of the headline)
            other_configs = {"config4": {"error": repr(e)[:200]}}

    # ---- the reference's OWN workloads on the side (N = 1, default run only) -------------------
    if world == 1 and args.config == "config2" and args.reference_shapes and args.schedule == "resident":
        other_configs = dict(other_configs or {})
        try:
            fp._plan = None
            fp._quick = None
            torch.cuda.empty_cache()
            other_configs.update(reference_shapes(torch))
        except Exception as e:              # (never at the cost of the headline)
            other_configs["reference_shapes"] = {"error": repr(e)[:300]}

    t_wall["config4"] = time.perf_counter()
    # ---- CPU baseline, two legs on bounded ray samples drawn from ALL reference images -------
    #  port              the C oracle's fused K1 / K2 (whole path, OpenMP over rays, all cores)
    #  numpy_restatement oracle/cpu_reference.py = the reference's own CPU implementation
    #                    (mrf/mrf_np.py: per-ray Python loop, one process) of the stages it
    #                    has one for -- the 3 BP sweeps + the depth sweep; the plane sweep has
    #                    no NumPy twin in the reference, its inputs come from the port
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_reference, oracle
        threads = oracle.Oracle.max_threads()
        o = oracle.Oracle(M=cfg["M"], D=cfg["D"], N=gp.neighbors + 1, F=cfg["F"], H=H, W=W,
                          padding=cfg["padding"], bbox=scene.bbox.ravel(), grid_shape=cfg["grid"],
                          threads=threads)
        vg = oracle.voxel_grid_centers(scene.bbox.ravel(), cfg["grid"])
        cams = []
        for r in range(V):
            views = scene.view_indices_with_neighbors(r, gp.neighbors)
            cams.append((bank.stacked(views).cpu().numpy(),
                         np.array([scene.get_image(v).camera.P for v in views], np.float32),
                         scene.get_image(r).camera.P_pinv.astype(np.float32),
                         scene.get_image(r).camera.center.ravel().astype(np.float32)))
        rng = np.random.default_rng(0)

        def cpu_run(n, keep=False):
            """n rays per reference image, the schedule of forward_pass.py:593-736."""
            ridx = [np.sort(rng.choice(H * W, n, replace=False)).astype(np.int32) for _ in cams]
            acc = o.prior(0.05)
            msgs = [np.zeros((n, cfg["M"]), np.float32) for _ in cams]
            kept = None
            t = time.perf_counter()
            for it in range(3):
                out = o.prior(0.05)
                for k, (f, P, Pi, cc) in enumerate(cams):
                    kept_k = o.fused_bp(ridx[k], f, P, Pi, cc, vg, acc, msgs[k], out)
                    if keep and it == 0:
                        kept = (kept or []) + [kept_k]
                acc = out
            for k, (f, P, Pi, cc) in enumerate(cams):
                o.fused_depth(ridx[k], f, P, Pi, cc, vg, acc, msgs[k])
            return time.perf_counter() - t, kept

        # grow the sample until a run takes about the time budget (small samples are all
        # OpenMP start-up; the rate is only meaningful once every core has rays)
        target = 0.6 * args.cpu_seconds
        n = min(2000, H * W)
        while True:
            tc, _ = cpu_run(n)
            if tc >= 0.5 * target or n >= H * W:
                break
            n = int(min(H * W, n * min(8.0, max(2.0, target / max(tc, 1e-6)))))
        # why the many-thread rate is what it is: the same leg on ONE thread (a smaller sample),
        # and what the box really grants this process (affinity mask, cgroup CPU quota)
        o1 = oracle.Oracle(M=cfg["M"], D=cfg["D"], N=gp.neighbors + 1, F=cfg["F"], H=H, W=W,
                           padding=cfg["padding"], bbox=scene.bbox.ravel(), grid_shape=cfg["grid"],
                           threads=1)
        o_all, o = o, o1
        n1 = min(400, H * W)
        t1, _ = cpu_run(n1)
        o = o_all
        one_thread = V * n1 / t1
        try:
            quota = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = "cgroup cpu.max %s" % ("unlimited" if quota[0] == "max" else
                                            "%.1f CPUs" % (float(quota[0]) / float(quota[1])))
        except Exception:
            quota = "cgroup quota unknown"
        try:
            affinity = len(os.sched_getaffinity(0))
        except Exception:
            affinity = None
        # the same leg with per-thread partial accumulators (merged once per K1 call) on as many
        # threads as the box GRANTS this process: no cache line of the accumulator is shared
        # between cores while the rays run -- the honest "all host cores" figure
        granted = affinity or threads
        try:
            q = open("/sys/fs/cgroup/cpu.max").read().split()
            if q[0] != "max":
                granted = max(1, min(granted, int(round(float(q[0]) / float(q[1])))))
        except Exception:
            pass
        granted = max(1, min(granted, threads))
        og = oracle.Oracle(M=cfg["M"], D=cfg["D"], N=gp.neighbors + 1, F=cfg["F"], H=H, W=W,
                           padding=cfg["padding"], bbox=scene.bbox.ravel(), grid_shape=cfg["grid"],
                           threads=granted)
        oracle.Oracle.set_private_accumulators(True)
        try:
            o = og
            tp, _ = cpu_run(n)
        finally:
            oracle.Oracle.set_private_accumulators(False)
            o = o_all
        shared_rate, private_rate = V * n / tc, V * n / tp
        best_rate, best_cores = (private_rate, granted) if private_rate >= shared_rate else (shared_rate, threads)
        port = dict(value=round(best_rate, 1), unit="rays/s", cores=best_cores, kind="port",
                    one_thread_rays_per_s=round(one_thread, 1),
                    speedup_over_one_thread=round(best_rate / one_thread, 2),
                    shared_accumulator=dict(rays_per_s=round(shared_rate, 1), threads=threads,
                                            what="`omp atomic` adds into ONE accumulator (the reference "
                                                 "kernel's own scheme, mrf_bp.cu:170-176)"),
                    private_accumulators=dict(rays_per_s=round(private_rate, 1), threads=granted,
                                              what="a zero-started accumulator per thread, merged at the "
                                                   "end of every K1 call; threads = the CPUs the box grants"),
                    sample="%d rays of each of the %d reference images (of %d per step), 3 BP "
                           "sweeps + depth sweep with the oracle's fused K1/K2 "
                           "(oracle/raynet_oracle.c, OpenMP over rays), twice: %d threads adding into one "
                           "shared accumulator with `omp atomic` (%.1f s, %.0f rays/s) and %d threads with "
                           "private accumulators (%.1f s, %.0f rays/s); value = the faster.  One thread "
                           "does %.0f rays/s (affinity mask %s CPUs, %s): the threads a process may START "
                           "are not the cores it GETS"
                           % (n, V, rays_per_step, threads, tc, shared_rate, granted, tp, private_rate,
                              one_thread, affinity, quota))

        def numpy_run(n_np):
            _, kept = cpu_run(n_np, keep=True)
            rvi = np.concatenate([k[0] for k in kept])
            rvc = np.concatenate([k[1] for k in kept])
            Sv = np.concatenate([k[2] for k in kept])
            t = time.perf_counter()
            m_np = np.zeros_like(Sv)
            acc_np, m_np = cpu_reference.belief_propagation(Sv, rvi, rvc, m_np, cfg["grid"],
                                                            gamma=0.05, bp_iterations=3)
            cpu_reference.compute_depth_distribution(Sv, rvi, rvc, m_np, acc_np)
            return time.perf_counter() - t, len(rvc)

        n_np = min(1000, H * W)
        tn, rays_np = numpy_run(n_np)
        want = 0.4 * args.cpu_seconds
        if tn < 0.5 * want and n_np < H * W:
            n_np = int(min(H * W, n_np * want / max(tn, 1e-6)))
            tn, rays_np = numpy_run(n_np)
        numpy_leg = dict(value=round(rays_np / tn, 1), unit="rays/s", cores=1,
                         kind="numpy_restatement",
                         sample="%d rays of each of the %d reference images; the reference's "
                                "NumPy path (mrf/mrf_np.py semantics, oracle/cpu_reference.py: "
                                "per-ray Python loop, float64 cumulative sums) for the stages it "
                                "covers -- 3 BP sweeps + depth sweep on the mapped columns; the "
                                "plane sweep has no NumPy twin in the reference -- %.1f s"
                                % (n_np, V, tn))
        cpu = dict(port, legs={"port": port, "numpy_restatement": numpy_leg})

    feat_per_ray = 4.0 * (gp.neighbors + 1) * cfg["F"] * (H + cfg["padding"] + 1) * \
        (W + cfg["padding"] + 1) / (H * W)
    path_bytes = rays_per_step * (3 * (feat_per_ray + 20 * mean_vox) + feat_per_ray + 8 * mean_vox + 4)
    path_gbps = path_bytes / elapsed * args.steps / 1e9
    path_roofline = dict(bytes_per_ray=round(path_bytes / rays_per_step, 1),
                         achieved=round(path_gbps, 1), peak=HBM_PEAK_GBS * world, unit="GB/s",
                         frac=round(path_gbps / (HBM_PEAK_GBS * world), 4))
    if live is not None and live.get("path_SQ_ACTIVE_INST_VALU"):
        # The bound every kernel of this path is nearest to is VALU issue, not HBM: the cycles the
        # chip's SIMDs were busy executing ALL kernels' VALU instructions of one pass (the counter
        # pass's SQ_ACTIVE_INST_VALU, quad-cycles, summed over every kernel) against the SIMD
        # cycles a step has
        busy_ms = 4.0 * live["path_SQ_ACTIVE_INST_VALU"] / (N_SIMDS * VALU_CLOCK_HZ) * 1e3
        path_roofline["valu"] = dict(
            insts_per_step=live.get("path_SQ_INSTS_VALU"), busy_ms_per_step=round(busy_ms, 3),
            frac_of_step=round(busy_ms / (elapsed / args.steps * 1e3), 4),
            what="4 x SQ_ACTIVE_INST_VALU of every kernel of a pass / (%d SIMDs x %.1f GHz): the share "
                 "of a step the VALUs are busy" % (N_SIMDS, VALU_CLOCK_HZ / 1e9))
    if rank == 0:
        result = {
            "metric": "rays/sec (whole node) at %d views x %d depths x %d^3 voxels" % (
                gp.neighbors + 1, cfg["D"], cfg["grid"][0]),
            "value": round(value, 1),
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "name": args.config,
                       "rays_per_step": rays_per_step, "bp_iterations": 3,
                       "schedule": args.schedule, "options": fp.options.as_dict(),
                       "parallelism": "rays sharded x%d, 1 all-reduce/BP iteration" % world
                       if world > 1 else "single GPU",
                       "mean_voxels_per_ray": round(mean_vox, 2),
                       "rays_missing_the_box": rays_missing,
                       "reused_between_steps": [
                           "feature maps (resident in HBM: the MV-CNN is outside the path)",
                           "the scene's plan: camera / feature-pointer tables, the patch-ordered ray "
                           "list, shard cuts (N > 1: from one voxel-count launch), the slab-box table "
                           "and scatter work list, the HBM buffers (voxel lists, columns, messages: "
                           "allocated once, every entry a pass reads is rewritten by that pass), the "
                           "pinned host maps",
                           "the scatter's settled tile shape; N > 1: the step's captured HIP graph",
                           "NOTHING of a pass's results: traversal, plane sweep, mapping, 3 BP "
                           "iterations, depth sweep and the maps' copies run in every step"]},
            "first_pass_ms": round(first_pass_ms, 3),
            "first_pass_ms_all": [round(t, 3) for t in first_passes],
            "cold_process_first_pass_ms": round(cold_process_first_pass_ms, 3),
            "first_pass": "first_pass_ms: this driver handed a NEW scene object (every per-scene "
                          "structure rebuilt, then one pass; three scenes in a row, the median) -- what "
                          "the reference's one-pass-per-scene caller pays per scene "
                          "(scripts/forward_pass.py:120-142); cold_process: "
                          "the process's very first pass (HIP module load, context, first-touch "
                          "allocation on top).  Both outside the timed region.",
            "ray_sweeps_per_s": round(4 * value, 1),
            # SURVEY.md 8(d)'s whole-path figure: (3 B_bp + B_de) bytes per ray, B_bp = 4NF HfWf/HW
            # + 20 c, B_de = 4NF HfWf/HW + 8 c + 4 (features once per sweep, per traversed voxel:
            # gather 4 + msg 4 + 4 + atomic RMW 8), over the step's wall time
            "path_roofline": path_roofline,
            "step_ms": dict(each=step_ms[:32], min=min(step_ms), median=sorted(step_ms)[len(step_ms) // 2],
                            what="this rank's wall time of every timed step (ms_per_step is their mean "
                                 "incl. the closing barrier, MAX over ranks)"),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "other_configs": other_configs,
            "wall_s": {"imports_scene": round(t_wall["scene"] - t_wall["start"], 1),
                       "warmup_breakdown_timed": round(t_wall["timed"] - t_wall["scene"], 1),
                       "counter_passes": round(t_wall["counters"] - t_wall["timed"], 1),
                       "config4": round(t_wall["config4"] - t_wall["counters"], 1),
                       "cpu_baseline": round(time.perf_counter() - t_wall["config4"], 1)},
            "kernels": kernels,
            "ranks": ranks_report,
            "step_capture": {"captured": bool(fp.captured), "extra_warmup_steps": extra_warmup,
                             "timed_region": "graph replays (launch durations: the eager breakdown "
                                             "steps before it)"
                             if graph_mode else "eager launches bracketed by event pairs"},
            "kernel_events": {"breakdown_steps_untimed": breakdown_steps,
                              "timed_region": "all launches" if only is None else
                              "launches of %s only" % dominant},
        }
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
