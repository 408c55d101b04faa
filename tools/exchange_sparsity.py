#!/usr/bin/env python3
"""How much of the voxel grid one rank of N touches, and how much of what it touches others touch
too: the bytes a rank NEEDS from the others per BP iteration, against the dense all-reduce the
path runs today.  One GPU: every rank of `--world` in turn builds its plan
with the collectives stubbed (tools/shard_proxy.py's stand-in) and its voxel lists are marked in a
grid mask.

    gpurun -- python tools/exchange_sparsity.py --config config2 --world 8
"""
import argparse
import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config2", choices=["config2", "config4"])
    ap.add_argument("--world", type=int, default=8)
    args = ap.parse_args()
    import torch
    import bench
    import raynet_amd.forward_pass as F
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.synthetic import make_synthetic_scene
    cfg = bench.CONFIGS[args.config]
    H, W, V, D, M, Fd, pad = (cfg[k] for k in ("H", "W", "views", "D", "M", "F", "padding"))
    G = cfg["grid"]
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=Fd, padding=pad, focal=1.5 * H, seed=1234)
    gp = GenerationParameters(depth_planes=D, neighbors=min(4, V - 1) if V <= 5 else V - 1,
                              grid_shape=np.array(G, np.int32), max_number_of_marched_voxels=M,
                              padding=pad, gamma_mrf=0.05)

    class FakeDist(object):
        ReduceOp = types.SimpleNamespace(SUM=0, MIN=1, MAX=2)
        capturable = False

        def __init__(self, world):
            self.world = world

        def get_backend(self):
            return "fake"

        def all_reduce(self, t, op=None):
            pass

        def all_gather_into_tensor(self, out, inp, async_op=False):
            out.view(self.world, -1).copy_(inp.view(1, -1).expand(self.world, -1))

    world = args.world
    nvox = G[0] * G[1] * G[2]
    masks = torch.zeros((world, nvox), dtype=torch.bool, device="cuda")
    visits = []
    for rank in range(world):
        fake = FakeDist(world)
        F._dist = lambda f=fake, r=rank, w=world: (f, r, w)
        fp = F.get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
        for _ in fp.forward_pass(scene, (0, V, 1)):
            pass
        plan = fp._plan
        vox, rvc = plan["vox"], plan["rvc"]
        n = rvc.shape[0]
        live = torch.arange(M, device="cuda")[None, :] < torch.where(rvc >= 2, rvc, torch.zeros_like(rvc))[:, None]
        v = vox[:n][live].long()
        lin = ((v >> 20) * G[1] + ((v >> 10) & 1023)) * G[2] + (v & 1023)
        masks[rank, lin] = True
        visits.append(int(live.sum()))
        del fp, plan, vox, rvc, live, v, lin
        torch.cuda.empty_cache()
    touched = masks.sum(1).cpu().numpy()                    # voxels rank r sends messages to
    cover = masks.sum(0)                                     # ranks per voxel
    need = []
    for r in range(world):
        # what rank r needs from the others: for each voxel it touches, one partial per OTHER rank that touches it
        need.append(int((cover[masks[r]] - 1).sum()))
    bytes_per = 4
    dense_allreduce = 2.0 * (world - 1) / world * nvox * bytes_per      # ring / RS + AG, per rank, sent = received
    rep = {"config": args.config, "world": world, "grid_voxels": nvox,
           "voxels_touched_by_anyone": int((cover > 0).sum()),
           "touched_fraction_per_rank": [round(float(t) / nvox, 4) for t in touched],
           "mean_ranks_per_touched_voxel": round(float(cover[cover > 0].float().mean()), 3),
           "visits_per_rank": visits,
           "partials_a_rank_needs_from_others_MB": [round(x * bytes_per / 1e6, 3) for x in need],
           "dense_all_reduce_per_rank_MB": round(dense_allreduce / 1e6, 3),
           "ratio_dense_over_needed": round(dense_allreduce / (max(need) * bytes_per), 2),
           "what": "needed = for every voxel a rank's rays visit, one fp32 partial from every OTHER rank whose rays "
                   "visit it (a pairwise exchange over the touched intersections); dense = what a bandwidth-optimal "
                   "all-reduce of the whole grid moves per rank (2 (N-1)/N x grid)"}
    print(json.dumps(rep, indent=1))
    out = os.path.join(REPO, "gpurun_out", "r05_exchange_sparsity_%s_w%d.json" % (args.config, world))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rep, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
