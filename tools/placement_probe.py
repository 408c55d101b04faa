"""How much of a step's time is WHERE its buffers landed?  K identical driver objects in one
process (each with its own plan: own rows, own accumulators), rounds taken in turns, per-family
kernel sums and the buffers' addresses printed per object.

    K=6 STEPS=20 ROUNDS=5 python tools/placement_probe.py"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd.common.generation_parameters import GenerationParameters  # noqa: E402
from raynet_amd.forward_pass import get_forward_pass_factory               # noqa: E402
from raynet_amd.synthetic import make_synthetic_scene                      # noqa: E402

H, W, V, D_, M_, G_ = 480, 640, 5, 64, 384, 128
K, STEPS, ROUNDS = int(os.environ.get("K", "6")), int(os.environ.get("STEPS", "20")), \
    int(os.environ.get("ROUNDS", "5"))
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=32, padding=11, focal=1.5 * H, seed=1234)
gp = GenerationParameters(depth_planes=D_, neighbors=4, grid_shape=np.array([G_] * 3, np.int32),
                          max_number_of_marched_voxels=M_, padding=11, gamma_mrf=0.05)
fps = []
for i in range(K):
    fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
    for _ in range(3):
        for _ in fp.forward_pass(scene, (0, V, 1)):
            pass
    fps.append(fp)
fam = [dict() for _ in fps]
for rnd in range(ROUNDS):
    for i, fp in enumerate(fps):
        torch.cuda.synchronize()
        fp._ctx.prof_begin(capacity=64 * STEPS)
        for _ in range(STEPS):
            for _ in fp.forward_pass(scene, (0, V, 1)):
                pass
        torch.cuda.synchronize()
        for name, _, ms in fp._ctx.prof_end():
            fam[i][name] = fam[i].get(name, 0.0) + ms / (STEPS * ROUNDS)
for i, fp in enumerate(fps):
    pl = fp._plan
    ptr = {k: pl[k].data_ptr() for k in ("vox", "Sr", "msgs", "acc_a", "acc_b", "rvc")}
    print("obj %d  total %.3f  %s" % (i, sum(fam[i].values()),
                                       " ".join("%s=%.3f" % kv for kv in sorted(fam[i].items()))))
    print("       " + " ".join("%s=0x%x" % kv for kv in ptr.items()))
