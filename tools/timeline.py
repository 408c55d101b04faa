"""Launch timeline of bench steps: start offset, duration and the idle gap in front of
every library kernel (hipEvents of rn_prof_begin/_end), to see what the host side costs."""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.forward_pass import get_forward_pass_factory
from raynet_amd.synthetic import make_synthetic_scene
H, W, V = 480, int(os.environ.get("W", "640")), 5
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H)
gp = GenerationParameters(depth_planes=64, neighbors=4, grid_shape=np.array([128] * 3, np.int32),
                          max_number_of_marched_voxels=384, padding=11, gamma_mrf=0.05)
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
for _ in range(3):
    list(fp.forward_pass(scene, (0, V, 1)))
ctx = fp._ctx
torch.cuda.synchronize()
ctx.prof_begin(capacity=256)
t0 = time.perf_counter()
for _ in range(3):
    list(fp.forward_pass(scene, (0, V, 1)))
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
launches = ctx.prof_end()
end_prev = 0.0
print("wall %.3f ms for 3 steps" % wall)
for (name, n, ms), st in zip(launches, ctx.prof_starts):
    print("%-10s start %8.3f  gap %7.3f  dur %7.3f" % (name, st, st - end_prev, ms))
    end_prev = st + ms
