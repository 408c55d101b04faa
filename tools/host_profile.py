"""cProfile of the Python side of a config-2 step (what the host spends between launches)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.forward_pass import get_forward_pass_factory
from raynet_amd.synthetic import make_synthetic_scene
import raynet_amd.forward_pass as F
H, W, V = 480, int(os.environ.get("W", "640")), 5
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H)
gp = GenerationParameters(depth_planes=64, neighbors=4, grid_shape=np.array([128]*3, np.int32),
                          max_number_of_marched_voxels=384, padding=11, gamma_mrf=0.05)
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
for _ in range(3): list(fp.forward_pass(scene, (0, V, 1)))
# host time of a step = time until the generator yields its first map minus GPU wait:
# measure with the GPU made "infinitely fast": cProfile the python side
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): list(fp.forward_pass(scene, (0, V, 1)))
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
