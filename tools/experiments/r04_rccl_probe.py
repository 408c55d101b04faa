"""Which piece of the one-rank RCCL path works: eager all-to-all epilogue, captured step with
all-reduce only (gather="all"), captured step with the all-to-all."""
import os, sys, faulthandler
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch, torch.distributed as dist
from raynet_amd.forward_pass import get_forward_pass_factory
from raynet_amd.hip_implementations.options import PathOptions
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.synthetic import make_synthetic_scene
which = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
H, W = 48, 64
scene, bank = make_synthetic_scene(H=H, W=W, n_views=5, focal=1.5 * H)
gp = GenerationParameters(depth_planes=32, neighbors=4, grid_shape=np.array((64, 64, 64), np.int32),
                          max_number_of_marched_voxels=192, padding=11, gamma_mrf=0.05)
opt = {"a2a_eager": PathOptions(deterministic=True, capture="off"),
       "all_eager": PathOptions(deterministic=True, capture="off", gather="all"),
       "all_captured": PathOptions(deterministic=True, gather="all"),
       "a2a_captured": PathOptions(deterministic=True)}[which]
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0, options=opt)
for i in range(45):
    d = [m.copy() for m in fp.forward_pass(scene, (0, 5, 1))]
    if fp.captured:
        break
print(which, "passes", i + 1, "captured", fp.captured, "a2a", fp._plan.get("a2a") is not None, flush=True)
d = [m.copy() for m in fp.forward_pass(scene, (0, 5, 1))]
torch.cuda.synchronize()
print(which, "OK", float(np.stack(d).mean()), flush=True)
dist.destroy_process_group()
