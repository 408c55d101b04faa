#!/usr/bin/env python3
"""Debug: where a k_sweep_map wavefront's cycles go (config 2).  On the GPU box:
     python tools/phase_timers.py
builds tools/libraynet_hip_phase.so with -DRN_PHASE_TIMERS, runs the scene and prints the
mean s_memtime cycles between the marks of every 16th ray."""
import ctypes, os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from raynet_amd import _lib
so = os.path.join(REPO, "tools", "libraynet_hip_phase.so")
subprocess.check_call(["/opt/rocm/bin/hipcc"] + _lib.HIPCC_FLAGS + sys.argv[1:] +
                      ["-DRN_PHASE_TIMERS", "-I", os.path.join(REPO, "include"),
                       os.path.join(_lib.CSRC, "raynet_hip.hip"), "-o", so])
_lib.LIB_PATH = so
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.forward_pass import get_forward_pass_factory
from raynet_amd.synthetic import make_synthetic_scene
H, W, V = 480, 640, 5
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=32, padding=11, focal=1.5 * H, seed=1234)
gp = GenerationParameters(depth_planes=64, neighbors=4, grid_shape=np.array([128] * 3, np.int32),
                          max_number_of_marched_voxels=int(os.environ.get("M", "384")), padding=11,
                          gamma_mrf=0.05)
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
lib = _lib.load()
def step():
    for _ in fp.forward_pass(scene, (0, V, 1)):
        pass
    torch.cuda.synchronize()
step(); step()
out = (ctypes.c_ulonglong * 16)()
lib.rn_debug_phase(None, 1)
step()
lib.rn_debug_phase(out, 0)
n = max(1, out[15])
names = ["(clock)", "ray index + segment", "projection + gathers + pair sums", "softmax",
         "planes -> voxels", "clip + renorm + store"]
tot = sum(out[k] for k in range(6))
print("sampled waves: %d, mean lifetime after launch %.0f cycles" % (n, tot / n))
for k, nm in enumerate(names):
    print("  %-36s %8.0f cycles  %5.1f %%" % (nm, out[k] / n, 100.0 * out[k] / max(1, tot)))
