"""Per-step wall times of repeated passes over one plan: where do outliers come from?
    python tools/step_jitter.py [n_steps]   (GC=0: garbage collector off; RAYNET_PLAN_PATH=0 ...)"""
import gc, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.forward_pass import get_forward_pass_factory
from raynet_amd.synthetic import make_synthetic_scene
H, W, V = 480, 640, 5
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=32, padding=11, focal=1.5 * H, seed=1234)
gp = GenerationParameters(depth_planes=64, neighbors=4, grid_shape=np.array([128] * 3, np.int32),
                          max_number_of_marched_voxels=384, padding=11, gamma_mrf=0.05)
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
if os.environ.get("GC") == "0":
    gc.disable()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ts = []
for i in range(n):
    t = time.perf_counter()
    for _ in fp.forward_pass(scene, (0, V, 1)):
        pass
    ts.append((time.perf_counter() - t) * 1e3)
ts = np.array(ts)
med = float(np.median(ts[3:]))
print("median %.3f ms; outliers (> 1.5 x median): %s" % (
    med, [(i, round(float(t), 2)) for i, t in enumerate(ts) if i >= 3 and t > 1.5 * med]))
