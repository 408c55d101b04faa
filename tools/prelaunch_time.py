"""Host time from entering forward_pass to its first kernel launch, and from the last
yield to the generator's end (what the GPU idles for between bench steps)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.forward_pass import get_forward_pass_factory
from raynet_amd.synthetic import make_synthetic_scene
H, W, V = 480, int(os.environ.get("W", "640")), 5
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H)
gp = GenerationParameters(depth_planes=64, neighbors=4, grid_shape=np.array([128]*3, np.int32),
                          max_number_of_marched_voxels=384, padding=11, gamma_mrf=0.05)
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
for _ in range(3): list(fp.forward_pass(scene, (0, V, 1)))
ctx = fp._ctx
first = {}
for name in ("scene_run", "scene_prepare_all", "scene_prepare"):
    if hasattr(ctx, name):
        orig = getattr(ctx, name)
        def wrap(*a, _o=orig, **k):
            first.setdefault("t", time.perf_counter())
            return _o(*a, **k)
        setattr(ctx, name, wrap)
pre, post = [], []
for _ in range(10):
    first.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = fp.forward_pass(scene, (0, V, 1))
    outs = []
    for o in g:
        outs.append(o)
        tl = time.perf_counter()
    t1 = time.perf_counter()
    pre.append((first["t"] - t0) * 1e3)
    post.append((t1 - tl) * 1e3)
print("entry -> first launch: %.3f ms (min %.3f)" % (np.mean(pre), np.min(pre)))
print("last yield -> return:  %.3f ms" % np.mean(post))
import cProfile, pstats
pr = cProfile.Profile()
def until_first():
    g = fp.forward_pass(scene, (0, V, 1))
    next(g)
    g.close()
torch.cuda.synchronize()
pr.enable()
for _ in range(5): until_first()
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(25)
