"""Host time at the boundary between two passes: from the moment the last map's copy is done to
the first launch of the next pass (the GPU idles for it), and the host time of a whole pass."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.forward_pass import get_forward_pass_factory
from raynet_amd.synthetic import make_synthetic_scene
H, W, V = 480, 640, 5
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H)
gp = GenerationParameters(depth_planes=64, neighbors=4, grid_shape=np.array([128]*3, np.int32),
                          max_number_of_marched_voxels=384, padding=11, gamma_mrf=0.05)
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
for _ in range(3): list(fp.forward_pass(scene, (0, V, 1)))
ctx = fp._ctx
marks = []
orig = ctx.scene_prepare_all
def first_launch(*a, **k):
    marks.append(("first_launch", time.perf_counter()))
    return orig(*a, **k)
ctx.scene_prepare_all = first_launch
orig_sync = torch.cuda.Event.synchronize
def sync(self):
    r = orig_sync(self)
    marks.append(("synced", time.perf_counter()))
    return r
torch.cuda.Event.synchronize = sync
orig_depth = ctx.scene_depth
def depth(*a, **k):
    r = orig_depth(*a, **k)
    marks.append(("depth_launched", time.perf_counter()))
    return r
ctx.scene_depth = depth
torch.cuda.synchronize()
for _ in range(6):
    marks.append(("enter", time.perf_counter()))
    for m in fp.forward_pass(scene, (0, V, 1)):
        pass
    marks.append(("leave", time.perf_counter()))
t0 = marks[0][1]
last = None
rows = []
for name, t in marks:
    rows.append((name, (t - t0) * 1e3))
# print the last two passes
idx = [i for i, (n, _) in enumerate(rows) if n == "enter"]
prev = rows[idx[-2]][1]
for name, t in rows[idx[-2]:]:
    print("%-15s %9.3f ms  (+%.3f)" % (name, t, t - prev))
    prev = t

# the prologue of a pass, piece by piece (host time, cached plan)
import raynet_amd.forward_pass as F
refs = list(range(V))
def timeit(f, n=200):
    t = time.perf_counter()
    for _ in range(n):
        r = f()
    return (time.perf_counter() - t) / n * 1e6, r
us, bank_ = timeit(lambda: fp._view_features(scene, refs))
print("_view_features        %7.1f us" % us)
Fdim = next(iter(bank_.values())).shape[-1]
us, ctx_ = timeit(lambda: fp._context(scene, Fdim))
print("_context              %7.1f us" % us)
us, bank2 = timeit(lambda: {v: f.to(ctx_.device, torch.float32).contiguous() for v, f in bank_.items()})
print("bank .to().contiguous %7.1f us" % us)
us, _ = timeit(lambda: fp._prior())
print("_prior                %7.1f us" % us)
d = F._dist()
us, plan = timeit(lambda: fp._build_plan(scene, refs, bank2, ctx_, *d))
print("_build_plan (cached)  %7.1f us" % us)
us, _ = timeit(lambda: F._dist())
print("_dist                 %7.1f us" % us)
us, _ = timeit(lambda: torch.cuda.current_stream(ctx_.device).wait_stream(fp._side_stream))
print("wait_stream(side)     %7.1f us" % us)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    fp._build_plan(scene, refs, bank2, ctx_, *d)
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(12)
