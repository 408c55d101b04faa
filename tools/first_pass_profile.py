import cProfile, pstats, sys, time, io
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.forward_pass import get_forward_pass_factory
from raynet_amd.synthetic import make_synthetic_scene, _FeatureOnlyImage, ring_cameras
from raynet_amd.common.scene import Scene
H, W, V = 480, 640, 5
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H, seed=1234)
gp = GenerationParameters(depth_planes=64, neighbors=4, grid_shape=np.array((128,128,128), np.int32), max_number_of_marched_voxels=384, padding=11, gamma_mrf=0.05)
fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
def step(sc):
    for _ in fp.forward_pass(sc, (0, V, 1)): pass
step(scene); torch.cuda.synchronize()
for rep in range(3):
    sc = Scene([_FeatureOnlyImage(H, W, c) for c in ring_cameras(V, H, W, focal=1.5 * H)], scene.bbox)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if rep == 2:
        pr = cProfile.Profile(); pr.enable()
    step(sc)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    if rep == 2:
        pr.disable()
    t2 = time.perf_counter()
    print("first pass on a new scene: host returns after %.2f ms, GPU done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    t0 = time.perf_counter(); step(sc); torch.cuda.synchronize(); print("  second pass %.2f ms" % ((time.perf_counter() - t0) * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
