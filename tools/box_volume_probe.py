"""Per-rank sums of the slab boxes' volumes (the scatter's flush work) next to the per-rank
voxel visits, for the shard bounds of WORLD ranks: why the rank with the top image strip is the
slowest although the visits are balanced, and what weight a box voxel would need."""
import os, sys, types
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import raynet_amd.forward_pass as F
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.synthetic import make_synthetic_scene
H, W, V, M = 480, 640, 5, 384
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=32, padding=11, focal=1.5 * H, seed=1234)
gp = GenerationParameters(depth_planes=64, neighbors=4, grid_shape=np.array([128] * 3, np.int32),
                          max_number_of_marched_voxels=M, padding=11, gamma_mrf=0.05)
fp = F.get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
for _ in fp.forward_pass(scene, (0, V, 1)):
    pass
torch.cuda.synchronize()
ctx, plan = fp._ctx, fp._plan
npad = plan["npad"]
nsl = (M + 15) // 16
boxes = ctx._slab_boxes[0][:V * npad // 64 * nsl * 2].view(V, npad // 64, nsl, 2).cpu().numpy()
lo, hi = boxes[..., 0], boxes[..., 1]
ok = hi >= 0
d = lambda a, sh: (a >> sh) & 1023
vol = np.where(ok, (d(hi, 20) - d(lo, 20) + 1) * (d(hi, 10) - d(lo, 10) + 1) * (d(hi, 0) - d(lo, 0) + 1), 0)
cnt = plan["rvc"].view(V, npad).cpu().numpy().astype(np.int64)
# slabs beyond a block's longest ray were never written
longest = cnt.reshape(V, npad // 64, 64).max(-1)
vol = np.where(np.arange(nsl)[None, None, :] * 16 < longest[..., None], vol, 0)
vol_blk = vol.sum(-1).sum(0)                       # [npad / 64] summed over slabs and images
cnt_blk = cnt.reshape(V, npad // 64, 64).sum(-1).sum(0)
print("box voxels per visit, whole scene: %.3f" % (vol_blk.sum() / cnt_blk.sum()))
for world in (4, 8):
    class FD(object):
        ReduceOp = types.SimpleNamespace(SUM=0, MIN=1, MAX=2)
        def all_reduce(self, t, op=None): pass
        def all_gather_into_tensor(self, out, inp):
            out.view(world, -1).copy_(inp.view(1, -1).expand(world, -1))
    F._dist = lambda w=world: (FD(), 0, w)
    fq = F.get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0)
    for _ in fq.forward_pass(scene, (0, V, 1)):
        pass
    cuts = fq._plan["bounds"][0]
    vs = [int(vol_blk[cuts[q] // 64:(cuts[q + 1] + 63) // 64].sum()) for q in range(world)]
    cs = [int(cnt_blk[cuts[q] // 64:(cuts[q + 1] + 63) // 64].sum()) for q in range(world)]
    print("world %d cuts %s" % (world, cuts))
    print("   visits / mean     %s" % np.round(np.array(cs) / np.mean(cs), 3).tolist())
    print("   box voxels / mean %s" % np.round(np.array(vs) / np.mean(vs), 3).tolist())
    print("   box voxels per visit %s" % np.round(np.array(vs) / np.array(cs), 3).tolist())

# merged boxes of the scatter's own tiles (128 rows x 32 steps): which overflow the 4096-voxel
# LDS box, and where in the row order they sit
lo3 = np.stack([d(lo, 20), d(lo, 10), d(lo, 0)], -1).astype(np.int64)
hi3 = np.stack([d(hi, 20), d(hi, 10), d(hi, 0)], -1).astype(np.int64)
valid = ok & (np.arange(nsl)[None, None, :] * 16 < longest[..., None])
lo3 = np.where(valid[..., None], lo3, 1 << 20)
hi3 = np.where(valid[..., None], hi3, -1)
nb, ns2 = (npad // 64) // 2 * 2, nsl // 2 * 2
L = lo3[:, :nb, :ns2].reshape(V, nb // 2, 2, ns2 // 2, 2, 3).min(axis=(2, 4))
Hh = hi3[:, :nb, :ns2].reshape(V, nb // 2, 2, ns2 // 2, 2, 3).max(axis=(2, 4))
mv = np.where(Hh[..., 0] >= 0, np.prod(np.maximum(Hh - L + 1, 0), -1), 0)      # [V, tiles, chunks]
over = mv > 4096
print("tiles of 128 rows: %d per image; chunks with a merged box > 4096 voxels: %d of %d" % (
    nb // 2, int(over.sum()), int((mv > 0).sum())))
for k in range(V):
    t, c = np.nonzero(over[k])
    if len(t):
        print("  image %d: tiles %s (rows %d .. %d), worst merged box %d voxels, chunks per such tile up to %d" % (
            k, sorted(set(t.tolist()))[:12], int(t.min()) * 128, int(t.max()) * 128 + 127, int(mv[k][over[k]].max()),
            int(np.bincount(t).max())))
