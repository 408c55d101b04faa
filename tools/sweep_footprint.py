"""CPU study (no GPU): which 128-byte feature vectors does the plane sweep touch, and how does the
ORDER in which an XCD's wavefronts take the rays decide what its 4 MB L2 has to hold?

For one reference image of the synthetic scene: every ray's D plane points projected into its N - 1
neighbour views (float64 here -- a footprint study, not an index map), then for a schedule (the
order in which consecutive wavefronts take rays, dealt to 8 XCDs in chunks) the stream of one XCD
is cut into windows of `inflight` rays (what its 32 CUs hold at once) and for consecutive windows:
  footprint = distinct vectors of a window (x 128 B: what the L2 must hold for the window to hit)
  new       = distinct vectors of a window that the previous window did not touch (~ its misses
              when two windows fit the L2)
Prints totals per schedule:  sum(new) x 128 B x 8 XCDs  =  the launch's L2-miss traffic estimate.
    python tools/sweep_footprint.py [config2|config4] [image]"""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd.synthetic import ring_cameras

cfgname = sys.argv[1] if len(sys.argv) > 1 else "config4"
ref = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if cfgname == "config4":
    H, W, V, D = 480, 640, 9, 128
else:
    H, W, V, D = 480, 640, 5, 64
pad = 11
Hf, Wf = H + pad + 1, W + pad + 1
cams = ring_cameras(V, H, W, focal=1.5 * H)
views = [ref] + [v for v in range(V) if v != ref]
bbox = np.array([-1, -1, -1, 1, 1, 1], np.float64)

# rays: idx = x * H + y
idx = np.arange(H * W)
px, py = (idx // H).astype(np.float64), (idx % H).astype(np.float64)
Pinv = np.asarray(cams[ref].P_pinv, np.float64)
cc = np.asarray(cams[ref].center, np.float64).ravel()[:3]
o = Pinv @ np.stack([px, py, np.ones_like(px)])
d = o[:3] / o[3] - cc[:, None]
with np.errstate(divide="ignore", invalid="ignore"):
    t1 = (bbox[:3, None] - cc[:, None]) / d
    t2 = (bbox[3:, None] - cc[:, None]) / d
tn = np.max(np.minimum(t1, t2), 0)
tf = np.min(np.maximum(t1, t2), 0)
hit = tf > tn
s = cc[:, None] + tn * d
e = cc[:, None] + tf * d
print("%s image %d: %d of %d rays hit the box" % (cfgname, ref, hit.sum(), H * W))

off = pad - (pad - 1) // 2
keys = np.zeros((H * W, D, V - 1), np.int32)      # vector index per (ray, plane, neighbour)
k = np.arange(D, dtype=np.float64) / (D - 1)
for j, v in enumerate(views[1:]):
    P = np.asarray(cams[v].P, np.float64)
    for c0 in range(0, H * W, 65536):
        sl = slice(c0, min(c0 + 65536, H * W))
        X = s[:, sl, None] + (e[:, sl] - s[:, sl])[:, :, None] * k[None, None, :]     # 3, n, D
        x = np.einsum("i,inD->nD", P[0, :3], X) + P[0, 3]
        y = np.einsum("i,inD->nD", P[1, :3], X) + P[1, 3]
        n = np.einsum("i,inD->nD", P[2, :3], X) + P[2, 3]
        with np.errstate(all="ignore"):
            fx = np.clip(np.nan_to_num(np.round(x / n)) + off, 0, W).astype(np.int64)
            fy = np.clip(np.nan_to_num(np.round(y / n)) + off, 0, H).astype(np.int64)
        z = (fx == 0) | (fy == 0)
        kk = fy * Wf + fx
        kk[z] = 0
        keys[sl, :, j] = (kk + j * Hf * Wf).astype(np.int32)
keys = keys.reshape(H * W, -1)
x_of, y_of = idx // H, idx % H


def patch_order(tile=16, along_rows=True):
    if along_rows:
        key = (((y_of // tile) * ((W + tile - 1) // tile) + x_of // tile) * tile + y_of % tile) * tile + x_of % tile
    else:
        key = (((x_of // tile) * ((H + tile - 1) // tile) + y_of // tile) * tile + x_of % tile) * tile + y_of % tile
    return np.argsort(key, kind="stable")


def strip_order(h, along_rows=True, seg=0):
    """strips of h image rows walked along x (seg > 0: in segments of seg columns, all h rows of
    a segment before the next)"""
    if along_rows:
        key = ((y_of // h) * W + x_of) * h + y_of % h
    else:
        key = ((x_of // h) * H + y_of) * h + x_of % h
    return np.argsort(key, kind="stable")


def study(name, order, chunk_rays, inflight=768):
    order = order[hit[order]]               # rays that miss the box are not swept
    n = len(order)
    nchunks = (n + chunk_rays - 1) // chunk_rays
    tot_new = tot_acc = 0
    fps = []
    for xcd in range(8):
        rays = np.concatenate([order[c * chunk_rays:(c + 1) * chunk_rays] for c in range(xcd, nchunks, 8)]
                              or [np.zeros(0, np.int64)])
        prev = np.zeros(0, np.int32)
        for w0 in range(0, len(rays), inflight):
            cur = np.unique(keys[rays[w0:w0 + inflight]])
            new = np.setdiff1d(cur, prev, assume_unique=True)
            tot_new += len(new)
            tot_acc += len(rays[w0:w0 + inflight]) * keys.shape[1]
            fps.append(len(cur))
            prev = cur
    fps = np.array(fps)
    print("%-44s chunk %5d rays: window footprint mean %.2f MB max %.2f MB | est. L2-miss traffic "
          "%.2f GB per image (accesses %.1f GB, unique %.2f GB)" % (
              name, chunk_rays, fps.mean() * 128 / 2 ** 20, fps.max() * 128 / 2 ** 20,
              tot_new * 128 / 1e9, tot_acc * 128 / 1e9, len(np.unique(keys[hit])) * 128 / 1e9))


study("16x16 patches along rows (today)", patch_order(16, True), 2048)
study("16x16 patches along rows", patch_order(16, True), 512)
for h in (1, 2, 4, 8):
    study("strips of %d rows, walked along x" % h, strip_order(h, True), h * W)
for h in (2, 4, 8):
    study("strips of %d cols, walked along y" % h, strip_order(h, False), h * H)
study("strips of 4 rows, 2 strips per chunk", strip_order(4, True), 2 * 4 * W)
study("strips of 2 rows, 2 strips per chunk", strip_order(2, True), 2 * 2 * W)
