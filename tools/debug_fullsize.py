"""Debug helper: run the resident path at full size step by step and locate
non-finite messages; cross-check offending rays against the oracle."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raynet_amd.hip_implementations import get_context
from raynet_amd.synthetic import make_synthetic_scene
from oracle import oracle

H, W, D, M, grid = 480, 640, 64, 384, (128, 128, 128)
V = 5
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H)
ctx = get_context(M, D, 5, 32, H, W, 11, scene.bbox.ravel(), grid)
vg = oracle.voxel_grid_centers(scene.bbox.ravel(), grid)
ctx.set_voxel_grid(torch.from_numpy(vg).cuda())
prior = float(np.float32(np.log(0.05) - np.log(0.95)))
st = {}
for r in range(V):
    views = scene.view_indices_with_neighbors(r, 4)
    P = ctx.dev(np.array([scene.get_image(v).camera.P for v in views], np.float32))
    Pi = ctx.dev(scene.get_image(r).camera.P_pinv.astype(np.float32))
    cc = ctx.dev(scene.get_image(r).camera.center.ravel().astype(np.float32))
    n = H * W
    ridx = torch.arange(n, dtype=torch.int32, device="cuda")
    vox = torch.zeros((n, M), dtype=torch.int32, device="cuda")
    rvc = torch.zeros((n,), dtype=torch.int32, device="cuda")
    Sr = torch.zeros((n, M), device="cuda")
    ctx.scene_prepare(ridx, [bank.view_features(scene, v) for v in views], P, Pi, cc, vox, rvc, Sr)
    st[r] = dict(vox=vox, rvc=rvc, Sr=Sr, msgs=torch.zeros((n, M), device="cuda"), cc=cc)
    print("image", r, "mean count", float(rvc.float().mean()), "max", int(rvc.max()),
          "Sr finite", bool(torch.isfinite(Sr).all()), "Sr min", float(Sr.min()), "max", float(Sr.max()))
acc_in = torch.full(grid, prior, device="cuda")
part = torch.zeros((ctx.acc_copies(),) + grid, device="cuda")
acc_next = torch.empty(grid, device="cuda")
o = oracle.Oracle(M=M, D=D, N=5, F=32, H=H, W=W, padding=11, bbox=scene.bbox.ravel(), grid_shape=grid)
for it in range(3):
    for r in range(V):
        s = st[r]
        before = s["msgs"].clone()
        ctx.scene_bp_sweep(s["Sr"], s["vox"], s["rvc"], acc_in, s["msgs"], part)
        bad = ~torch.isfinite(s["msgs"])
        print("it", it, "img", r, "nonfinite msgs", int(bad.sum()), "absmax finite",
              float(s["msgs"][~bad].abs().max()))
        if bad.any():
            rows = bad.any(1).nonzero().ravel()[:3].cpu().numpy()
            for row in rows:
                c = int(s["rvc"][row])
                v = s["vox"][row, :c].cpu().numpy().astype(np.int64)
                rvi = np.stack([v >> 20, (v >> 10) & 1023, v & 1023], -1).astype(np.int32)
                rvi_full = np.zeros((1, M, 3), np.int32); rvi_full[0, :c] = rvi
                Srow = s["Sr"][row].cpu().numpy()[None]
                m_in = before[row].cpu().numpy()[None].copy()
                out = np.zeros(grid, np.float32)
                o.bp_sweep(Srow, rvi_full, np.array([c], np.int32), acc_in.cpu().numpy(), m_in, out)
                hip = s["msgs"][row, :c].cpu().numpy()
                print(" ray", row, "count", c, "oracle nonfinite", int((~np.isfinite(m_in[0, :c])).sum()),
                      "hip nonfinite", int((~np.isfinite(hip)).sum()))
                k = np.where(~np.isfinite(hip))[0][:5]
                print("  idx", k, "hip", hip[k], "oracle", m_in[0, k])
                a = acc_in.cpu().numpy()[tuple(rvi.T)]
                print("  acc at idx", a[k], "msg_in", before[row, :c].cpu().numpy()[k], "Sr", Srow[0, k])
            sys.exit(0)
    ctx.acc_combine(part, prior, acc_next)
    acc_in, acc_next = acc_next, acc_in
    print("it", it, "acc finite", bool(torch.isfinite(acc_in).all()), float(acc_in.min()), float(acc_in.max()))
