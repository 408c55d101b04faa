"""How far the float-mode eight-rank run of config 2 lands from the one-rank fixed-point run, over
repeated runs (float atomics and the gloo all-reduce re-associate differently every time): the
tolerances of tests/test_config3_gpu.py come from here."""
import sys, os, tempfile, pathlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import test_config3_gpu as T

def main():
    from oracle import oracle
    from raynet_amd.forward_pass import get_forward_pass_factory
    c, scene, bank, gp = T._scene("config2")
    V, H, W, M = c["V"], c["H"], c["W"], c["M"]
    one = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0, deterministic=True)
    depth_1 = np.stack(list(one.forward_pass(scene, (0, V, 1))))
    acc_1 = one.accumulator.cpu().numpy()
    o = oracle.Oracle(M=M, D=c["D"], N=5, F=32, H=H, W=W, padding=11, bbox=scene.bbox.ravel(), grid_shape=c["grid"], threads=oracle.Oracle.max_threads())
    vg = oracle.voxel_grid_centers(scene.bbox.ravel(), c["grid"])
    for rep in range(6):
        d = pathlib.Path(tempfile.mkdtemp())
        acc_f, depth_f, _ = T._run_ranks(d, "config2", False)
        ulp = np.abs(acc_f - acc_1).max() / np.spacing(np.abs(acc_1).max())
        gaps = []
        for r in range(V):
            dd = np.abs(depth_f[r] - depth_1[r]).T.ravel()
            bad = np.where(dd > 1e-4)[0]
            if not len(bad): continue
            views = scene.view_indices_with_neighbors(r, 4)
            f = bank.stacked(views).cpu().numpy()
            P = np.array([scene.get_image(v).camera.P for v in views], np.float32)
            Pi = scene.get_image(r).camera.P_pinv.astype(np.float32); cc = scene.get_image(r).camera.center.ravel().astype(np.float32)
            rows = {int(q): k for k, q in enumerate(one.ray_index[r].cpu().numpy())}
            for idx in bad:
                rvi, rvc, Sv = o.fused_bp(np.array([idx], np.int32), f, P, Pi, cc, vg, o.prior(0.05), np.zeros((1, M), np.float32), o.prior(0.05))
                m = one.messages[r][rows[int(idx)]].cpu().numpy()[None]
                top = np.sort(o.depth_distribution(Sv, rvi, rvc, acc_1, m)[0])[::-1]
                gaps.append(float(top[0] - top[1]))
        print("rep", rep, "acc diff %.2f ulp (%.3g)" % (ulp, np.abs(acc_f - acc_1).max()), "differing", len(gaps), "gaps", ["%.2g" % g for g in gaps], flush=True)


if __name__ == "__main__":
    main()
