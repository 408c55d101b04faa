#!/usr/bin/env python3
"""Golden vectors from a RUN OF THE REFERENCE'S OWN KERNELS on an MI355X.

oracle/build_ref_cu.py compiles the reference's .cu text -- placeholders substituted as
cuda_implementations/raynet_fp.py:230-248 does, nothing else changed -- for gfx950; this script
(run on the GPU box: `gpurun -- python tests/golden/gen_refcu_from_reference.py`) launches those
kernels through tests/ref_cu.py on seeded inputs and stores inputs + outputs as
tests/golden/ref_cu_gfx950.npz (written to gpurun_out/ on the box, copied into place afterwards).
Nothing of the reference travels or is stored but arrays.

Per case (`<case>/<field>`):
  inputs    M D N F H W padding bbox grid seed gamma P P_inv center ray_idxs, and `features_seed`
            (features = default_rng(seed).standard_normal((N, H+p+1, W+p+1, F), float32) * 0.25,
            as tests/test_hip_parity_gpu.make_case builds them)
  a1        points [64, D, 4], points_first / points_last [n, 3]
                                       batch_sample_points_in_bbox   (sampling_schemes.cu:92-122)
  a2        S_nofma [n, D]             batch_compute_similarities    (feature_similarities.cu:126-146)
                                       on `starts` / `ends` (the oracle's; starts == points_first);
            fma_rows, S_fma_rows, fma_other_max: the contracted build's output where it differs
            mvcnn_equals_a2            batch_multi_view_cnn_forward_pass (similarities.py:44-81)
                                       gives a2's bits (both builds)
  a3        rvi [n, M, 3] rvc [n]      batch_voxel_traversal         (ray_tracing.cu:145-163)
  a4        S_voxel [n, M]             batch_planes_voxels_mapping   (planes_voxels_mapping.cu:94-119)
  a5        msgs1 acc1_* msgs2 acc2_*     two sweeps of batch_belief_propagation (mrf_bp.cu:180-204) in
                                       the schedule of forward_pass.py:533-538, 678: acc_out starts
                                       at the prior, messages alias in and out; rays with fewer
                                       than 2 voxels are left out (`bp_valid`; SURVEY Q4: the
                                       kernel writes +inf for count == 1)
  a6        S_new [n, M]               batch_depth_estimation        (mrf_bp.cu:206-229)
The `_fma` arrays come from the default build (contraction on: what nvcc / PyCUDA would run, up to
the compiler's own choice of which products it fuses), everything else from -ffp-contract=off.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def features_for(seed, N, H, W, pad, F):
    rng = np.random.default_rng(int(seed))
    return rng.standard_normal((N, H + pad + 1, W + pad + 1, F), dtype=np.float32) * np.float32(0.25)


def sparse(acc, prior):
    """an accumulator as (flat [gx][gy][gz] index, value) of the voxels some ray touched"""
    a = acc.cpu().numpy().ravel()
    at = np.nonzero(a != np.float32(prior))[0].astype(np.int32)
    return at, a[at]


def cases():
    """The three shapes of the host cross-check (all H*W rays of view 0) and 2048 rays of
    config 2's view 0 (M-sized outputs for the first 256 of them)."""
    from conftest import load_cases
    from raynet_amd.synthetic import ring_cameras
    out = {}
    for name, c in load_cases("crosscheck_cu_host.npz").items():
        d = {k: c[k] for k in ("M", "D", "N", "F", "H", "W", "padding", "bbox", "grid", "seed", "gamma",
                               "P", "P_inv", "center")}
        d["ray_idxs"] = np.arange(int(c["H"]) * int(c["W"]), dtype=np.int32)
        d["m_rays"] = len(d["ray_idxs"])
        out[name] = d
    H, W, N = 480, 640, 5
    cams = ring_cameras(N, H, W, focal=1.5 * H)        # the cameras of bench.py's scene
    rng = np.random.default_rng(77)
    out["config2"] = dict(
        M=384, D=64, N=N, F=32, H=H, W=W, padding=11, bbox=np.array([-1, -1, -1, 1, 1, 1], np.float32),
        grid=np.array([128, 128, 128], np.int32), seed=2024, gamma=np.float32(0.05),
        P=np.stack([np.asarray(c.P, np.float32) for c in cams]),
        P_inv=np.asarray(cams[0].P_pinv, np.float32), center=np.asarray(cams[0].center, np.float32).ravel(),
        ray_idxs=rng.choice(H * W, size=2048, replace=False).astype(np.int32), m_rays=256)
    return out


def main():
    import torch
    import ref_cu
    from oracle import oracle
    assert torch.cuda.is_available() and ref_cu.available()
    flat = {}
    for name, c in cases().items():
        M, D, N, F, H, W, pad = (int(c[k]) for k in ("M", "D", "N", "F", "H", "W", "padding"))
        feats = features_for(c["seed"], N, H, W, pad, F)
        o = oracle.Oracle(M=M, D=D, N=N, F=F, H=H, W=W, padding=pad, bbox=c["bbox"], grid_shape=c["grid"],
                          threads=oracle.Oracle.max_threads())
        vg = oracle.voxel_grid_centers(c["bbox"], c["grid"])
        starts, ends = o.sample(c["ray_idxs"], c["P_inv"], c["center"])
        res = {}
        mods = {v: ref_cu.RefCu(name, v) for v in ("nofma", "fma")}
        r = mods["nofma"]
        f_d, P_d = r.dev(feats), r.dev(c["P"].reshape(-1))
        pts = r.sample_points(c["ray_idxs"], c["P_inv"].reshape(-1), c["center"]).cpu().numpy()
        # [n, D, 4] is the bulk of the file and a function of its first and last point: all D
        # points of the first 64 rays, first / last point of every ray
        res["points"] = pts[:64]
        res["points_first"], res["points_last"] = pts[:, 0, :3].copy(), pts[:, -1, :3].copy()
        res["starts"], res["ends"] = starts, ends
        S = {}
        for v, m in mods.items():
            S[v] = m.similarities(f_d, P_d, starts, ends).cpu().numpy()
            S["mvcnn_" + v] = m.mvcnn_similarities(c["ray_idxs"], f_d, P_d, c["P_inv"].reshape(-1),
                                                   c["center"]).cpu().numpy()
        res["S_nofma"] = S["nofma"]
        # a1 + a2 in one kernel: the same bits as a2 on a1's end points (the oracle's: bit-equal
        # to points_first / the kernel's own ray_end, checked by the tests), so only the verdict
        res["mvcnn_equals_a2"] = np.array([np.array_equal(S["mvcnn_" + v], S[v]) for v in ("nofma", "fma")])
        # the contracted build: the rays it moves by more than 1e-6, and the largest move of the rest
        d = np.abs(S["fma"] - S["nofma"]).max(1)
        rows = np.nonzero(d > 1e-6)[0].astype(np.int32)
        res["fma_rows"], res["S_fma_rows"] = rows, S["fma"][rows]
        res["fma_other_max"] = np.float32(d[d <= 1e-6].max() if (d <= 1e-6).any() else 0.0)
        res["S_fma"] = S["fma"]
        k = int(c["m_rays"])
        s_d, e_d = r.dev(starts[:k]), r.dev(ends[:k])
        rvi, rvc = r.traversal(s_d, e_d)
        S_d = r.dev(res["S_nofma"][:k])
        Sv = r.planes_to_voxels(vg, rvi, rvc, s_d, e_d, S_d)
        res["rvi"], res["rvc"] = rvi.cpu().numpy().astype(np.int16), rvc.cpu().numpy()
        res["S_voxel"] = Sv.cpu().numpy()
        ok = (rvc >= 2)
        sel = torch.nonzero(ok).ravel()
        rvi_v, rvc_v = rvi[sel].contiguous(), rvc[sel].contiguous()
        prior = float(np.float32(np.log(c["gamma"]) - np.log(1 - c["gamma"])))
        grid = tuple(int(g) for g in c["grid"])
        acc0 = torch.full(grid, prior, device="cuda")
        msgs = torch.zeros((len(sel), M), device="cuda")
        acc1 = torch.full(grid, prior, device="cuda")
        r.bp_sweep(Sv[sel].clone(), rvi_v, rvc_v, acc0, msgs, acc1)      # S is clipped in place
        res["msgs1"] = msgs.cpu().numpy().copy()
        res["acc1_at"], res["acc1_val"] = sparse(acc1, prior)
        acc2 = torch.full(grid, prior, device="cuda")
        r.bp_sweep(Sv[sel].clone(), rvi_v, rvc_v, acc1, msgs, acc2)
        res["msgs2"] = msgs.cpu().numpy().copy()
        res["acc2_at"], res["acc2_val"] = sparse(acc2, prior)
        res["S_new"] = r.depth_estimation(Sv[sel].clone(), rvi_v, rvc_v, acc2, msgs).cpu().numpy()
        res["bp_valid"] = ok.cpu().numpy()
        # what the box says about the oracle, for the log
        So = o.similarities(feats, c["P"], starts, ends)
        d_no = np.abs(So - res["S_nofma"]).max()
        ray_f = np.abs(res["S_fma"] - res["S_nofma"]).max(1) > 1e-5
        print("%-8s n=%d  |oracle - ref(nofma)| max %.3g   rays where fma/nofma builds differ > 1e-5: %d "
              "(%.3f %%)   mean count %.1f  bp-valid %d" % (
                  name, len(starts), d_no, int(ray_f.sum()), 100.0 * ray_f.mean(),
                  float(res["rvc"].mean()), int(ok.sum())), flush=True)
        for key in ("M", "D", "N", "F", "H", "W", "padding", "bbox", "grid", "seed", "gamma", "P", "P_inv",
                    "center", "ray_idxs", "m_rays"):
            flat["%s/%s" % (name, key)] = np.asarray(c[key])
        res.pop("S_fma")
        for key, v in res.items():
            flat["%s/%s" % (name, key)] = np.asarray(v)
    out_dir = os.path.join(REPO, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "ref_cu_gfx950.npz")
    np.savez_compressed(out, **flat)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
