"""The reference-named convenience wrappers around K5 / K6 that no other test calls:
`perform_ray_marching` + `get_voxel_traversal_backend` (raynet/ray_marching/ray_marching.py:46-90,
ray_tracing_cuda.py:92-143) and `depth_to_voxels` + `get_depth_to_voxels_backend`
(raynet/planes_voxels_mapping/planes_voxels_mapping_cuda.py:70-124, depth_to_voxels.py:4-39)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(oracle_mod):
    import torch
    assert torch.cuda.is_available()
    from raynet_amd import _lib
    _lib.build()
    from raynet_amd.synthetic import make_synthetic_scene
    H, W = 20, 28
    scene, _ = make_synthetic_scene(H=H, W=W, n_views=3, focal=1.5 * H)
    return scene, H, W


def test_perform_ray_marching_and_selector(setup, oracle_mod):
    from raynet_amd.ray_marching.ray_marching import get_voxel_traversal_backend
    scene, H, W = setup
    grid = np.array([32, 32, 32], np.int32)
    M = 128
    rays = np.arange(0, H * W, 3, dtype=np.int32)
    march = get_voxel_traversal_backend("hip")
    rvi, rvc = march(scene, 1, M, rays, grid, batch_size=100)        # several batches
    assert rvi.shape == (len(rays), M, 3) and rvi.dtype == np.int32 and rvc.dtype == np.int32
    cam = scene.get_image(1).camera
    o = oracle_mod.Oracle(M=M, D=8, N=2, F=4, H=H, W=W, padding=1, bbox=scene.bbox.ravel(),
                          grid_shape=grid)
    s, e = o.sample(rays, np.asarray(cam.P_pinv, np.float32), np.asarray(cam.center, np.float32).ravel())
    rvi_o, rvc_o = o.traversal(s, e)
    assert np.array_equal(rvc, rvc_o) and np.array_equal(rvi, rvi_o)      # bit-exact index maps
    assert rvc.max() > 20
    # a ray that fills all M slots is an error, as in ray_marching.py:41-42
    with pytest.raises(ValueError, match="Nr="):
        march(scene, 1, int(rvc.max()), rays, grid)
    for name in ("cython", "cuda", "numpy"):
        with pytest.raises(NotImplementedError):
            get_voxel_traversal_backend(name)


def test_depth_to_voxels_and_selector(setup, oracle_mod):
    from raynet_amd.common.scene import get_voxel_grid
    from raynet_amd.planes_voxels_mapping.depth_to_voxels import get_depth_to_voxels_backend
    from raynet_amd.ray_marching.ray_marching import get_voxel_traversal_backend
    scene, H, W = setup
    grid = np.array([32, 32, 32], np.int32)
    M, D = 128, 16
    n = H * W
    all_rays = np.arange(n, dtype=np.int32)
    rvi, rvc = get_voxel_traversal_backend("hip")(scene, 0, M, all_rays, grid)
    cam = scene.get_image(0).camera
    o = oracle_mod.Oracle(M=M, D=D, N=2, F=4, H=H, W=W, padding=1, bbox=scene.bbox.ravel(),
                          grid_shape=grid)
    s, e = o.sample(all_rays, np.asarray(cam.P_pinv, np.float32),
                    np.asarray(cam.center, np.float32).ravel())
    k = np.arange(D, dtype=np.float32)[None, :, None]
    pts = np.ones((4, n, D), np.float32)              # the reference's [4, N, D] layout
    pts[:3] = (s[:, None, :] + k * (e - s)[:, None, :] / np.float32(D - 1)).transpose(2, 0, 1)
    rng = np.random.default_rng(2)
    S = rng.random((n, D)).astype(np.float32)
    S /= S.sum(1, keepdims=True)
    vg = get_voxel_grid(scene.bbox, grid)             # [3, gx, gy, gz]
    sel = np.sort(rng.choice(n, 150, replace=False))
    S_new = np.full((n, M), 7.0, np.float32)          # must be zero-filled by the call
    out = get_depth_to_voxels_backend("hip", rvc, rvi, sel, vg, pts, S, S_new)
    assert out is S_new
    want = o.planes_to_voxels(np.ascontiguousarray(vg.transpose(1, 2, 3, 0)), rvi[sel], rvc[sel],
                              s[sel], e[sel], S[sel])
    assert np.abs(S_new[sel] - want).max() <= 5e-7       # values <= 1: a few fp32 ulp
    rest = np.setdiff1d(all_rays, sel)
    assert np.all(S_new[rest] == 0)
    hit = rvc[sel] > 0
    assert np.abs(S_new[sel][hit].sum(1) - 1).max() < 1e-5
    with pytest.raises(NotImplementedError):
        get_depth_to_voxels_backend("numpy", rvc, rvi, sel, vg, pts, S, S_new)
