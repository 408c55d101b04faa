"""Seeded synthetic scene for benchmarks and parity tests.

N pinhole cameras on a ring around the bounding box look at the origin.  Instead
of images + a trained MV-CNN (the reference ships no weights), each view gets a
feature map with *planted surfaces*: the feature at a pixel is a smooth function of
the 3-D point its ray first hits on a sphere or on a ground disc (both inside the
bounding box and visible from every camera), plus noise; rays that hit nothing carry
noise only.  The same surface point therefore has the same feature in every view
that sees it, so the plane-sweep similarity has a true peak and BP has something to
converge to, while nothing outside the box pretends to be a surface.
"""
import numpy as np
import torch

from .common.camera import Camera
from .common.scene import Image, Scene


class _FeatureOnlyImage(Image):
    def __init__(self, height, width, camera):
        self.image = None
        self.camera = camera
        self.height, self.width = height, width


class FeatureBank(object):
    """Stands where the Keras model stands in the reference drivers: it hands out the
    per-view feature map [H+p+1, W+p+1, F] (what `model.predict` on the zero-padded
    image returns, forward_pass.py:622-624)."""

    def __init__(self, maps):
        self._maps = maps

    def view_features(self, scene, view):
        return self._maps[view]

    def stacked(self, views):
        return torch.stack([self._maps[v] for v in views]).contiguous()


def ring_cameras(n_views, H, W, radius=3.0, focal=None, arc=2 * np.pi, heights=None):
    focal = 0.85 * H if focal is None else focal
    cams = []
    for v in range(n_views):
        a = arc * v / n_views
        h = (0.3 + 0.2 * v) if heights is None else heights[v]
        cams.append(Camera.look_at([radius * np.cos(a), radius * np.sin(a), h], [0, 0, 0],
                                   focal, H, W))
    return cams


def planted_feature_maps(cameras, H, W, F=32, padding=11, seed=1234, sphere_radius=0.5,
                         sphere_center=(0.0, 0.0, -0.1), ground_z=-0.7, ground_radius=0.95,
                         noise=0.25, device="cuda"):
    """One [H+p+1, W+p+1, F] float32 map per camera."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    freq = (torch.randn((3, F), generator=g) * 2.5).to(device)
    phase = (torch.rand((F,), generator=g) * 2 * np.pi).to(device)
    Hf, Wf = H + padding + 1, W + padding + 1
    off = padding - (padding - 1) // 2        # feature index = round(pixel) + off
    fy, fx = torch.meshgrid(torch.arange(Hf, device=device, dtype=torch.float64),
                            torch.arange(Wf, device=device, dtype=torch.float64), indexing="ij")
    pix = torch.stack([fx - off, fy - off, torch.ones_like(fx)], dim=-1)      # (Hf, Wf, 3)
    sc = torch.tensor(sphere_center, dtype=torch.float64, device=device)
    maps = []
    for k, cam in enumerate(cameras):
        Kinv = torch.tensor(np.linalg.inv(cam.K), dtype=torch.float64, device=device)
        Rt = torch.tensor(np.asarray(cam.R, np.float64).T, dtype=torch.float64, device=device)
        o = torch.tensor(np.asarray(cam.center, np.float64).ravel()[:3], device=device)
        d = (pix @ Kinv.T) @ Rt.T
        d = d / d.norm(dim=-1, keepdim=True)
        # sphere
        oc = o - sc
        b = (d * oc).sum(-1)
        disc = b * b - ((oc * oc).sum() - sphere_radius ** 2)
        t_s = -b - torch.sqrt(torch.clamp(disc, min=0))
        hit_s = (disc > 0) & (t_s > 0)
        # ground disc z = ground_z, seen from above
        t_g = (ground_z - o[2]) / d[..., 2]
        Xg = o + t_g[..., None] * d
        hit_g = (d[..., 2] < 0) & (t_g > 0) & ((Xg[..., 0] ** 2 + Xg[..., 1] ** 2) < ground_radius ** 2)
        inf = torch.full_like(t_s, float("inf"))
        t = torch.minimum(torch.where(hit_s, t_s, inf), torch.where(hit_g, t_g, inf))
        hit = torch.isfinite(t)
        X = (o + torch.where(hit, t, torch.zeros_like(t))[..., None] * d).to(torch.float32)
        f = torch.cos(X @ freq + phase) * hit[..., None].to(torch.float32)
        gk = torch.Generator(device="cpu").manual_seed(seed + 1000 + k)
        n = torch.randn((Hf, Wf, F), generator=gk).to(device)
        maps.append((f + noise * n).to(torch.float32).contiguous())
    return maps


def make_synthetic_scene(H=480, W=640, n_views=5, F=32, padding=11,
                         bbox=(-1, -1, -1, 1, 1, 1), seed=1234, device="cuda", radius=3.0,
                         focal=None, arc=2 * np.pi):
    """-> (Scene, FeatureBank).  Scene images carry cameras only (features replace them)."""
    cams = ring_cameras(n_views, H, W, radius=radius, focal=focal, arc=arc)
    scene = Scene([_FeatureOnlyImage(H, W, c) for c in cams], bbox)
    bank = FeatureBank(planted_feature_maps(cams, H, W, F=F, padding=padding, seed=seed,
                                            device=device))
    return scene, bank
