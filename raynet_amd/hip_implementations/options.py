"""PathOptions -- every schedule / A-B knob of the MI355X path in ONE object.

The reference has no counterpart (its only launch knob is `threads=2048`,
cuda_implementations/raynet_fp.py:288; everything else is baked into the PyCUDA source as
`$literals`).  No RESULT depends on an option beyond the summation order of the accumulator
scatter (and `deterministic`, which removes even that).  `RayNetForwardPass(..., options=...)`
takes one; `bench.py` echoes it in its JSON line so that an A/B run can be reproduced from the
line alone.  Environment variables are overrides only and are read HERE and nowhere else in the
package (`PathOptions.from_env`); the C library reads its own six in `rn_create`
(include/raynet_hip.h, `rn_options`) for callers that bind the C ABI directly -- the Python
mirror always sets them explicitly.
"""
import os
from dataclasses import asdict, dataclass, fields
from typing import Optional, Tuple


def _tile(text):
    """"16x16" -> (16, 16); "0" / "" / "none" -> None (ray-index order)."""
    text = str(text).strip().lower()
    if "x" not in text:
        return None
    a, b = text.split("x")
    return int(a), int(b)


def _flag(text):
    return str(text).strip().lower() not in ("0", "", "false", "no", "off")


@dataclass
class PathOptions:
    # ---- row layout and schedule of the resident buffers (host side) -------------------
    ray_tile: Optional[Tuple[int, int]] = (16, 16)   # pixel patch per 256 rows; None: ray-index order
    scatter_items: bool = True        # the box scatter takes a work list built from the rays' voxel
    #                                   counts (a tile's live chunks in pieces, longest first) from a
    #                                   plan's second pass on, instead of tiles x a fixed split
    slab_boxes: bool = True           # the scatters merge the traversal's slab boxes instead of scanning
    plan_path: bool = True            # one C call per phase of a pass (rn_scene_run) when the pass qualifies
    capture: str = "auto"             # the plan path's whole step (phases, exchanges, epilogue) as ONE
    #                                   captured HIP graph per plan, replayed per pass -- no interpreter and
    #                                   no launch overhead between the launches: "on", "off", or "auto" =
    #                                   with a process group only (one GPU gains nothing: the host runs
    #                                   ahead of a 6.7 ms step anyway; a 1 ms step of eight ranks does not
    #                                   wait for it: -5 %).  RCCL's collectives are captured with the
    #                                   launches; other transports keep the eager schedule
    maps: str = "lease"               # what a pass yields: "lease" = arrays that ARE the pinned host
    #                                   memory the GPU wrote, each holding a lease that ends when the
    #                                   array and all its views are garbage-collected -- no later pass
    #                                   touches leased memory (forward_pass._take_set): the reference's
    #                                   fresh-array semantics (.get(), forward_pass.py:739-744) without a
    #                                   copy; "copy" = pageable copies
    # ---- memory ---------------------------------------------------------------------------
    resident_gb: float = 0.0          # HBM budget of the resident schedule; 0: 90 % of what is free
    # ---- multi-GPU ------------------------------------------------------------------------
    deterministic: Optional[bool] = None   # 64-bit fixed-point sums: same bits for any run / rank
    #                                   count.  None (default) = ON when the pass runs on more than
    #                                   one rank -- the N-rank result then EQUALS the one-rank
    #                                   fixed-point result bit for bit, which is what SURVEY.md 8(e)
    #                                   asks of a sharded run (float sums only reach 32 ulp of the
    #                                   largest accumulator, ~1e-3) -- and off on one GPU (+2.5 %);
    #                                   True / False: as said, whatever the rank count
    shard: str = "voxels"             # "voxels": work-balanced cuts; "rays": equal ray counts
    shard_alpha: Optional[float] = None   # per-ray constant of the balance in units of the mean voxel
    #                                       count; None: derived from the shape (shard_alpha_for)
    exchange: str = "allreduce"       # deterministic mode only: "reduce_scatter" = int64 reduce-scatter,
    #                                   combine on the rank's slab, float all-gather (3/4 of the bytes)
    # ---- the context's own options (rn_options in include/raynet_hip.h) ---------------------
    scatter_mode: int = -1            # -1 by row layout, 0 slab scatter, 2 LDS-box scatter
    box_level: int = 0                # tile shape the adaptive box scatter starts from
    box_pin: bool = False             # stay at box_level
    overlap: int = 2                  # second stream: 0 off, 1 on, 2 when the scatter runs at level 1
    generic_sweep: bool = False       # reference-order plane sweep even for F = 32 (tests)
    sweep_rays_per_wave: int = 0      # cooperative sweep: 0 = by D (4 rays per wavefront for D <= 16, 2
    #                                   for D <= 32, else 1), 1 = always one (same bits; A/B and tests)

    # environment variable -> (field, parser); overrides only
    ENV = {
        "RAYNET_RAY_TILE": ("ray_tile", _tile),
        "RAYNET_SLAB_BOXES": ("slab_boxes", _flag),
        "RAYNET_SCATTER_ITEMS": ("scatter_items", _flag),
        "RAYNET_PLAN_PATH": ("plan_path", _flag),
        "RAYNET_CAPTURE": ("capture", lambda t: {"0": "off", "1": "on"}.get(str(t).strip(), str(t).strip())),
        "RAYNET_MAPS": ("maps", str),
        "RAYNET_RESIDENT_GB": ("resident_gb", float),
        "RAYNET_DETERMINISTIC": ("deterministic",
                                 lambda t: None if str(t).strip().lower() == "auto" else _flag(t)),
        "RAYNET_SHARD": ("shard", str),
        "RAYNET_SHARD_ALPHA": ("shard_alpha", float),
        "RAYNET_EXCHANGE": ("exchange", str),
        "RAYNET_HIP_SCATTER_MODE": ("scatter_mode", int),
        "RAYNET_HIP_BOX_LEVEL": ("box_level", int),
        "RAYNET_HIP_BOX_PIN": ("box_pin", _flag),
        "RAYNET_HIP_OVERLAP": ("overlap", lambda t: 1 if int(t) else 0),
        "RAYNET_HIP_GENERIC_SWEEP": ("generic_sweep", _flag),
        "RAYNET_HIP_SWEEP_RAYS_PER_WAVE": ("sweep_rays_per_wave", int),
    }

    def __post_init__(self):
        if self.ray_tile is not None:
            self.ray_tile = (int(self.ray_tile[0]), int(self.ray_tile[1]))
        assert self.shard in ("voxels", "rays"), self.shard
        assert self.exchange in ("allreduce", "reduce_scatter"), self.exchange
        assert self.scatter_mode in (-1, 0, 2) and self.box_level in (0, 1, 2)
        assert self.overlap in (0, 1, 2)
        assert self.sweep_rays_per_wave in (0, 1), self.sweep_rays_per_wave
        assert self.maps in ("lease", "copy"), self.maps
        assert self.capture in ("auto", "on", "off"), self.capture

    @classmethod
    def from_env(cls, environ=None, **overrides):
        """Defaults, then the environment's overrides, then explicit keyword overrides."""
        environ = os.environ if environ is None else environ
        kw = {}
        for name, (field, parse) in cls.ENV.items():
            if name in environ:
                kw[field] = parse(environ[name])
        kw.update({k: v for k, v in overrides.items() if v is not None})
        return cls(**kw)

    def fixed_point(self, world):
        """Whether a pass over `world` ranks sums in fixed point (`deterministic`, None = by world)."""
        return bool(world > 1) if self.deterministic is None else bool(self.deterministic)

    def replace(self, **kw):
        d = {f.name: getattr(self, f.name) for f in fields(self)}
        d.update(kw)
        return PathOptions(**d)

    def as_dict(self):
        d = asdict(self)
        d["ray_tile"] = "%dx%d" % self.ray_tile if self.ray_tile else None
        return d

    def context_options(self):
        """The six values of rn_options."""
        return (int(self.scatter_mode), int(self.box_level), 1 if self.box_pin else 0,
                int(self.overlap), 1 if self.generic_sweep else 0, int(self.sweep_rays_per_wave))

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name != "_key":
            object.__setattr__(self, "_key", None)      # (plans are keyed on key())

    def key(self):
        k = self.__dict__.get("_key")
        if k is None:
            # (what a plan's buffers and tables depend on: not how its passes are issued / awaited)
            k = tuple(sorted((n, v) for n, v in self.as_dict().items()
                             if n != "capture"))
            object.__setattr__(self, "_key", k)
        return k


def shard_alpha_for(n_views, depth_planes, mean_voxels):
    """Per-ray constant of the work balance, in units of the mean voxel count.

    A rank's work is  sum over its rays of (t_v * count + t_r):  t_v per traversed voxel (three BP
    sweeps + scatters, depth sweep, traversal, mapping) and t_r per ray (the plane sweep:
    projections and gathers of N views on D planes, independent of the ray's length).  Measured
    on MI355X (profiles/r02_k_*): t_r = 4.5 ps x N x D (config 2: 2.13 ms of
    plane sweep for 1.536 M rays at N x D = 320; config 4: N x D = 1152, 5.5 ns per ray) and t_v =
    25 - 31 ps per visit.  alpha = t_r / (t_v * mean count) = 0.16 N D / mean count: 0.37 at
    config 2, 0.68 at config 4 (round 2 used 0.6 everywhere: the border strips' ranks, many
    short rays, came out 8 % light at config 2)."""
    return 0.16 * float(n_views) * float(depth_planes) / max(float(mean_voxels), 1.0)
