"""Forward-pass drivers -- mirror of raynet/forward_pass.py.

`get_forward_pass_factory(name)` returns a class with the reference's constructor
`(model, generation_params, sampling_scheme, image_shape, rays_batch,
filter_out_rays=False)` whose `forward_pass(scene, (start, end, skip))` is a
generator yielding one (H, W) float32 depth map per reference image
(forward_pass.py:208-223, :744).

RayNetForwardPass runs the schedule of forward_pass.py:579-748 (bp_iterations
sweeps over all reference images, accumulator swap + prior refill, then a depth
sweep) in one of two ways:

  schedule="resident" (default, the MI355X-native path)
      Features of every view are computed once; per reference image the K1 prefix
      (sample, plane sweep, traversal, mapping, clip+renorm) runs once and its
      per-ray columns stay in HBM together with the messages; each BP sweep and
      the depth sweep stream them.  The reference recomputes all of it four
      times and round-trips messages through a disk memmap.
  schedule="reference"
      Literal K1 / K2 launches per ray batch, everything recomputed per sweep,
      like the reference (used by the parity tests and as an A/B in bench.py).

Both give the same result (tests/test_forward_pass_gpu.py).  Decisions on the
reference driver's quirks: messages persist across
iterations (Q1) and every image uses its own messages in the depth sweep (Q2);
`reference_quirks=True` reproduces the shipped behaviour instead.

With torch.distributed initialised, the rays of every reference image are
sharded contiguously over the ranks; each rank scatters into its own zeroed
accumulator and one all-reduce (RCCL) per BP iteration merges them before the
prior is added once.
"""
import ctypes
import itertools
import weakref

import numpy as np
import torch

from . import _lib
from .hip_implementations.options import PathOptions, shard_alpha_for
from .hip_implementations.mvcnn_with_ray_marching_and_voxels_mapping import \
    batch_mvcnn_voxel_traversal_with_ray_marching_with_depth_estimation
from .hip_implementations.raynet_fp import perform_raynet_fp
from .hip_implementations.similarities import \
    perform_multi_view_cnn_forward_pass_with_depth_estimation


_SIDE_STREAMS = {}


def _side_streams(dev):
    """(exchange / stitch stream, copy stream) of a device, created ONCE per process: HIP maps
    streams onto a handful of hardware queues in creation order, and a side stream that lands on
    the main stream's queue serialises with it (measured: every second driver object of a
    process, each creating its own pair, ran 0.25 ms per step slower)."""
    key = (dev.type, dev.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return _SIDE_STREAMS[key]


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def _backend_of(dist):
    """"nccl" (= RCCL), "gloo", ... of the default group; a stand-in object (tools/shard_proxy.py)
    answers for itself."""
    try:
        return str(dist.get_backend())
    except Exception:
        return ""


def map_owner(k, n_images, world):
    """Rank that assembles image k's depth map when the rays are sharded (the reference needs a
    map once, forward_pass.py:739-744): the images dealt out in contiguous blocks -- monotone in
    k, so a rank's depth rows are already ordered by destination."""
    return (k * world) // n_images


def sweep_direction(H, W, images):
    """"rows" if the epipolar lines of the neighbour views run along image rows of the
    reference view, else "cols".

    A ray reads the feature vectors along its epipolar line in every neighbour view.
    Rays whose pixels lie ALONG an epipolar line share those lines, so walking the
    image in that direction keeps a few feature rows per view hot in the XCD's L2
    instead of streaming the whole map for every image column."""
    ref = images[0].camera
    P = np.asarray(ref.P, np.float64)
    dx = dy = 0.0
    for im in images[1:]:
        e = P.dot(np.asarray(im.camera.center, np.float64).reshape(4))    # epipole
        if abs(e[2]) > 1e-9 * (abs(e[0]) + abs(e[1]) + 1e-30):
            v = e[:2] / e[2] - np.array([W / 2.0, H / 2.0])
        else:
            v = e[:2]
        nrm = np.hypot(v[0], v[1]) + 1e-30
        dx += abs(v[0]) / nrm
        dy += abs(v[1]) / nrm
    return "rows" if dx > dy else "cols"


def tile_order(ray_idxs, H, W, tile_x, tile_y, along_rows=False):
    """The ray list (idx = x*H + y, sampling_schemes.cu:5-8) re-ordered into tile_x x tile_y
    pixel patches: consecutive rows of the per-ray buffers are then neighbouring rays in BOTH
    image directions, so a scatter tile sums ~11 rays per voxel in LDS before it touches the
    accumulator (k_scatter_box).  along_rows: patches (and the pixels inside them) are
    enumerated along image rows instead of columns -- the rays a GPU works on at the same
    time then share the epipolar lines of neighbour views that lie along image rows (see
    sweep_direction).  Row order never changes results -- every per-ray quantity is computed
    from the ray index -- only where a ray's row lives."""
    idx = ray_idxs.to(torch.int64)
    x, y = idx // H, idx % H
    if along_rows:
        tiles_x = (W + tile_x - 1) // tile_x
        key = (((y // tile_y) * tiles_x + x // tile_x) * tile_y + y % tile_y) * tile_x + x % tile_x
    else:
        tiles_y = (H + tile_y - 1) // tile_y
        key = (((x // tile_x) * tiles_y + y // tile_y) * tile_x + x % tile_x) * tile_y + y % tile_y
    return ray_idxs[torch.argsort(key)].to(torch.int32).contiguous()


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of an n-long ray list owned by `rank`."""
    return (n * rank) // world, (n * (rank + 1)) // world


class ForwardPass(object):
    """forward_pass.py:25-223."""

    def __init__(self, model, generation_params, sampling_scheme, image_shape,
                 rays_batch=50000, filter_out_rays=False):
        self._model = model
        self._generation_params = generation_params
        self._sampling_scheme = sampling_scheme
        self.rays_batch = rays_batch
        self._filter_out_rays = filter_out_rays
        self._fp = None

    def get_valid_rays_per_image(self, scene, i):
        H, W = scene.image_shape
        idxs = np.arange(H * W, dtype=np.int32)
        if self._filter_out_rays:
            idxs = idxs.reshape(W, H).T
            G = scene.get_depth_map(i)
            return idxs[G != 0].ravel()
        return idxs

    def _to_list_with_zeropadded_images(self, images, inputs=None):
        # forward_pass.py:181-198
        if inputs is None:
            inputs = []
        H, W, C = images[0].image.shape
        p = self._generation_params.padding
        for im in images:
            zeropadded = np.zeros((H + 2 * p, W + 2 * p, C), dtype=np.float32)
            zeropadded[p:p + H, p:p + W, :] = im.image
            inputs.append(zeropadded)
        return inputs

    def _features(self, scene, ref_idx, images):
        """(N, Hf, Wf, F) float32 CUDA tensor for [reference, neighbours...]."""
        if hasattr(self._model, "view_features"):     # precomputed per-view maps
            views = scene.view_indices_with_neighbors(ref_idx, self._generation_params.neighbors)
            return torch.stack([self._model.view_features(scene, v).to("cuda", torch.float32)
                                for v in views]).contiguous()
        f = self._model.predict(np.stack(self._to_list_with_zeropadded_images(images), axis=0))
        if not isinstance(f, torch.Tensor):
            f = torch.from_numpy(np.ascontiguousarray(f, dtype=np.float32))
        return f.to("cuda", torch.float32).contiguous()

    def _camera_arrays(self, images):
        P = np.ascontiguousarray(np.array([im.camera.P for im in images], dtype=np.float32))
        P_inv = np.ascontiguousarray(images[0].camera.P_pinv, dtype=np.float32)
        center = np.ascontiguousarray(images[0].camera.center, dtype=np.float32).ravel()
        return P, P_inv, center

    def forward_pass(self, scene, images_range):
        raise NotImplementedError()


class MultiViewCNNForwardPass(ForwardPass):
    """forward_pass.py:226-344 (kernel K10)."""

    def forward_pass(self, scene, images_range):
        assert isinstance(images_range, tuple)
        start, end, skip = images_range
        gp = self._generation_params
        H, W = scene.image_shape
        B = self.rays_batch
        ref_idx = start
        while ref_idx < end:
            ray_idxs = self.get_valid_rays_per_image(scene, ref_idx)
            images = scene.get_image_with_neighbors(ref_idx, gp.neighbors)
            features = self._features(scene, ref_idx, images)
            F = features.shape[-1]
            if self._fp is None:
                self._fp = perform_multi_view_cnn_forward_pass_with_depth_estimation(
                    gp.depth_planes, gp.neighbors + 1, F, H, W, gp.padding, scene.bbox.ravel(),
                    self._sampling_scheme)
            ctx = self._fp.context
            P, P_inv, center = (ctx.dev(a) for a in self._camera_arrays(images))
            # The reference uploads the ray list and allocates zero-filled S / points per image
            # (forward_pass.py:283-300).  K10 writes every entry of the rows it is handed, so the
            # scratch buffers live with the driver (no 66 MB memset per image at 130,000 rays x 32
            # planes), and the full ray list of an unfiltered image is uploaded once.
            nb = max(1, min(B, len(ray_idxs)) if B else len(ray_idxs))
            buf = getattr(self, "_k10_buffers", None)
            if buf is None or buf["key"] != (nb, gp.depth_planes, H * W, str(ctx.device)):
                buf = dict(key=(nb, gp.depth_planes, H * W, str(ctx.device)),
                           s=torch.empty((nb, gp.depth_planes), dtype=torch.float32, device=ctx.device),
                           pts=torch.empty((nb, gp.depth_planes, 4), dtype=torch.float32, device=ctx.device),
                           all_rays=None)
                self._k10_buffers = buf
            if self._filter_out_rays:
                ridx = ctx.dev(ray_idxs.astype(np.int32))
            else:
                if buf["all_rays"] is None:
                    buf["all_rays"] = ctx.dev(ray_idxs.astype(np.int32))
                ridx = buf["all_rays"]
            s, pts = buf["s"], buf["pts"]
            depth_map = torch.zeros((H * W,), dtype=torch.float32, device=ctx.device)
            for i in range(0, len(ridx), nb):
                self._fp(ridx[i:i + nb], features, P, P_inv, center, s, pts, depth_map[i:i + nb])
            ref_idx += skip
            yield depth_map.cpu().numpy().reshape(W, H).T


class MultiViewCNNVoxelSpaceForwardPass(ForwardPass):
    """forward_pass.py:347-485 (kernel K12)."""

    def forward_pass(self, scene, images_range):
        assert isinstance(images_range, tuple)
        start, end, skip = images_range
        gp = self._generation_params
        H, W = scene.image_shape
        M, B = gp.max_number_of_marched_voxels, self.rays_batch
        vg = None
        ref_idx = start
        while ref_idx < end:
            ray_idxs = self.get_valid_rays_per_image(scene, ref_idx)
            images = scene.get_image_with_neighbors(ref_idx, gp.neighbors)
            features = self._features(scene, ref_idx, images)
            F = features.shape[-1]
            if self._fp is None:
                grid_shape = np.array(scene.voxel_grid(gp.grid_shape).shape[1:])
                self._fp = batch_mvcnn_voxel_traversal_with_ray_marching_with_depth_estimation(
                    M, gp.depth_planes, gp.neighbors + 1, F, H, W, gp.padding,
                    scene.bbox.ravel(), grid_shape, self._sampling_scheme)
            ctx = self._fp.context
            if vg is None:
                vg = ctx.dev(np.ascontiguousarray(
                    scene.voxel_grid(gp.grid_shape).transpose(1, 2, 3, 0)))
            P, P_inv, center = (ctx.dev(a) for a in self._camera_arrays(images))
            # (as MultiViewCNNForwardPass: K12 writes every count, and reads a list and a column only
            # up to the ray's count -- the scratch rows need no 1 GB memset per batch of 130,000
            # rays x 650 voxels, and live with the driver)
            nb = max(1, min(B, len(ray_idxs)) if B else len(ray_idxs))
            buf = getattr(self, "_k12_buffers", None)
            if buf is None or buf["key"] != (nb, M, H * W, str(ctx.device)):
                buf = dict(key=(nb, M, H * W, str(ctx.device)),
                           s=torch.zeros((nb, M), dtype=torch.float32, device=ctx.device),
                           rvi=torch.zeros((nb, M, 3), dtype=torch.int32, device=ctx.device),
                           rvc=torch.zeros((nb,), dtype=torch.int32, device=ctx.device), all_rays=None)
                self._k12_buffers = buf
            if self._filter_out_rays:
                ridx = ctx.dev(ray_idxs.astype(np.int32))
            else:
                if buf["all_rays"] is None:
                    buf["all_rays"] = ctx.dev(ray_idxs.astype(np.int32))
                ridx = buf["all_rays"]
            s, rvi, rvc = buf["s"], buf["rvi"], buf["rvc"]
            depth_map = torch.zeros((H * W,), dtype=torch.float32, device=ctx.device)
            for i in range(0, len(ridx), nb):
                k = len(ridx[i:i + nb])
                self._fp(ridx[i:i + nb], features, P, P_inv, center, vg, rvi[:k], rvc[:k], s[:k],
                         depth_map[i:i + nb])
            ref_idx += skip
            yield depth_map.cpu().numpy().reshape(W, H).T


class _Messages(dict):
    """Per-image message rows of the resident path.  The kernels neither read nor write a
    row beyond its ray's voxel count, so the buffers are not zero-filled per pass (2.4 GB of
    memset at config-2 size); whoever LOOKS at an image's messages gets the reference's
    zero-initialised view -- the tail (and the rows of rays that send nothing, count <= 1)
    is cleared on first access."""

    def __init__(self):
        super(_Messages, self).__init__()
        self._tail = {}
        self.width = None      # the caller's M when the rows are stored with a padded stride

    def put(self, r, msgs, counts):
        dict.__setitem__(self, r, msgs)
        self._tail[r] = counts

    def clear(self):
        dict.clear(self)
        self._tail.clear()

    def __getitem__(self, r):
        m = dict.__getitem__(self, r)
        counts = self._tail.pop(r, None)
        if counts is not None and m.numel():
            live = torch.where(counts > 1, counts, torch.zeros_like(counts))
            m.masked_fill_(torch.arange(m.shape[1], device=m.device)[None, :] >= live[:, None], 0.0)
        if self.width is not None and m.shape[1] != self.width:
            return m[:, :self.width]      # [rows, M] as the reference has it (forward_pass.py:560-566)
        return m


class RayNetForwardPass(ForwardPass):
    """forward_pass.py:488-748."""

    def __init__(self, model, generation_params, sampling_scheme, image_shape, rays_batch,
                 filter_out_rays=False, bp_iterations=3, schedule="resident",
                 reference_quirks=False, deterministic=None, options=None):
        super(RayNetForwardPass, self).__init__(model, generation_params, sampling_scheme,
                                                image_shape, rays_batch, filter_out_rays)
        assert schedule in ("resident", "reference")
        self.bp_iterations = bp_iterations      # the reference hard-codes 3 (forward_pass.py:590)
        self.schedule = schedule
        self.reference_quirks = reference_quirks
        # every schedule / A-B knob lives in ONE PathOptions (hip_implementations/options.py);
        # the environment only overrides its defaults, read there.  `deterministic=True`
        # (messages summed as 64-bit fixed-point integers in the scatter, the accumulator and the
        # all-reduce: the same bits from run to run and for any number of GPUs)
        # is the one option with a constructor argument of its own.
        self.options = options if options is not None else PathOptions.from_env()
        if deterministic is not None:
            self.options = self.options.replace(deterministic=bool(deterministic))
        self._plan = None          # see _build_plan
        self.shard_balance = None  # per image: traversed voxels of every rank's shard (world > 1)
        self.shard_alpha = None    # ... and the per-ray constant the cuts weighed rays with
        self._side_stream = self._copy_stream = None
        self._pass_complete = True
        self._quick = None         # the last call's identity, see _forward_pass_resident
        self.trace = None          # a list: eager passes bracket their exchanges with events (_mark)
        self.captured = False      # whether the last pass was a graph replay
        self.ref_idx = -1
        self._ctx = None
        self._de = None
        self.timings = {}
        # state kept for inspection by tests / tools
        self._acc_flat = self._acc_grid = None
        self._acc_bias = 0.0
        self.messages = _Messages()   # per image: [rows, M] messages of this rank's rays
        self.voxel_count = {}
        self.ray_index = {}      # per image: ray index (pixel x*H + y) of every row
        self._ray_lists = {}     # (shape, patch, direction, device) -> the scene-wide ray list (_plan_lists)

    # the options tests and tools flip on an existing object
    ray_tile = property(lambda self: self.options.ray_tile,
                        lambda self, v: setattr(self, "options", self.options.replace(ray_tile=v)))
    deterministic = property(lambda self: self.options.deterministic,
                             lambda self, v: setattr(self, "options", self.options.replace(
                                 deterministic=None if v is None else bool(v))))
    @property
    def accumulator(self):
        """[gx][gy][gz] log-odds accumulator of the last pass (mrf_bp.cu:3-10 layout).  The
        resident pass keeps it bricked -- and, on the plan path, as the sum of the messages
        without the prior; the regrid (and `prior + sum`) happens when somebody asks."""
        if self._acc_grid is None and self._acc_flat is not None:
            grid = self._ctx.acc_to_grid(self._acc_flat)
            if self._acc_bias != 0.0:
                grid = grid.add_(self._acc_bias)      # prior + sum, the value a sweep reads
            self._acc_grid = grid
        return self._acc_grid

    @accumulator.setter
    def accumulator(self, value):
        self._acc_grid, self._acc_flat, self._acc_bias = value, None, 0.0

    # -- helpers -----------------------------------------------------------
    def _rows_M(self):
        """Row length of the resident buffers (see _row_stride)."""
        return getattr(self, "_M_rows", None) or self._generation_params.max_number_of_marched_voxels

    def _row_stride(self, M, grid_shape):
        """Row length of the resident [rows][M] buffers.  They are this driver's own (the reference
        keeps its messages in a memmap of the same shape, forward_pass.py:560-566, and everything
        else in per-batch arrays), so a row may be LONGER than M as long as no ray's list can be:
        a DDA moves monotonically along every axis, a list holds at most gx + gy + gz - 2 voxels, and
        with M at or above that the `count > M` truncation never fires -- the lists, columns,
        messages and maps are what they are with rows of exactly M.  The kernels take whole
        16-byte / 16-step pieces of a row when its length is a multiple of 16 (the traversal's
        flush, the scatters' loads): the reference's own default M = 650 (scripts/arguments.py:221)
        becomes 656 here (k_traverse 2.9 -> 1.5 ms per step at its CLI defaults).  The literal
        K1 / K2 schedule and truncating shapes keep M."""
        if self.schedule != "resident" or self.reference_quirks or M % 16 == 0:
            return int(M)
        if int(M) < int(np.sum(grid_shape)) - 2:
            return int(M)
        return (int(M) + 15) // 16 * 16

    def _context(self, scene, F):
        if self._ctx is None:
            gp = self._generation_params
            H, W = scene.image_shape
            grid_shape = np.array(scene.voxel_grid(gp.grid_shape).shape[1:])
            self._M_rows = self._row_stride(gp.max_number_of_marched_voxels, grid_shape)
            self.messages.width = gp.max_number_of_marched_voxels
            self._fp, self._de = perform_raynet_fp(
                self._M_rows, gp.depth_planes, gp.neighbors + 1, F, H, W,
                gp.padding, scene.bbox.ravel(), grid_shape, self._sampling_scheme)
            self._ctx = self._fp.context
            self._vg = self._ctx.dev(np.ascontiguousarray(
                scene.voxel_grid(gp.grid_shape).transpose(1, 2, 3, 0)))   # forward_pass.py:573-575
            self._ctx.set_voxel_grid(self._vg)
            self._ctx._grid_src = self._vg
        if hasattr(self._ctx, "set_options"):
            self._ctx.set_options(self.options)     # contexts are shared per configuration
        return self._ctx

    def _view_features(self, scene, ref_idxs):
        """Per-view feature maps, each computed once: {view index: [Hf, Wf, F] tensor}."""
        gp = self._generation_params
        bank = {}
        if hasattr(self._model, "view_features"):
            for r in ref_idxs:
                for v in scene.view_indices_with_neighbors(r, gp.neighbors):
                    if v not in bank:
                        bank[v] = self._model.view_features(scene, v)
            return bank
        for r in ref_idxs:
            views = scene.view_indices_with_neighbors(r, gp.neighbors)
            todo = [v for v in views if v not in bank]
            if todo:
                f = self._features(scene, r, [scene.get_image(v) for v in todo])
                for k, v in enumerate(todo):
                    bank[v] = f[k].contiguous()
        return bank

    def _prior(self):
        gamma = self._generation_params.gamma_mrf
        return float(np.float32(np.log(gamma) - np.log(1 - gamma)))

    # -- the generator -------------------------------------------------------
    def forward_pass(self, scene, images_range):
        assert isinstance(images_range, tuple)
        if self.schedule == "reference":
            return self._forward_pass_reference(scene, images_range)
        return self._forward_pass_resident(scene, images_range)

    # -- the plan of a resident pass -------------------------------------------
    # Everything of a pass that depends on the scene's cameras, the image range, the options and
    # the sharding only -- camera table, feature-pointer table, ray lists, the ranks' shard
    # bounds, the HBM buffers, the C plan, the host buffers the maps land in -- is built once
    # and reused while those stay the same (0.2 ms of host work per pass otherwise; 10 % of a
    # rank's step at 8 GPUs).  Three steps: lists, cuts, buffers.
    def _plan_lists(self, scene, refs, views_of, ctx):
        """The images' ray lists in ROW order (what row i of an image's buffers holds)."""
        H, W = scene.image_shape
        opt, dev = self.options, ctx.device
        patch_rows = opt.ray_tile is not None
        # patches are enumerated along the direction of the neighbour views' epipolar lines (of
        # the first reference image: one list serves the whole scene)
        epipolar_rows = bool(refs) and sweep_direction(
            H, W, [scene.get_image(v) for v in views_of[refs[0]]]) == "rows"
        along_rows = patch_rows and epipolar_rows
        lists, shared = {}, None
        for r in refs:
            if self._filter_out_rays:
                rays = ctx.dev(np.ascontiguousarray(
                    self.get_valid_rays_per_image(scene, r).astype(np.int32)))
                if patch_rows:
                    rays = tile_order(rays, H, W, *opt.ray_tile, along_rows=along_rows)
            else:       # all H*W rays (forward_pass.py:166-168): built on the device, once
                if shared is None:
                    # (the list depends on the image shape, the patch and the direction only -- not on
                    # the cameras: a caller looping over scenes, one pass each as the reference's
                    # script does, gets it back instead of a sort per scene; read-only from here on)
                    lkey = (H, W, opt.ray_tile, along_rows, str(dev))
                    shared = self._ray_lists.get(lkey)
                    if shared is None:
                        shared = torch.arange(H * W, dtype=torch.int32, device=dev)
                        if patch_rows:
                            shared = tile_order(shared, H, W, *opt.ray_tile, along_rows=along_rows)
                        if len(self._ray_lists) >= 4:
                            self._ray_lists.clear()
                        self._ray_lists[lkey] = shared
                rays = shared
            lists[r] = rays
        self._along_rows = along_rows
        return lists, shared

    def _plan_cuts(self, ctx, dist, refs, lists, shared, cam_dev, world):
        """bounds[k][q] .. bounds[k][q+1] = rows of image k owned by rank q, and (world > 1,
        work-balanced shards) every rank's traversed voxels per image."""
        gp, opt = self._generation_params, self.options
        V = len(refs)
        bounds = [[shard_bounds(len(lists[r]), q, world)[0] for q in range(world)] +
                  [len(lists[r])] for r in refs]
        if not (world > 1 and V > 0 and opt.shard == "voxels" and hasattr(ctx, "count_voxels")):
            return bounds, None
        # equal RAY counts leave the ranks unequal work: the border strips of an image miss
        # most of the box (config 2: the strips' voxel totals spread 0.1 .. 1.6 x the mean).
        # The traversal's counts (bit-exact integers, the same on every rank: no exchange)
        # weigh every ray as `count + alpha * mean count`, alpha = the plane sweep's per-ray
        # cost in units of the mean ray's per-voxel work (options.shard_alpha_for: from the
        # shape, or PathOptions.shard_alpha); cuts fall on whole 256-row scatter tiles.
        # Images that share one ray list get the SAME cuts (their traversal / sweep is one
        # launch over all of them, and so is every later kernel: only a rank's total
        # counts), from the weights summed over the images.
        N, D = gp.neighbors + 1, gp.depth_planes

        def cut(c, images):
            """c: int64 [n] voxel counts per row (summed over `images` images) -> world + 1
            row bounds of equal weight"""
            n_k = int(c.numel())
            if n_k == 0:
                return [0] * (world + 1)
            total = int(c.sum().item())
            alpha = opt.shard_alpha if opt.shard_alpha is not None else \
                shard_alpha_for(N, D, total / float(n_k * images))
            self.shard_alpha = alpha
            # integer weights: their prefix sums are exact whatever the scan order, so
            # every rank computes the same cuts from the same (bit-exact) counts
            extra = int(round(16 * alpha * total / n_k))
            w = torch.cumsum(c * 16 + extra, 0)
            total_w = int(w[-1].item())
            targets = torch.tensor([total_w * q // world for q in range(1, world)],
                                   dtype=torch.int64, device=w.device)
            b = torch.searchsorted(w, targets).cpu().tolist()
            # whole 256-row scatter tiles where the image is large enough for that to
            # leave the balance intact, finer otherwise
            align = 256
            while align > 1 and align * 4 * world > n_k:
                align //= 2
            cuts = [0]
            for v in b:
                cuts.append(max(cuts[-1], min(n_k, (int(v) + align // 2) // align * align)))
            return cuts + [n_k]

        def shares(c, cuts):
            cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=c.device),
                            torch.cumsum(c, 0)]).cpu()
            return [int(cs[cuts[q + 1]] - cs[cuts[q]]) for q in range(world)]

        if shared is not None:
            counts = ctx.count_voxels(shared, cam_dev).to(torch.int64)      # [V, n]
            cuts = cut(counts.sum(0), V)
            bounds = [cuts for _ in refs]
            balance = [shares(counts[k], cuts) for k in range(V)]
        else:
            bounds, balance = [], []
            for k, r in enumerate(refs):
                c = ctx.count_voxels(lists[r], cam_dev[k:k + 1])[0].to(torch.int64)
                bounds.append(cut(c, 1))
                balance.append(shares(c, bounds[-1]))
        if dist is not None:         # belt and braces: the ranks must agree on the cuts
            t = torch.tensor(bounds, dtype=torch.int64, device=ctx.device)
            lo_t, hi_t = t.clone(), t.clone()
            dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_t, op=dist.ReduceOp.MAX)
            assert torch.equal(lo_t, hi_t), "ranks disagree on the shard bounds"
        return bounds, balance

    def _plan_buffers(self, ctx, plan, refs, old_bytes):
        """HBM: messages and counts of ALL images stay resident (they carry state across the
        iterations); the voxel lists and columns of as many images as fit -- the rest are
        recomputed group by group in every sweep, like the reference recomputes everything."""
        gp, opt, dev = self._generation_params, self.options, ctx.device
        M, V, npad = self._rows_M(), len(refs), plan["npad"]
        G = ctx.acc_size()
        per_image = npad * M * 4
        budget = float(opt.resident_gb) * 2 ** 30
        if budget <= 0 and dev.type == "cuda":
            free, _ = torch.cuda.mem_get_info(dev)
            # (everything resident: what this plan takes at most.  When the driver's own free memory
            # holds it twice over there is nothing to decide, and the allocator's statistics --
            # 0.3 ms of host time per new scene -- are not asked for)
            if (V * per_image + 4 * G * 8 + V * npad * 48 + 2 * V * per_image) * 2 > free + old_bytes:
                # what the caching allocator holds but nobody uses is ours to take as well, and so
                # are the outgoing plan's buffers (released below, before anything is allocated)
                free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
            free += old_bytes
            budget = 0.9 * free
        fixed = V * per_image + 4 * G * 8 + V * npad * 48
        if budget > 0 and fixed + 2 * per_image > budget:
            raise MemoryError(
                "resident schedule: the messages of %d reference images (%.1f GB) do not leave "
                "room for one image's columns in %.1f GB of HBM; run fewer images per call"
                % (V, V * per_image / 2 ** 30, budget / 2 ** 30))
        Vg = V if budget <= 0 else int(max(1, min(V, (budget - fixed) // (2 * per_image))))
        if Vg < V:
            import warnings
            warnings.warn("resident schedule: the columns of %d of %d reference images fit the HBM "
                          "budget (%.1f GB); traversal and plane sweep are recomputed group by "
                          "group in every sweep" % (Vg, V, budget / 2 ** 30))
        rows_g = Vg * npad
        fixed_pt = opt.fixed_point(plan["world"]) and hasattr(ctx, "scene_bp_sweep_fixed")
        prior = plan["prior"]
        plan.update(
            groups=[list(range(g, min(g + Vg, V))) for g in range(0, V, Vg)] if V else [], Vg=Vg,
            fixed=fixed_pt, bytes=fixed + 2 * rows_g * M * 4,
            vox=torch.empty((rows_g, M), dtype=torch.int32, device=dev),
            Sr=torch.empty((rows_g, M), dtype=torch.float32, device=dev),
            msgs=torch.empty((V * npad, M), dtype=torch.float32, device=dev),   # see _Messages
            rvc=torch.zeros((V * npad,), dtype=torch.int32, device=dev),   # padding rays: count 0
            depth=torch.zeros((V * npad,), dtype=torch.float32, device=dev),
            # granular path: iteration 0 reads the prior from a buffer no sweep ever writes
            acc_prior=torch.full((G,), prior, dtype=torch.float32, device=dev),
            acc_a=torch.empty((G,), dtype=torch.float32, device=dev),
            acc_b=torch.empty((G,), dtype=torch.float32, device=dev),
            acc_part=(torch.zeros((G,), dtype=torch.int64, device=dev) if fixed_pt else
                      torch.zeros((ctx.acc_copies(), G), dtype=torch.float32, device=dev)))

    def _build_plan(self, scene, refs, bank, ctx, dist, rank, world):
        gp, opt = self._generation_params, self.options
        M, N = self._rows_M(), gp.neighbors + 1
        H, W = scene.image_shape
        dev = ctx.device
        V = len(refs)
        views_of = {r: tuple(scene.view_indices_with_neighbors(r, gp.neighbors)) for r in refs}
        ptrs = tuple(tuple(bank[v].data_ptr() for v in views_of[r]) for r in refs)
        cams = tuple(scene.get_image(v).camera for r in refs for v in views_of[r])
        # geometry: the cameras (objects whose P / P_pinv / center are computed once and cached,
        # common/camera.py -- a changed camera is a new object; the plan holds them, so an id is
        # never recycled), the neighbour selection, the image range, the shapes, the options and
        # the sharding.  The feature maps' ADDRESSES are not geometry: when only they moved
        # (the allocator placed recomputed maps elsewhere) the pointer table is refreshed.
        key = (id(scene), tuple(id(c) for c in cams), tuple(sorted(views_of.items())), tuple(refs),
               H, W, M, N, gp.depth_planes, world, rank, dist is not None, opt.key(),
               self.rays_batch, str(dev), self._prior())
        plan = self._plan
        if plan is not None and plan["key"] == key and not self._filter_out_rays:
            if plan["ptrs"] != ptrs:
                plan["table"].copy_(torch.tensor(ptrs, dtype=torch.int64))
                plan["ptrs"] = ptrs
            return plan
        stride = 12 * N + 12 + 4
        cam_host = np.zeros((V, stride), dtype=np.float32)
        for k, r in enumerate(refs):
            P, P_inv, center = self._camera_arrays([scene.get_image(v) for v in views_of[r]])
            cam_host[k, :12 * N] = P.ravel()
            cam_host[k, 12 * N:12 * N + 12] = P_inv.ravel()
            cam_host[k, 12 * N + 12:] = center
        # all camera matrices go up in ONE copy: a pageable host->device copy is a stream
        # synchronisation point, one per image would drain the GPU between images
        cam_dev = ctx.dev(cam_host)
        if hasattr(ctx, "scatter_reset"):
            ctx.scatter_reset()          # new cameras: the scatter re-learns its tile shape
        patch_rows = opt.ray_tile is not None
        lists, shared = self._plan_lists(scene, refs, views_of, ctx)
        bounds, balance = self._plan_cuts(ctx, dist, refs, lists, shared, cam_dev, world)
        shards = []
        for k, r in enumerate(refs):
            lo, hi = bounds[k][rank], bounds[k][rank + 1]
            shards.append((lists[r][lo:hi], lo, hi, len(lists[r])))
        # rows per image: the largest shard ANY rank holds (the same number on every rank, the
        # depth maps are exchanged with an all-gather), rounded to whole scatter tiles
        npad = max([bounds[k][q + 1] - bounds[k][q] for k in range(V) for q in range(world)] + [1])
        npad = (npad + 255) // 256 * 256            # scatter tiles never straddle two images

        old_bytes = plan["bytes"] if plan is not None else 0
        if self._side_stream is not None:           # nobody reads the outgoing buffers any more
            torch.cuda.current_stream(dev).wait_stream(self._side_stream)
            torch.cuda.current_stream(dev).wait_stream(self._copy_stream)
        # release the old buffers first -- ALL references this driver holds to them: the plan, the
        # record of the last call (it holds the plan), and the last pass's outputs that are views
        # of the plan's buffers (messages, counts, the accumulator).  A new scene invalidates them
        # exactly as the next pass over the same scene would have rewritten them; whatever a
        # caller still holds stays alive through the caller's own reference.  The allocator then
        # hands the very blocks to the new plan: no second 7 GB set, no fresh allocation.
        self._plan = plan = None
        self._quick = None
        self.messages.clear()
        self.voxel_count.clear()
        self._acc_flat = self._acc_grid = None
        if hasattr(ctx, "bind_slab_boxes"):
            ctx.bind_slab_boxes(None)                # (the binding holds the old list buffer)
        if getattr(ctx, "_scatter_items", None) is not None:
            ctx.bind_scatter_items(None)             # (... and so does the scatter's work list)
        plan = dict(key=key, ptrs=ptrs, scene=scene, cams=cams, prior=self._prior(), dirty=False,
                    cam_dev=cam_dev, views_of=views_of, lists=lists, bounds=bounds,
                    balance=balance, shards=shards, npad=npad, shared=shared,
                    patch_rows=patch_rows, fast=None, direct=False, world=world,
                    along_rows=self._along_rows,
                    table=torch.tensor(ptrs, dtype=torch.int64).to(dev) if V else None)
        self._plan_buffers(ctx, plan, refs, old_bytes)
        # static views of every image's rows in the scene-wide buffers
        per_image = {}
        for k, r in enumerate(refs):
            ridx, lo, hi, total = shards[k]
            n, row0 = len(ridx), k * npad
            per_image[r] = dict(k=k, ridx=ridx, n=n, lo=lo, hi=hi, total=total, row0=row0,
                                center=cam_dev[k, 12 * N + 12:],
                                rvc=plan["rvc"][row0:row0 + n], msgs=plan["msgs"][row0:row0 + n],
                                depth=plan["depth"][row0:row0 + n])
        plan["per_image"] = per_image
        plan["slab_table"] = None
        if hasattr(ctx, "bind_slab_boxes") and plan["vox"].numel() >= M and opt.slab_boxes:
            # the scatters merge boxes the traversal left
            plan["slab_table"] = ctx.bind_slab_boxes(plan["vox"])
        # the plan path (one C call per phase, no combine kernel, maps into plan-owned host
        # buffers): every image casts the same ray list in one launch, all columns resident
        fast_ok = (opt.plan_path and hasattr(ctx, "scene_run") and shared is not None and V > 0 and
                   not self.rays_batch and len(plan["groups"]) == 1 and not self.reference_quirks)
        if dist is not None and not (dev.type == "cuda" and (H * W) % 4 == 0 and
                                     hasattr(ctx, "stitch_rows")):
            fast_ok = False     # the owners' maps are put together by rn_stitch_rows (float4 stores)
        if dist is not None and world > 1 and hasattr(ctx, "scene_run"):
            # the two paths exchange differently (per-image all-gathers / one): every rank must
            # take the same one, and a rank's HBM budget may have decided otherwise
            t = torch.tensor([1 if fast_ok else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            fast_ok = bool(int(t.item()))
        if fast_ok:
            # no process group: the depth sweeps write the maps in pixel order themselves
            # (rn_scene_plan.depth_image) -- no reordering pass between sweep and copy.  (With
            # a group the rows are exchanged first, _epilogue_buffers.)
            plan["direct"] = dist is None
            if plan["direct"]:
                plan["maps_dev"] = torch.zeros((V, H * W), dtype=torch.float32, device=dev)
            plan["fast"] = ctx.scene_plan(
                V, npad, shards[0][0], plan["table"], cam_dev, plan["vox"], plan["rvc"],
                plan["Sr"], plan["msgs"], plan["acc_a"], plan["acc_b"], plan["depth"],
                plan["prior"], patch_rows, acc_fixed=plan["acc_part"] if plan["fixed"] else None,
                **({"depth_image": plan["maps_dev"]} if plan["direct"] else {}))
        if not self._filter_out_rays:
            self._plan = plan
        return plan

    # -- where the maps land -------------------------------------------------------------------
    # -- where the maps land -------------------------------------------------------------------
    # A pass's maps are written by the GPU straight into pinned host memory (the stitch kernel /
    # the copies of the epilogue).  What the caller gets (PathOptions.maps):
    #   "lease" (default)  arrays that ARE that pinned memory, each holding a lease on it: the
    #            memory goes back to the plan's pool when the array -- and every view or slice
    #            made of it -- has been garbage-collected (weakref.finalize on the array's buffer
    #            owner; no reference counts are looked at).  Until then no later pass touches it:
    #            a pass takes a set of maps nobody holds a lease on, or a new one.  The reference's
    #            semantics -- fresh arrays from `.get()`, forward_pass.py:739-744 -- without a
    #            copy; a caller that drops a pass's maps before the next pass (bench.py, the
    #            script) stays on one set, one that keeps the last map alternates between two.
    #   "copy"   pageable copies out of ONE pinned scratch set (also what "lease" falls back to
    #            when a caller holds leases on MAX_LEASED_SETS sets: pinned memory is not for
    #            hoarding).
    MAX_LEASED_SETS = 4

    @staticmethod
    def _new_set(V, HW, cuda):
        # `live`: one token per array out there.  set.add / set.discard are single operations under
        # the GIL, so a finaliser running in the middle of a pass (or in another thread) cannot
        # lose an update the way a counter's read-modify-write could
        return dict(host=torch.empty((V, HW), dtype=torch.float32, pin_memory=cuda), live=set(),
                    tokens=itertools.count())

    def _take_set(self, plan, V, HW, cuda):
        """-> (key, set, leased): a set of pinned maps this pass may write -- one nobody holds a
        lease on (key = its index in the pool), a new one, or the scratch set (key -1, copies)."""
        if self.options.maps == "lease":
            pool = plan["sets"]
            for i, st in enumerate(pool):
                if not st["live"]:
                    return i, st, True
            if len(pool) < self.MAX_LEASED_SETS:
                pool.append(self._new_set(V, HW, cuda))
                return len(pool) - 1, pool[-1], True
        if plan["scratch"] is None:
            plan["scratch"] = self._new_set(V, HW, cuda)
        return -1, plan["scratch"], False

    @staticmethod
    def _lease(st, k):
        """Image k of the set as an ndarray that owns a lease on the set's memory."""
        row = st["host"][k]
        owner = (ctypes.c_float * row.numel()).from_address(row.data_ptr())
        token = next(st["tokens"])
        st["live"].add(token)
        # (the finaliser's arguments keep the set -- and its pinned tensor -- alive until then)
        weakref.finalize(owner, lambda st, token: st["live"].discard(token), st, token)
        return np.frombuffer(owner, dtype=np.float32)

    def _epilogue_buffers(self, plan, refs, H, W, dev, world, rank, collective):
        """Plan-owned output side: per-image events, the device-side pixel-order maps, the pool of
        pinned host maps (see _take_set) -- a pass allocates nothing once the pool has the sets
        its caller's habits need."""
        if "sets" in plan:
            return
        cuda = dev.type == "cuda"
        V, HW = len(refs), H * W
        plan["sets"], plan["scratch"] = [], None
        plan["graphs"] = {}
        plan["passes"] = 0
        if "maps_dev" not in plan:
            plan["maps_dev"] = torch.zeros((V, HW), dtype=torch.float32, device=dev)
        plan["wait_ev"] = [None] * V
        if cuda:
            plan["ev_ready"] = [torch.cuda.Event() for _ in range(V)]
            plan["ev_done"] = [torch.cuda.Event() for _ in range(V)]
            plan["ev_all"] = torch.cuda.Event()
            if self._side_stream is None:
                self._side_stream, self._copy_stream = _side_streams(dev)
        npad, lists, bounds = plan["npad"], plan["lists"], plan["bounds"]
        plan["owners"] = [map_owner(k, V, world) for k in range(V)] if collective else None
        if not collective:
            return
        # With a process group (the plan path then requires rn_stitch_rows, _build_plan): image
        # k's map is assembled on ONE rank.  ONE depth launch over all of a rank's rows, ONE
        # all-gather that hands every rank everybody's rows (5 MB more per rank and step than an
        # all-to-all of only what an owner needs, on links that are idle at that point -- and
        # the one of the two that RCCL captures into a HIP graph on this runtime), and on an
        # owner ONE rn_stitch_rows launch that puts its images into pixel order and writes the
        # pinned host maps across PCIe itself.
        owners = plan["owners"]
        mine = [k for k in range(V) if owners[k] == rank]
        blk = V * npad                               # one source rank's block in `recv`
        recv = torch.zeros((world * blk + 1,), dtype=torch.float32, device=dev)
        table = None
        if mine:
            idx = torch.empty((len(mine), HW), dtype=torch.int32, device=dev)
            for j, k in enumerate(mine):
                # per pixel of image k: where its ray's depth sits in `recv` (the zero behind
                # the blocks for a pixel without a ray)
                at = torch.full((HW,), world * blk, dtype=torch.int64, device=dev)
                rays, cuts = lists[refs[k]].long(), bounds[k]
                for q in range(world):
                    lo_q, hi_q = cuts[q], cuts[q + 1]
                    at[rays[lo_q:hi_q]] = q * blk + k * npad + torch.arange(
                        hi_q - lo_q, dtype=torch.int64, device=dev)
                idx[j] = at.to(torch.int32)
            table = idx.reshape(-1)
        plan["rows"] = dict(mine=mine, recv=recv, table=table)

    def _mark(self, name, begin):
        """bench.py --gpus N: a pair of events around every exchange of an eager pass
        (self.trace = []); nothing otherwise."""
        if self.trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.trace.append((name, begin, ev))

    def _emit_direct(self, plan, groups, host_set):
        """The maps of the image groups [a, b) -- written in pixel order by their depth launches,
        ev_ready[a] recorded behind each -- go to the pinned host maps, one copy per group."""
        copy = self._copy_stream
        with torch.cuda.stream(copy):
            for a, b in groups:
                copy.wait_event(plan["ev_ready"][a])
                host_set[a:b].copy_(plan["maps_dev"][a:b], non_blocking=True)
                plan["ev_done"][a].record()
                for k in range(a, b):
                    plan["wait_ev"][k] = plan["ev_done"][a]

    def _exchange(self, plan, ctx, dist, world, it):
        """The ranks' partial sums of BP iteration `it` become everybody's accumulator: ONE
        all-reduce (float sums, or 64-bit fixed-point integers in the deterministic mode) --
        or, deterministic mode with exchange="reduce_scatter", an int64 reduce-scatter, the
        fixed -> float combine on the rank's own slab, and an all-gather of FLOATS (3/4 of the
        all-reduce's bytes, the combine sharded N ways)."""
        fast = plan["fast"]
        if not plan["fixed"]:
            if dist is not None:
                self._mark("exchange", True)
                dist.all_reduce(plan["acc_b" if it & 1 else "acc_a"], op=dist.ReduceOp.SUM)
                self._mark("exchange", False)
            return
        part = plan["acc_part"]
        out = plan["acc_b" if it & 1 else "acc_a"]
        if dist is not None and self.options.exchange == "reduce_scatter" and \
                part.numel() % (64 * world) == 0 and hasattr(ctx, "acc_combine_fixed_range"):
            slab = part.numel() // world
            if "slab_i" not in plan:
                plan["slab_i"] = torch.empty((slab,), dtype=torch.int64, device=part.device)
                plan["slab_f"] = torch.empty((slab,), dtype=torch.float32, device=part.device)
            self._mark("exchange", True)
            dist.reduce_scatter_tensor(plan["slab_i"], part, op=dist.ReduceOp.SUM)
            part.zero_()
            ctx.acc_combine_fixed_range(plan["slab_i"], plan["prior"], plan["slab_f"])
            dist.all_gather_into_tensor(out, plan["slab_f"])
            self._mark("exchange", False)
            return
        if dist is not None:
            self._mark("exchange", True)
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
            self._mark("exchange", False)
        ctx.scene_run(fast, _lib.RN_RUN_COMBINE, it)

    def _run_plan_path(self, plan, ctx, refs, dist, world, host_set, captured=False):
        """One pass as phases of the C plan (include/raynet_hip.h, rn_scene_run): 1 + T calls
        for the K1 prefix and the T BP iterations, the exchange between them, then the depth
        sweep and the maps' way to the host.  Eager, or -- `captured` -- recorded into a HIP graph
        (then nothing here may be waited for by the host: the side streams join the capturing
        stream at the end and the graph's owner records ONE event behind each replay)."""
        fast, T = plan["fast"], self.bp_iterations
        fixed = plan["fixed"]
        if T == 0:
            plan["msgs"].zero_()      # the depth sweep reads the initial, zero messages
            ctx.scene_run(fast, _lib.RN_RUN_PREPARE)
            if fixed:
                plan["acc_b"].fill_(plan["prior"])
            else:
                plan["acc_b"].zero_()
        for it in range(T):
            ctx.scene_run(fast, (_lib.RN_RUN_PREPARE if it == 0 else 0) | _lib.RN_RUN_SWEEP, it)
            self._exchange(plan, ctx, dist, world, it)
        final = plan["acc_b" if (T - 1) & 1 else "acc_a"]
        self._acc_flat, self._acc_bias = final, (0.0 if fixed else plan["prior"])
        V = len(refs)
        if plan["direct"]:
            # one GPU, no process group: the depth launches write the maps in pixel order
            # themselves (rn_scene_plan.depth_image).  Every launch first (the GPU never waits for
            # the host between them), then the copies: all images but the last in ONE launch and
            # ONE copy, under the last image's launch
            groups = [(0, V - 1), (V - 1, V)] if V >= 3 else [(k, k + 1) for k in range(V)]
            for a, b in groups:
                if b - a > 1:
                    ctx.scene_run(fast, _lib.RN_RUN_DEPTH_RANGE, T, a | ((b - a) << 16))
                else:
                    ctx.scene_run(fast, _lib.RN_RUN_DEPTH, T, a)
                plan["ev_ready"][a].record()
            self._emit_direct(plan, groups, host_set)
        else:
            # sharded rays, owner-only maps (_epilogue_buffers): ONE depth launch over all of
            # this rank's rows, ONE all-gather of the ranks' rows, ONE stitch launch on an owner
            rows = plan["rows"]
            ctx.scene_run(fast, _lib.RN_RUN_DEPTH_RANGE, T, 0 | (V << 16))
            plan["ev_ready"][0].record()
            side = self._side_stream
            with torch.cuda.stream(side):
                side.wait_event(plan["ev_ready"][0])
                self._mark("gather", True)
                dist.all_gather_into_tensor(rows["recv"][:-1], plan["depth"])
                self._mark("gather", False)
                mine = rows["mine"]
                if mine:
                    HW = plan["maps_dev"].shape[1]
                    host = host_set.view(-1)
                    ctx.stitch_rows(rows["recv"], rows["table"],
                                    host[mine[0] * HW:(mine[-1] + 1) * HW])
                plan["ev_done"][0].record()
            for k in range(V):
                plan["wait_ev"][k] = plan["ev_done"][0]
        if captured:
            cur = torch.cuda.current_stream(ctx.device)
            cur.wait_stream(self._side_stream)
            cur.wait_stream(self._copy_stream)

    def _scatter_work_list(self, plan, ctx):
        """From a plan's second pass on (the first one's traversal left the voxel counts) the box
        scatter takes a work list -- a tile's live chunks in pieces, longest first -- instead of
        tiles x a fixed split: a shard of an eight-rank run has ~1500 tiles whose longest (12
        chunks) IS the launch otherwise.  The counts depend on cameras and shard only, so the
        list is built once per plan and tile level (one host synchronisation) and re-bound when
        another driver used the shared context in between."""
        if not hasattr(ctx, "bind_scatter_items"):
            return
        use = self.options.scatter_items and plan["patch_rows"] and plan["passes"] >= 1 and \
            len(plan["groups"]) == 1 and plan["vox"].shape[0] == plan["rvc"].shape[0]
        level = ctx.scatter_state()[0] if use else 2
        if not use or level > 1:
            if getattr(ctx, "_scatter_items", None) is not None:
                ctx.bind_scatter_items(None)
            return
        if plan.get("items_level") != level:
            import os
            plan["items"] = ctx.bind_scatter_items(
                plan["vox"], plan["rvc"], level,
                target_items=int(os.environ.get("RAYNET_SCATTER_TARGET", "2048")))    # (A/B knob; profiles/r04_exp_scatter_items.txt)
            plan["items_level"] = level
            plan["graphs"] = {}                  # (a captured step has the old launch shape in it)
        elif plan["items"] is not None and not ctx.scatter_items_bound(plan["items"]):
            ctx.bind_scatter_items(plan["vox"], level=level, items=plan["items"])

    def _capturable(self, plan, ctx, dist):
        """Whether this pass may be recorded into a HIP graph: a CUDA device, a transport whose
        collectives are stream work (RCCL; gloo runs on the host), the scatter's adaptive tile
        shape settled (its probe launches copy counters to the host, and the shape is baked into
        the graph), nobody bracketing launches with events."""
        mode = self.options.capture
        if mode == "off" or (mode == "auto" and dist is None) or ctx.device.type != "cuda" or \
                not hasattr(ctx, "scatter_settled"):
            return False
        if self.trace is not None or getattr(ctx, "prof_active", False):
            return False
        if dist is not None and not (getattr(dist, "capturable", False) or _backend_of(dist) == "nccl"):
            return False
        return plan["passes"] >= 2 and ctx.scatter_settled()

    # -- the resident schedule ------------------------------------------------------------------
    def _forward_pass_resident(self, scene, images_range):
        start, end, skip = images_range
        gp = self._generation_params
        H, W = scene.image_shape
        refs = list(range(start, end, skip))
        if not refs:            # an empty image range yields nothing (forward_pass.py:597-602)
            return
        dist, rank, world = _dist()
        # an initialised process group runs its collectives even when it has ONE rank (that is
        # how a single-GPU box exercises the RCCL path); no group, no collectives

        # The same call as last time -- same scene object, range, model, options, sharding, the
        # cameras and feature maps the very objects (at the very addresses) the plan was built
        # from: the plan as it is.  ~10 identity checks instead of the bank, the neighbour lists,
        # the pointer table and the plan key: 17 us of interpreter per pass (tools/pass_overhead.py:
        # 198 -> 181 us for a pass whose kernels are a few us each), on the critical path of a
        # rank that waits for its map between two replays of its graph.
        gp_now = (gp.max_number_of_marched_voxels, gp.neighbors, gp.depth_planes, gp.gamma_mrf,
                  gp.padding)
        qkey = (id(scene), start, end, skip, id(self._model), dist is not None, rank, world,
                self.options.key(), self.rays_batch, gp_now, self._filter_out_rays)
        q = self._quick
        plan = None
        if q is not None and q["key"] == qkey and q["plan"] is self._plan and self._ctx is not None:
            vf = self._model.view_features
            if all(scene.get_image(v).camera is c for v, c in zip(q["views"], q["cams"])) and \
                    all(vf(scene, v) is t and t.data_ptr() == a
                        for v, t, a in zip(q["views"], q["tensors"], q["ptrs"])):
                plan, ctx = q["plan"], self._ctx
                ctx.set_options(self.options)
                dev = ctx.device
        q = None        # (this frame must not keep the old plan alive while a new one is allocated)
        self._acc_flat = self._acc_grid = None
        self._acc_bias = 0.0
        if plan is None:
            bank = self._view_features(scene, refs)
            F = next(iter(bank.values())).shape[-1]
            ctx = self._context(scene, F)
            dev = ctx.device
            moved = False
            for v, f in bank.items():
                if f.device != dev or f.dtype != torch.float32 or not f.is_contiguous():
                    bank[v] = f.to(dev, torch.float32).contiguous()
                    moved = True
            plan = self._build_plan(scene, refs, bank, ctx, dist, rank, world)
            self._quick = None
            if plan["fast"] is not None and self._plan is plan and not moved and \
                    hasattr(self._model, "view_features"):
                views = sorted(bank)
                self._quick = dict(key=qkey, plan=plan, views=views,
                                   cams=[scene.get_image(v).camera for v in views],
                                   tensors=[bank[v] for v in views],
                                   ptrs=[bank[v].data_ptr() for v in views])
        self.shard_balance = plan["balance"]
        if plan["slab_table"] is not None and not ctx.slab_boxes_bound_to(plan["vox"]):
            # another driver object used the (shared) context in between
            ctx.bind_slab_boxes(plan["vox"], plan["slab_table"])
        elif plan["slab_table"] is None and hasattr(ctx, "bind_slab_boxes") and \
                getattr(ctx, "_slab_boxes", None) is not None:
            ctx.bind_slab_boxes(None)
        per_image = plan["per_image"]
        for r in refs:
            self.ray_index[r] = per_image[r]["ridx"]
        if self._side_stream is not None and not self._pass_complete:
            # an abandoned earlier pass may still be copying (a pass consumed to its last map has
            # waited for every one of its copies)
            torch.cuda.current_stream(dev).wait_stream(self._side_stream)
            torch.cuda.current_stream(dev).wait_stream(self._copy_stream)
        self._pass_complete = False

        if plan["fast"] is not None:
            self._epilogue_buffers(plan, refs, H, W, dev, world, rank, dist is not None)
            V = len(refs)
            self._scatter_work_list(plan, ctx)
            # the pinned maps this pass writes: a set nobody holds a lease on (see _take_set)
            set_key, mset, leased = self._take_set(plan, V, H * W, dev.type == "cuda")
            host_set = mset["host"]
            # (a pass whose launches or exchanges are bracketed by events runs eagerly)
            eager = self.trace is not None or getattr(ctx, "prof_active", False) or \
                self.options.capture == "off"
            gkey = (set_key, self.bp_iterations)    # (what a recorded step has baked in)
            graph = None if eager else plan["graphs"].get(gkey)
            if graph is None and not eager and self._capturable(plan, ctx, dist):
                # the whole step -- phases, exchanges, epilogue -- as ONE graph (per host set):
                # no interpreter and no launch overhead between its launches from now on
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        self._run_plan_path(plan, ctx, refs, dist, world, host_set, captured=True)
                    plan["graphs"][gkey] = graph
                except Exception as e:        # a transport / runtime that cannot be captured
                    import warnings
                    warnings.warn("raynet_amd: step capture failed (%s); eager schedule" % (e,))
                    self.options = self.options.replace(capture="off")
                    plan["graphs"], graph = {}, None
                    torch.cuda.synchronize(dev)
            if graph is not None:
                graph.replay()
                plan["ev_all"].record()
                plan["wait_ev"] = [plan["ev_all"]] * V
                T = self.bp_iterations
                self._acc_flat = plan["acc_b" if (T - 1) & 1 else "acc_a"]
                self._acc_bias = 0.0 if plan["fixed"] else plan["prior"]
            else:
                self._run_plan_path(plan, ctx, refs, dist, world, host_set)
            self.captured = graph is not None
            plan["passes"] += 1
            for r in refs:
                st = per_image[r]
                self.messages.put(r, st["msgs"], st["rvc"])
                self.voxel_count[r] = st["rvc"]
            owners = plan["owners"]
            for k, r in enumerate(refs):
                self.ref_idx = r
                if owners is not None and owners[k] != rank:
                    yield None             # another rank assembles this image (map_owner)
                    continue
                plan["wait_ev"][k].synchronize()
                a = self._lease(mset, k) if leased else host_set[k].numpy().copy()
                yield a.reshape(W, H).T
                del a
            self._pass_complete = True
            return
        if getattr(ctx, "_scatter_items", None) is not None:
            ctx.bind_scatter_items(None)     # (a work list is the plan path's; contexts are shared)
        for out in self._run_granular(scene, refs, bank, ctx, plan, dist, rank, world):
            yield out

    def _run_granular(self, scene, refs, bank, ctx, plan, dist, rank, world):
        """The resident schedule launch by launch: ray batches (`rays_batch`), filtered ray lists,
        memory-bounded groups of images, the reference's quirks, and back ends without the plan
        entry (the host stand-in of the gloo tests).  Same results as the plan path."""
        gp = self._generation_params
        H, W = scene.image_shape
        dev = ctx.device
        collective = dist is not None
        prior = plan["prior"]
        N = gp.neighbors + 1
        V = len(refs)
        cam_dev, views_of, lists, shards = plan["cam_dev"], plan["views_of"], plan["lists"], plan["shards"]
        npad, groups, patch_rows, fixed = plan["npad"], plan["groups"], plan["patch_rows"], plan["fixed"]
        vox_g, Sr_g, msgs_all, rvc_all = plan["vox"], plan["Sr"], plan["msgs"], plan["rvc"]
        # resident accumulators are flat buffers in the backend's own layout (4x4x4 bricks
        # on the GPU, include/raynet_hip.h); `self.accumulator` is handed out as [gx][gy][gz]
        acc_in, acc_next, acc_spare, acc_part = plan["acc_prior"], plan["acc_a"], plan["acc_b"], plan["acc_part"]
        if plan["dirty"]:                      # an earlier pass was abandoned inside an iteration
            acc_part.zero_()
        if self.bp_iterations == 0 or self.reference_quirks:
            # no sweep writes them (the depth sweep then reads the initial, zero messages) /
            # quirk Q2 decodes every image with the LAST image's rows, beyond that image's own
            # counts: the reference's zero-filled memmap is zero there
            msgs_all.zero_()
        per_image = plan["per_image"]

        whole = plan["shared"] is not None and not self.rays_batch and V > 0 and \
            hasattr(ctx, "scene_prepare_all")

        def prepare(group):
            """K1 prefix (traversal, plane sweep, mapping, clip + renormalise) of the group's
            images into the group buffers; image k of the group sits at rows [j*npad, ...)."""
            g0, g1 = group[0], group[-1] + 1
            if whole:
                # every image traverses / sweeps the same ray list: two launches for the group
                ridx = shards[g0][0]
                ctx.scene_prepare_all(g1 - g0, npad, ridx, plan["table"][g0:g1], cam_dev[g0:g1],
                                      vox_g[:(g1 - g0) * npad], rvc_all[g0 * npad:g1 * npad],
                                      Sr_g[:(g1 - g0) * npad])
                return
            for j, k in enumerate(group):
                r = refs[k]
                st = per_image[r]
                ridx, n = st["ridx"], st["n"]
                views = views_of[r]
                P = cam_dev[k, :12 * N]
                P_inv = cam_dev[k, 12 * N:12 * N + 12]
                B = self.rays_batch if self.rays_batch else max(n, 1)
                vox_k, Sr_k = vox_g[j * npad:j * npad + n], Sr_g[j * npad:j * npad + n]
                for i in range(0, n, B):
                    ctx.scene_prepare(ridx[i:i + B], [bank[v] for v in views], P, P_inv,
                                      st["center"], vox_k[i:i + B], st["rvc"][i:i + B],
                                      Sr_k[i:i + B])

        # K1 prefix once per reference image when every image's columns fit in HBM (they do at
        # every configuration of BASELINE.json); otherwise group by group inside every sweep.
        # The per-ray columns of a group's images live in one buffer each (image j owns rows
        # [j*npad, j*npad + n)), so that a BP iteration is ONE launch per kernel over the group
        one_group = len(groups) <= 1
        if one_group and groups:
            prepare(groups[0])
        sweep = ctx.scene_bp_sweep_fixed if fixed else ctx.scene_bp_sweep
        combine = ctx.acc_combine_fixed if fixed else ctx.acc_combine
        for it in range(self.bp_iterations):
            # iteration 0 starts from zero messages (forward_pass.py:613-615); with the
            # shipped quirk every iteration does (memmap reopened with mode="w+", Q1)
            first = it == 0 or self.reference_quirks
            plan["dirty"] = True
            for group in groups:
                if not one_group:
                    prepare(group)
                n_g = len(group) * npad
                g_row0 = group[0] * npad
                B_g = self.rays_batch // 256 * 256 if self.rays_batch and self.rays_batch >= 256 else n_g
                for i in range(0, n_g, B_g):
                    j = min(i + B_g, n_g)
                    sweep(Sr_g[i:j], vox_g[i:j], rvc_all[g_row0 + i:g_row0 + j], acc_in,
                          msgs_all[g_row0 + i:g_row0 + j], acc_part,
                          first_sweep=first, patch_rows=patch_rows,
                          uniform_acc=it == 0)      # iteration 0: the prior everywhere
            # swap + prior refill of forward_pass.py:676-678; across ranks the partial sums are
            # merged first (integer sums in the deterministic mode: the same bits whatever the
            # ring order) and the prior is added once, after the sum
            if collective:
                dist.all_reduce(acc_part, op=dist.ReduceOp.SUM)
            combine(acc_part, prior, acc_next)
            # (the prior buffer never becomes a destination)
            acc_in, acc_next = acc_next, (acc_spare if it == 0 else acc_in)
            plan["dirty"] = False
        self._acc_flat = acc_in            # `self.accumulator` regrids it when somebody looks

        # depth sweep.  One rank: image by image, and while image k+1 is decoded a side stream
        # maps image k's rows to pixels and copies the map to (pinned) host memory -- the
        # reference yields after each image's `.get()`.  Several ranks: one launch over the
        # group (each image measures from its own camera centre) and one all-gather.
        identity = not patch_rows and not self._filter_out_rays
        cuda = dev.type == "cuda"

        def to_host(src, r):
            if not identity:
                # rows -> pixels on the device (rays that were filtered out stay 0)
                full = torch.zeros((H * W,), dtype=torch.float32, device=dev)
                full.index_copy_(0, lists[r].long(), src)
                src = full
            host = torch.empty((H * W,), dtype=torch.float32, pin_memory=cuda)
            host.copy_(src, non_blocking=True)
            done = torch.cuda.Event() if cuda else None
            if done is not None:
                done.record()
            return host, done

        n_all = V * npad
        depth_all = plan["depth"]
        pending = []
        if not collective:
            if cuda and self._side_stream is None:
                self._side_stream, self._copy_stream = _side_streams(dev)
            side = self._side_stream if cuda else None
            if side is not None:
                depth_all.record_stream(side)
            last = per_image[refs[-1]] if refs else None
            for group in groups:
                if not one_group:
                    prepare(group)
                for j, k in enumerate(group):
                    r = refs[k]
                    st = per_image[r]
                    msgs = st["msgs"]
                    if self.reference_quirks and last["n"] == st["n"]:
                        msgs = last["msgs"]   # every image decoded with the LAST one's
                    dst = st["depth"]
                    Sr_k, vox_k = Sr_g[j * npad:j * npad + st["n"]], vox_g[j * npad:j * npad + st["n"]]
                    B = self.rays_batch // 256 * 256 if self.rays_batch and self.rays_batch >= 256 \
                        else max(st["n"], 1)
                    for i in range(0, st["n"], B):
                        ctx.scene_depth(Sr_k[i:i + B], vox_k[i:i + B], st["rvc"][i:i + B],
                                        acc_in, msgs[i:i + B], st["center"], None, dst[i:i + B])
                    if side is None:
                        pending.append((r,) + to_host(dst, r))
                    else:
                        ready = torch.cuda.Event()
                        ready.record()
                        with torch.cuda.stream(side):
                            side.wait_event(ready)
                            pending.append((r,) + to_host(dst, r))
        else:
            centers = cam_dev[:, 12 * N + 12:].contiguous()
            for group in groups:
                if not one_group:
                    prepare(group)
                g0, n_g = group[0], len(group) * npad
                ctx.scene_depth(Sr_g[:n_g], vox_g[:n_g], rvc_all[g0 * npad:g0 * npad + n_g], acc_in,
                                msgs_all[g0 * npad:g0 * npad + n_g], centers[g0:g0 + len(group)],
                                None, depth_all[g0 * npad:g0 * npad + n_g], rays_per_center=npad)
            # ONE all-gather of the ranks' row blocks (each rank sends only its own rows), ONE
            # gather that puts every image's rows of every rank into pixel order (its index
            # map depends on the ray lists and the sharding only: built once), ONE copy to the
            # host.  Pixels without a ray (filtered out) read the zero behind the blocks.
            HW = H * W
            if plan.get("stitch_all") is None:
                # held by the plan: the gathers never write the zero behind the blocks
                plan["gathered_all"] = torch.zeros((world * n_all + 1,), dtype=torch.float32, device=dev)
            flat = plan["gathered_all"]
            dist.all_gather_into_tensor(flat[:-1], depth_all)
            if plan.get("stitch_all") is None:
                src = torch.full((V * HW,), world * n_all, dtype=torch.int64, device=dev)
                for k, r in enumerate(refs):
                    rays = lists[r].long()
                    cuts = plan["bounds"][k]
                    for q in range(world):
                        lo_q, hi_q = cuts[q], cuts[q + 1]
                        src[k * HW + rays[lo_q:hi_q]] = (
                            q * n_all + k * npad +
                            torch.arange(hi_q - lo_q, dtype=torch.int64, device=dev))
                plan["stitch_all"] = src
            maps = flat.index_select(0, plan["stitch_all"])
            host = torch.empty((V * HW,), dtype=torch.float32, pin_memory=cuda)
            host.copy_(maps, non_blocking=True)
            done = torch.cuda.Event() if cuda else None
            if done is not None:
                done.record()
            for k, r in enumerate(refs):
                pending.append((r, host[k * HW:(k + 1) * HW], done if k == 0 else None))
        for r in refs:
            st = per_image[r]
            self.messages.put(r, st["msgs"], st["rvc"])
            self.voxel_count[r] = st["rvc"]
        # with a process group, image k's map is handed out by ONE rank (map_owner; this
        # launch-by-launch path still moves every map to every rank: it is the fallback)
        for k, (r, host, done) in enumerate(pending):
            if done is not None:
                done.synchronize()
            self.ref_idx = r
            if collective and map_owner(k, V, world) != rank:
                yield None
                continue
            yield host.numpy().reshape(W, H).T

    def _forward_pass_reference(self, scene, images_range):
        """Literal schedule of forward_pass.py:579-748 with K1 / K2 (single rank)."""
        start, end, skip = images_range
        gp = self._generation_params
        M = gp.max_number_of_marched_voxels
        H, W = scene.image_shape
        refs = list(range(start, end, skip))
        if not refs:
            return
        prior = self._prior()
        msgs_all = {}
        ctx = None
        acc = acc_out = None
        for it in range(self.bp_iterations):
            for r in refs:
                ray_idxs = self.get_valid_rays_per_image(scene, r)
                images = scene.get_image_with_neighbors(r, gp.neighbors)
                features = self._features(scene, r, images)
                ctx = self._context(scene, features.shape[-1])
                if acc is None:
                    acc = torch.full(ctx.grid_shape, prior, dtype=torch.float32, device=ctx.device)
                    acc_out = torch.full(ctx.grid_shape, prior, dtype=torch.float32,
                                         device=ctx.device)
                P, P_inv, center = (ctx.dev(a) for a in self._camera_arrays(images))
                ridx = ctx.dev(ray_idxs.astype(np.int32))
                B = self.rays_batch if self.rays_batch else len(ridx)
                B = min(B, len(ridx))
                if r not in msgs_all or (self.reference_quirks and it > 0):
                    msgs_all[r] = torch.zeros((len(ridx), M), dtype=torch.float32,
                                              device=ctx.device)
                s = torch.zeros((B, M), dtype=torch.float32, device=ctx.device)
                rvi = torch.zeros((B, M, 3), dtype=torch.int32, device=ctx.device)
                rvc = torch.zeros((B,), dtype=torch.int32, device=ctx.device)
                for i in range(0, len(ridx), B):
                    s.zero_()
                    rvi.zero_()
                    rvc.zero_()              # forward_pass.py:646-648
                    k = len(ridx[i:i + B])
                    self._fp(ridx[i:i + B], features, P, P_inv, center, self._vg, rvi[:k], rvc[:k],
                             s[:k], acc, msgs_all[r][i:i + B], acc_out)
            acc_out, acc = acc, acc_out
            acc_out.fill_(prior)             # forward_pass.py:676-678
        self.accumulator = acc
        for r in refs:
            ray_idxs = self.get_valid_rays_per_image(scene, r)
            images = scene.get_image_with_neighbors(r, gp.neighbors)
            features = self._features(scene, r, images)
            P, P_inv, center = (ctx.dev(a) for a in self._camera_arrays(images))
            ridx = ctx.dev(ray_idxs.astype(np.int32))
            B = min(self.rays_batch if self.rays_batch else len(ridx), len(ridx))
            msgs = msgs_all[refs[-1]] if self.reference_quirks else msgs_all[r]
            s = torch.zeros((B, M), dtype=torch.float32, device=ctx.device)
            rvi = torch.zeros((B, M, 3), dtype=torch.int32, device=ctx.device)
            rvc = torch.zeros((B,), dtype=torch.int32, device=ctx.device)
            depth = torch.zeros((H * W,), dtype=torch.float32, device=ctx.device)
            for i in range(0, len(ridx), B):
                s.zero_()
                rvi.zero_()
                rvc.zero_()
                k = len(ridx[i:i + B])
                self._de(ridx[i:i + B], features, P, P_inv, center, self._vg, rvi[:k], rvc[:k],
                         s[:k], acc, msgs[i:i + B], depth[i:i + B])
            self.messages[r] = msgs_all[r]
            self.ref_idx = r
            yield depth.cpu().numpy().reshape(W, H).T


def get_forward_pass_factory(name):
    # forward_pass.py:859-865.  "hartmann_fp" (a different network, not on the path)
    # is not provided.
    return {
        "multi_view_cnn": MultiViewCNNForwardPass,
        "multi_view_cnn_voxel_space": MultiViewCNNVoxelSpaceForwardPass,
        "raynet": RayNetForwardPass,
    }[name]
