"""HipContext: one rn_ctx (include/raynet_hip.h) plus the torch plumbing.

The reference compiles one PyCUDA module per (M, D, N, F, H, W, padding, bbox,
grid_shape) tuple (cuda_implementations/raynet_fp.py:230-248); here the same
tuple creates one context and the kernels take the sizes at run time.  torch is
used for device memory and streams only.
"""
import ctypes

import numpy as np
import torch

from .. import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    """Raw device pointer of a tensor for the C ABI (NULL for None).  Every pointer handed to
    the library goes through here: it must be a contiguous CUDA tensor -- a strided view or a
    host tensor would be read as something else, silently."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("the C ABI takes contiguous CUDA tensors (got %s, %s)" % (
            "cuda" if t.is_cuda else "cpu", "contiguous" if t.is_contiguous() else "strided"))
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype, min_numel=0, name="tensor", optional=False, align=4):
    """The C ABI takes raw device pointers: a tensor handed to it must be what the kernels
    assume -- on the GPU, contiguous, of the stated dtype, at least `min_numel` elements,
    `align`-byte aligned -- or the call fails HERE, not as a silent out-of-bounds access.
    Per-ray index / count arrays (ray_idxs, rvc, order, depth) are read element by element:
    4 bytes, so that any slice of them (a ray batch of 50, an odd shard bound) is accepted like
    the reference accepts it; the [n][M] row arrays are read with 128-bit accesses where M is a
    multiple of 4 (`HipContext._row_align`) and only then need 16."""
    if t is None:
        if optional:
            return t
        raise ValueError("%s: required" % name)
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError("%s: expected a CUDA tensor" % name)
    if t.dtype != dtype:
        raise ValueError("%s: expected %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s: must be contiguous" % name)
    if t.numel() < min_numel:
        raise ValueError("%s: %d elements, need >= %d" % (name, t.numel(), min_numel))
    if t.numel() and t.data_ptr() % align:
        raise ValueError("%s: data pointer is not %d-byte aligned" % (name, align))
    return t


SCATTER_ITEM_TILES = 1 << 19      # tiles an int32 item `tile << 12 | ...` can name


def scatter_work_list(rvc, M, level=0, target_items=2048):
    """int32 items `tile << 12 | first chunk << 6 | chunks` for the LDS-box scatter at tile
    `level` (0: 128 rays x 32 steps, 1: 256 x 16) from the rays' voxel counts [rows]: a tile's
    LIVE chunks (up to its longest sending ray; rays with <= 1 voxels send nothing, mrf_np.py:300)
    in items of U chunks, U such that the launch has about `target_items` workgroups -- a tile of
    long rays becomes several items, a border tile one, a tile no ray of which sends anything none
    -- sorted longest first, tiles in order within a length.  None when nothing is live."""
    tile, steps = (128, 32) if level == 0 else (256, 16)
    c = rvc.to(torch.int64).clamp(max=M)
    c = torch.where(c <= 1, torch.zeros_like(c), c)
    pad = (-c.numel()) % tile
    if pad:
        c = torch.cat([c, torch.zeros((pad,), dtype=c.dtype, device=c.device)])
    nch = (c.view(-1, tile).max(1).values + steps - 1) // steps          # live chunks per tile
    total = int(nch.sum().item())
    if total == 0 or len(nch) > SCATTER_ITEM_TILES:
        # (an item's tile field has 19 bits: a larger plan keeps the scatter's tiles x split launch;
        # rn_scene_bind_scatter_items refuses such rows as well)
        return None
    U = int(min(63, max(1, -(-total // int(target_items)))))
    per_tile = (nch + U - 1) // U                                        # items per tile
    tile_of = torch.repeat_interleave(torch.arange(len(nch), device=c.device), per_tile)
    first = torch.cumsum(per_tile, 0) - per_tile
    begin = (torch.arange(len(tile_of), device=c.device) - first[tile_of]) * U
    cnt = torch.minimum(torch.full_like(begin, U), nch[tile_of] - begin)
    order = torch.sort(-cnt, stable=True).indices
    return ((tile_of << 12) | (begin << 6) | cnt)[order].to(torch.int32).contiguous()


def to_device(x, dtype=None, device="cuda"):
    """The reference's `to_gpu` / all_arrays_to_gpu (cuda_implementations/utils.py:11-22):
    NumPy arrays are uploaded, device tensors pass through untouched."""
    if isinstance(x, torch.Tensor):
        t = x
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        if not t.is_cuda:
            t = t.to(device)
        return t.contiguous()
    a = np.ascontiguousarray(x)
    t = torch.from_numpy(a)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.to(device)


class HipContext(object):
    def __init__(self, M, D, N, F, H, W, padding, bbox, grid_shape, device=None):
        if not torch.cuda.is_available():
            raise _lib.RaynetHipError(
                "no GPU visible: raynet_amd runs its hot path on MI355X only (no CPU fallback)")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None
                                   else device)
        bbox = np.asarray(bbox, dtype=np.float32).reshape(-1)
        assert bbox.shape == (6,)
        grid_shape = [int(g) for g in np.asarray(grid_shape).reshape(-1)]
        assert len(grid_shape) == 3
        cfg = _lib.Config()
        cfg.M, cfg.D, cfg.N, cfg.F = int(M), int(D), int(N), int(F)
        cfg.H, cfg.W, cfg.padding = int(H), int(W), int(padding)
        for i in range(3):
            cfg.grid[i] = grid_shape[i]
        for i in range(6):
            cfg.bbox[i] = float(bbox[i])
        cfg.device = self.device.index
        self.M, self.D, self.N, self.F = int(M), int(D), int(N), int(F)
        self.H, self.W, self.padding = int(H), int(W), int(padding)
        self.bbox = bbox
        self.grid_shape = tuple(grid_shape)
        self.G = grid_shape[0] * grid_shape[1] * grid_shape[2]
        self.feature_shape = (self.N, self.H + self.padding + 1, self.W + self.padding + 1, self.F)
        # rows of the [n][M] arrays start 16-byte aligned iff M % 4 == 0 -- exactly when the
        # kernels use their 128-bit row accesses (k_traverse's flush, k_scatter_slab's loads)
        self._row_align = 16 if self.M % 4 == 0 else 4
        handle = ctypes.c_void_p()
        rc = self.lib.rn_create(ctypes.byref(cfg), ctypes.byref(handle))
        if rc != _lib.RN_OK:
            raise _lib.RaynetHipError("rn_create failed: %s" % _lib.STATUS.get(rc, rc))
        self._h = handle
        self._grid_set = False
        self._options = None

    def set_options(self, options):
        """Apply a PathOptions' context half (rn_options); a no-op when nothing changed."""
        want = options.context_options()
        if self._options == want:
            return
        o = _lib.Options(*want)
        self._check(self.lib.rn_set_options(self._h, ctypes.byref(o)))
        self._options = want

    def get_options(self):
        o = _lib.Options()
        self._check(self.lib.rn_get_options(self._h, ctypes.byref(o)))
        return dict(scatter_mode=o.scatter_mode, box_level=o.box_level, box_pin=bool(o.box_pin),
                    overlap=o.overlap, generic_sweep=bool(o.generic_sweep),
                    sweep_rays_per_wave=o.sweep_rays_per_wave)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.rn_destroy(h)
            self._h = None

    # ------------------------------------------------------------------
    def _check(self, rc):
        if rc != _lib.RN_OK:
            raise _lib.RaynetHipError("%s: %s" % (
                _lib.STATUS.get(rc, rc), self.lib.rn_last_error(self._h).decode()))

    def dev(self, x, dtype=None):
        return to_device(x, dtype, self.device)

    def set_voxel_grid(self, voxel_grid):
        vg = self.dev(voxel_grid, torch.float32)
        assert vg.numel() == self.G * 3
        self._check(self.lib.rn_set_voxel_grid(self._h, _ptr(vg), _stream()))
        self._grid_set = True
        self._vg_keepalive = vg

    def acc_copies(self):
        return int(self.lib.rn_acc_copies(self._h))

    def scatter_reset(self):
        self._check(self.lib.rn_scatter_reset(self._h))

    def bind_slab_boxes(self, vox, table=None):
        """Let rn_scene_prepare_all keep slab boxes for the list buffer `vox` ([rows][M] i32;
        None unbinds): the scatters of a pass then merge boxes instead of scanning the lists
        (include/raynet_hip.h).  Returns the scratch table (pass it back in to re-bind the same
        buffer later: contexts are shared, the binding is the last caller's)."""
        if vox is None:
            self._check(self.lib.rn_scene_bind_slab_boxes(self._h, None, 0, None))
            self._slab_boxes = None
            return None
        _chk(vox, torch.int32, self.M, "vox", align=self._row_align)
        rows = vox.numel() // self.M
        size = int(self.lib.rn_slab_boxes_size(self._h, rows))
        if table is None:
            table = torch.empty((size,), dtype=torch.int32, device=self.device)
        _chk(table, torch.int32, size, "slab-box table", align=8)
        self._slab_boxes = (table, vox)
        self._check(self.lib.rn_scene_bind_slab_boxes(self._h, _ptr(vox), rows, _ptr(table)))
        return table

    def bind_scatter_items(self, vox, rvc=None, level=0, target_items=2048, items=None):
        """Work list of the LDS-box scatter over ALL rows of `vox` ([rows][M]; None unbinds), built
        from the rays' voxel counts `rvc` [rows] by scatter_work_list (or `items`: a list built
        before, to bind again -- contexts are shared, the binding is the last caller's).  One host
        synchronisation (the item count) per build; the list depends on the counts only, i.e. on
        cameras and shard, not on a pass (rn_scene_bind_scatter_items)."""
        if vox is None:
            self._check(self.lib.rn_scene_bind_scatter_items(self._h, None, 0, 0, None, 0))
            self._scatter_items = None
            return None
        rows = vox.numel() // self.M
        if items is None:
            items = scatter_work_list(rvc, self.M, level, target_items)
        if items is None or items.numel() == 0:
            return self.bind_scatter_items(None)
        _chk(items, torch.int32, 1, "scatter items")
        self._scatter_items = (items, vox, level)
        self._check(self.lib.rn_scene_bind_scatter_items(self._h, _ptr(vox), rows, int(level),
                                                         _ptr(items), int(items.numel())))
        return items

    def _drop_foreign_items(self, vox):
        """A work list bound for another tensor must not meet a launch whose buffer happens to sit
        at the address (and have the row count) the list was built for: unbind unless `vox` IS the
        tensor it describes."""
        cur = getattr(self, "_scatter_items", None)
        if cur is not None and cur[1] is not vox:
            self.bind_scatter_items(None)

    def scatter_items_bound(self, items):
        cur = getattr(self, "_scatter_items", None)
        return cur is not None and cur[0] is items

    def slab_boxes_bound_to(self, vox):
        sb = getattr(self, "_slab_boxes", None)
        return sb is not None and sb[1] is vox

    def scatter_state(self):
        """-> (tile level in use, chunks, overflowed chunks of the last counted launch)"""
        lv, ch, ov = ctypes.c_int32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self.lib.rn_scatter_state(self._h, ctypes.byref(lv), ctypes.byref(ch),
                                              ctypes.byref(ov)))
        return int(lv.value), int(ch.value), int(ov.value)

    def scatter_settled(self):
        """No scatter launch touches the host any more and the tile shape stays as it is
        (rn_scatter_settled): a step's launches may be recorded into a HIP graph."""
        return bool(self.lib.rn_scatter_settled(self._h))

    # resident accumulators are 4x4x4-bricked (include/raynet_hip.h); the reference's
    # [gx][gy][gz] view is produced / consumed through these two
    def acc_size(self):
        return int(self.lib.rn_acc_size(self._h))

    def acc_to_grid(self, acc):
        out = torch.empty(self.grid_shape, dtype=torch.float32, device=acc.device)
        self._check(self.lib.rn_acc_to_grid(self._h, _ptr(acc), _ptr(out), _stream()))
        return out

    def stitch_rows(self, rows, index, out):
        """out[i] = rows[index[i]] (rn_stitch_rows) on the current stream; `out` is a CUDA tensor
        or a PINNED host tensor -- the kernel then writes the map across PCIe itself.  The
        caller vouches for index < rows.numel() (a plan checks its tables once)."""
        n = int(index.numel())
        _chk(rows, torch.float32, 1 if n else 0, "rows")
        _chk(index, torch.int32, n, "index", align=16)
        if not (isinstance(out, torch.Tensor) and out.dtype == torch.float32 and
                out.is_contiguous() and out.numel() >= n and (out.is_cuda or out.is_pinned()) and
                (n == 0 or out.data_ptr() % 16 == 0)):
            raise ValueError("out: expected a contiguous float32 CUDA or pinned tensor of >= %d "
                             "elements, 16-byte aligned" % n)
        self._check(self.lib.rn_stitch_rows(self._h, n, _ptr(rows), _ptr(index), out.data_ptr(),
                                            _stream()))
        return out

    def acc_from_grid(self, grid):
        grid = self.dev(grid, torch.float32).contiguous()
        out = torch.zeros((self.acc_size(),), dtype=torch.float32, device=grid.device)
        self._check(self.lib.rn_acc_from_grid(self._h, _ptr(grid), _ptr(out), _stream()))
        return out

    # -- timing (bench.py): hipEvents on the stream the kernels run on ------
    def timer_start(self):
        self._check(self.lib.rn_timer_start(self._h, _stream()))

    def timer_stop(self):
        ms = ctypes.c_float()
        self._check(self.lib.rn_timer_stop(self._h, _stream(), ctypes.byref(ms)))
        return float(ms.value)

    # ---- differentiable MRF block (training) -----------------------------------------
    def _chk_api_rows(self, rvi, rvc, *float_rows):
        """K-API layouts: rvi [n][M][3] i32, rvc [n] i32, float rows [n][M] f32."""
        n = len(rvc)
        _chk(rvi, torch.int32, n * self.M * 3, "ray_voxel_indices")
        _chk(rvc, torch.int32, n, "ray_voxel_count")
        for k, t in enumerate(float_rows):
            _chk(t, torch.float32, n * self.M, "row array %d" % k, optional=True)
        return n

    def plane_weights(self, rvi, rvc, starts, ends, left, c1, c2):
        n = self._chk_api_rows(rvi, rvc, c1, c2)
        _chk(left, torch.int32, n * self.M, "left")
        _chk(starts, torch.float32, 3 * n, "starts", align=4)
        _chk(ends, torch.float32, 3 * n, "ends", align=4)
        self._check(self.lib.rn_plane_weights(self._h, len(rvc), _ptr(rvi), _ptr(rvc), _ptr(starts),
                                              _ptr(ends), _ptr(left), _ptr(c1), _ptr(c2),
                                              _stream()))

    def train_bp_sweep(self, Sr, rvi, rvc, acc_in, msgs_in, acc_out, msgs_out):
        self._chk_api_rows(rvi, rvc, Sr, msgs_in, msgs_out)
        _chk(acc_in, torch.float32, self.G, "acc_in"); _chk(acc_out, torch.float32, self.G, "acc_out")
        self._check(self.lib.rn_train_bp_sweep(self._h, len(rvc), _ptr(Sr), _ptr(rvi), _ptr(rvc),
                                               _ptr(acc_in), _ptr(msgs_in), _ptr(acc_out),
                                               _ptr(msgs_out), _stream()))

    def train_depth(self, Sr, rvi, rvc, acc, msgs, S_new):
        self._chk_api_rows(rvi, rvc, Sr, msgs, S_new)
        _chk(acc, torch.float32, self.G, "acc")
        self._check(self.lib.rn_train_depth(self._h, len(rvc), _ptr(Sr), _ptr(rvi), _ptr(rvc),
                                            _ptr(acc), _ptr(msgs), _ptr(S_new), _stream()))

    def train_bp_sweep_bwd(self, Sr, rvi, rvc, acc_in, msgs_in, g_msgs_out, g_acc_out, g_Sr,
                           g_acc_in, g_msgs_in):
        self._chk_api_rows(rvi, rvc, Sr, msgs_in, g_msgs_out, g_Sr, g_msgs_in)
        for name, t in (("acc_in", acc_in), ("g_acc_out", g_acc_out), ("g_acc_in", g_acc_in)):
            _chk(t, torch.float32, self.G, name)
        self._check(self.lib.rn_train_bp_sweep_bwd(
            self._h, len(rvc), _ptr(Sr), _ptr(rvi), _ptr(rvc), _ptr(acc_in), _ptr(msgs_in),
            _ptr(g_msgs_out), _ptr(g_acc_out), _ptr(g_Sr), _ptr(g_acc_in), _ptr(g_msgs_in),
            _stream()))

    def train_depth_bwd(self, Sr, rvi, rvc, acc, msgs, g_S_new, g_Sr, g_acc, g_msgs):
        self._chk_api_rows(rvi, rvc, Sr, msgs, g_S_new, g_Sr, g_msgs)
        _chk(acc, torch.float32, self.G, "acc"); _chk(g_acc, torch.float32, self.G, "g_acc")
        self._check(self.lib.rn_train_depth_bwd(
            self._h, len(rvc), _ptr(Sr), _ptr(rvi), _ptr(rvc), _ptr(acc), _ptr(msgs),
            _ptr(g_S_new), _ptr(g_Sr), _ptr(g_acc), _ptr(g_msgs), _stream()))

    # ---- consumers of the depth maps (point clouds, metrics) --------------------------
    def depthmap_points(self, H, W, P_pinv, center, depth_map, points):
        self._check(self.lib.rn_depthmap_points(self._h, int(H), int(W), _ptr(P_pinv), _ptr(center),
                                                _ptr(depth_map), _ptr(points), _stream()))

    def consistency_tau(self, H, W, first, points, P, center, depth_map, tau):
        self._check(self.lib.rn_consistency_tau(self._h, points.shape[1], int(H), int(W),
                                                1 if first else 0, _ptr(points), _ptr(P),
                                                _ptr(center), _ptr(depth_map), _ptr(tau),
                                                _stream()))

    def nearest_neighbors(self, ref_xyzw, query_xyzw, dist, idx=None):
        self._check(self.lib.rn_nearest_neighbors(self._h, ref_xyzw.shape[0], _ptr(ref_xyzw),
                                                  query_xyzw.shape[0], _ptr(query_xyzw),
                                                  _ptr(dist), _ptr(idx), _stream()))

    KERNEL_NAMES = {1: "traverse", 2: "sweep_map", 3: "bp", 4: "depth", 5: "acc", 6: "other", 7: "scatter"}

    def prof_begin(self, capacity=4096, only=None):
        """only: kernel family names to bracket (None = all of them)."""
        mask = 0xFFFFFFFF
        if only is not None:
            ids = {v: k for k, v in self.KERNEL_NAMES.items()}
            mask = 0
            for name in only:
                mask |= 1 << ids[name]
        self._check(self.lib.rn_prof_select(self._h, mask))
        self._prof_cap = capacity
        self.prof_active = True       # (event pairs around launches: such a pass is not captured)
        self._check(self.lib.rn_prof_begin(self._h, capacity))

    def prof_end(self):
        """-> list of (kernel name, n_rays, milliseconds), one per launch."""
        cap = self._prof_cap
        self.prof_active = False
        ids = (ctypes.c_int32 * cap)()
        rays = (ctypes.c_int32 * cap)()
        ms = (ctypes.c_float * cap)()
        n = ctypes.c_int32()
        self._check(self.lib.rn_prof_end(self._h, ctypes.byref(n), ids, rays, ms))
        starts = (ctypes.c_float * cap)()
        self._check(self.lib.rn_prof_offsets(self._h, starts))
        self.prof_starts = [float(starts[i]) for i in range(n.value)]
        return [(self.KERNEL_NAMES.get(ids[i], str(ids[i])), int(rays[i]), float(ms[i]))
                for i in range(n.value)]

    def selftest_arith(self, a, out):
        """out[2][n]: roundf(a), round_half_away(a) (tests only)."""
        self._check(self.lib.rn_selftest_arith(self._h, a.numel(), _ptr(a), _ptr(out), _stream()))

    def selftest_quotient(self, x, d, out):
        """out[3][n]: round_half_away(x / d), round_quotient_fast's value, sure (tests only)."""
        self._check(self.lib.rn_selftest_quotient(self._h, x.numel(), _ptr(x), _ptr(d), _ptr(out),
                                                  _stream()))

    def selftest_feature_offsets(self, P, starts, ends, out):
        """out [n][N][D][2] int32: feature-vector index per ray / view / plane by the generic
        and by the cooperative sweep's index arithmetic (tests only)."""
        self._check(self.lib.rn_selftest_feature_offsets(self._h, len(starts), _ptr(P), _ptr(starts),
                                                         _ptr(ends), _ptr(out), _stream()))

    def selftest_mapping(self, a, b, t, out):
        """out[5][n]: a / b, Markstein's quotient, usable, the walk's plane index, the table's
        (tests only)."""
        self._check(self.lib.rn_selftest_mapping(self._h, a.numel(), _ptr(a), _ptr(b), _ptr(t),
                                                 _ptr(out), _stream()))

    # -- thin wrappers; argument order is the header's ------------------------
    def fill_f32(self, t, value):
        self._check(self.lib.rn_fill_f32(self._h, _ptr(t), t.numel(), float(value), _stream()))

    def fill_i32(self, t, value):
        self._check(self.lib.rn_fill_i32(self._h, _ptr(t), t.numel(), int(value), _stream()))

    def sample_rays(self, ray_idxs, P_inv, center, starts, ends):
        self._check(self.lib.rn_sample_rays(self._h, len(ray_idxs), _ptr(ray_idxs), _ptr(P_inv),
                                            _ptr(center), _ptr(starts), _ptr(ends), _stream()))

    def sample_points(self, ray_idxs, P_inv, center, points):
        self._check(self.lib.rn_sample_points(self._h, len(ray_idxs), _ptr(ray_idxs), _ptr(P_inv),
                                              _ptr(center), _ptr(points), _stream()))

    def compute_similarities(self, features, P, starts, ends, S):
        self._check(self.lib.rn_compute_similarities(self._h, len(starts), _ptr(features), _ptr(P),
                                                     _ptr(starts), _ptr(ends), _ptr(S), _stream()))

    def voxel_traversal(self, starts, ends, rvi, rvc):
        self._check(self.lib.rn_voxel_traversal(self._h, len(starts), _ptr(starts), _ptr(ends),
                                                _ptr(rvi), _ptr(rvc), _stream()))

    def planes_to_voxels(self, rvi, rvc, starts, ends, S, S_new):
        self._check(self.lib.rn_planes_to_voxels(self._h, len(rvc), _ptr(rvi), _ptr(rvc),
                                                 _ptr(starts), _ptr(ends), _ptr(S), _ptr(S_new),
                                                 _stream()))

    def bp_sweep(self, S, rvi, rvc, acc_in, msgs_in, acc_out, msgs_out):
        self._check(self.lib.rn_bp_sweep(self._h, len(rvc), _ptr(S), _ptr(rvi), _ptr(rvc),
                                         _ptr(acc_in), _ptr(msgs_in), _ptr(acc_out),
                                         _ptr(msgs_out), _stream()))

    def depth_estimation(self, S, rvi, rvc, acc, msgs, S_new):
        self._check(self.lib.rn_depth_estimation(self._h, len(rvc), _ptr(S), _ptr(rvi), _ptr(rvc),
                                                 _ptr(acc), _ptr(msgs), _ptr(S_new), _stream()))

    def mvcnn_similarities(self, ray_idxs, features, P, P_inv, center, S):
        self._check(self.lib.rn_mvcnn_similarities(self._h, len(ray_idxs), _ptr(ray_idxs),
                                                   _ptr(features), _ptr(P), _ptr(P_inv),
                                                   _ptr(center), _ptr(S), _stream()))

    def mvcnn_depth(self, ray_idxs, features, P, P_inv, center, S, points, depth_map):
        self._check(self.lib.rn_mvcnn_depth(self._h, len(ray_idxs), _ptr(ray_idxs), _ptr(features),
                                            _ptr(P), _ptr(P_inv), _ptr(center), _ptr(S),
                                            _ptr(points), _ptr(depth_map), _stream()))

    def mvcnn_voxel_space(self, ray_idxs, features, P, P_inv, center, rvi, rvc, S_voxel,
                          depth_map=None):
        if depth_map is None:
            self._check(self.lib.rn_mvcnn_voxel_space(
                self._h, len(ray_idxs), _ptr(ray_idxs), _ptr(features), _ptr(P), _ptr(P_inv),
                _ptr(center), _ptr(rvi), _ptr(rvc), _ptr(S_voxel), _stream()))
        else:
            self._check(self.lib.rn_mvcnn_voxel_space_depth(
                self._h, len(ray_idxs), _ptr(ray_idxs), _ptr(features), _ptr(P), _ptr(P_inv),
                _ptr(center), _ptr(rvi), _ptr(rvc), _ptr(S_voxel), _ptr(depth_map), _stream()))

    def fused_bp_sweep(self, ray_idxs, features, P, P_inv, center, rvi, rvc, S_voxel, acc_in,
                       msgs_in, acc_out, msgs_out):
        self._check(self.lib.rn_fused_bp_sweep(
            self._h, len(ray_idxs), _ptr(ray_idxs), _ptr(features), _ptr(P), _ptr(P_inv),
            _ptr(center), _ptr(rvi), _ptr(rvc), _ptr(S_voxel), _ptr(acc_in), _ptr(msgs_in),
            _ptr(acc_out), _ptr(msgs_out), _stream()))

    def fused_depth(self, ray_idxs, features, P, P_inv, center, rvi, rvc, S_voxel, acc, msgs,
                    depth_map):
        self._check(self.lib.rn_fused_depth(
            self._h, len(ray_idxs), _ptr(ray_idxs), _ptr(features), _ptr(P), _ptr(P_inv),
            _ptr(center), _ptr(rvi), _ptr(rvc), _ptr(S_voxel), _ptr(acc), _ptr(msgs),
            _ptr(depth_map), _stream()))

    # -- resident-scene path ---------------------------------------------------
    def _segments(self, rows):
        """Scratch for the rays' bbox entry / exit points ([rows][8] f32), kept per context."""
        seg = getattr(self, "_seg_scratch", None)
        if seg is None or seg.shape[0] < rows:
            seg = torch.empty((rows, 8), dtype=torch.float32, device=self.device)
            self._seg_scratch = seg
        return seg

    def scene_prepare(self, ray_idxs, feature_views, P, P_inv, center, vox, rvc, Sr, order=None):
        assert len(feature_views) == self.N
        n = len(ray_idxs)
        f32, i32 = torch.float32, torch.int32
        fdim = self.feature_shape[1] * self.feature_shape[2] * self.F
        for k, fv in enumerate(feature_views):
            _chk(fv, f32, fdim, "feature map %d" % k, align=16)
        _chk(ray_idxs, i32, n, "ray_idxs"); _chk(order, i32, n, "order", optional=True)
        _chk(P, f32, 12 * self.N, "P"); _chk(P_inv, f32, 12, "P_inv")
        _chk(center, f32, 3, "center")
        ra = self._row_align
        _chk(vox, i32, n * self.M, "vox", align=ra); _chk(rvc, i32, n, "rvc")
        _chk(Sr, f32, n * self.M, "Sr", align=ra)
        assert order is None or len(order) == n
        arr = (ctypes.c_void_p * self.N)(*[fv.data_ptr() for fv in feature_views])
        self._check(self.lib.rn_scene_prepare(self._h, n, _ptr(ray_idxs), arr, _ptr(P),
                                              _ptr(P_inv), _ptr(center), _ptr(order), _ptr(vox),
                                              _ptr(rvc), _ptr(Sr),
                                              _ptr(self._segments(n)), _stream()))

    def scene_prepare_all(self, n_images, rows_per_image, ray_idxs, feature_table, cameras, vox,
                          rvc, Sr, order=None):
        """feature_table: int64 CUDA tensor [n_images, N] of device pointers;
        cameras: float32 CUDA tensor [n_images, 12N + 16]."""
        n, rows = len(ray_idxs), int(n_images) * int(rows_per_image)
        f32, i32 = torch.float32, torch.int32
        _chk(feature_table, torch.int64, n_images * self.N, "feature_table", align=8)
        assert tuple(feature_table.shape) == (n_images, self.N)
        _chk(cameras, f32, n_images * (12 * self.N + 16), "cameras")
        assert tuple(cameras.shape) == (n_images, 12 * self.N + 16)
        _chk(ray_idxs, i32, n, "ray_idxs"); _chk(order, i32, n, "order", optional=True)
        ra = self._row_align
        _chk(vox, i32, rows * self.M, "vox", align=ra); _chk(rvc, i32, rows, "rvc")
        _chk(Sr, f32, rows * self.M, "Sr", align=ra)
        assert order is None or len(order) == n
        self._check(self.lib.rn_scene_prepare_all(
            self._h, int(n_images), n, int(rows_per_image), _ptr(ray_idxs),
            _ptr(feature_table), _ptr(cameras), _ptr(order), _ptr(vox), _ptr(rvc), _ptr(Sr),
            _ptr(self._segments(rows)), _stream()))

    def scene_plan(self, n_images, rows_per_image, ray_idxs, feature_table, cameras, vox, rvc, Sr,
                   msgs, acc0, acc1, depth, prior, patch_rows, acc_fixed=None, order=None,
                   depth_image=None, sweep_xcd_chunk=0):
        """An rn_scene_plan over the caller's buffers, every tensor validated ONCE here; the
        returned object (which keeps them alive) goes to scene_run.  depth_image [n_images, R]:
        the depth sweeps write the maps in ray-index (pixel) order there instead of `depth`."""
        n, rows = len(ray_idxs), int(n_images) * int(rows_per_image)
        f32, i32, ra = torch.float32, torch.int32, self._row_align
        assert rows_per_image % 256 == 0 and n <= rows_per_image
        _chk(feature_table, torch.int64, n_images * self.N, "feature_table", align=8)
        assert tuple(feature_table.shape) == (n_images, self.N)
        _chk(cameras, f32, n_images * (12 * self.N + 16), "cameras")
        assert tuple(cameras.shape) == (n_images, 12 * self.N + 16)
        _chk(ray_idxs, i32, n, "ray_idxs"); _chk(order, i32, n, "order", optional=True)
        _chk(vox, i32, rows * self.M, "vox", align=ra); _chk(rvc, i32, rows, "rvc")
        _chk(Sr, f32, rows * self.M, "Sr", align=ra); _chk(msgs, f32, rows * self.M, "msgs", align=ra)
        _chk(depth, f32, rows, "depth")
        G = self.acc_size()
        _chk(acc0, f32, G, "acc[0]", align=16); _chk(acc1, f32, G, "acc[1]", align=16)
        _chk(acc_fixed, torch.int64, G, "acc_fixed", optional=True, align=16)
        seg = self._segments(rows)
        pl = _lib.ScenePlan()
        pl.n_images, pl.n, pl.rows_per_image = int(n_images), n, int(rows_per_image)
        pl.ray_idxs, pl.order = ray_idxs.data_ptr(), (order.data_ptr() if order is not None else None)
        pl.features_views, pl.cameras = feature_table.data_ptr(), cameras.data_ptr()
        pl.vox, pl.rvc, pl.Sr, pl.msgs = vox.data_ptr(), rvc.data_ptr(), Sr.data_ptr(), msgs.data_ptr()
        pl.ray_segments = seg.data_ptr()
        pl.acc[0], pl.acc[1] = acc0.data_ptr(), acc1.data_ptr()
        pl.acc_fixed = acc_fixed.data_ptr() if acc_fixed is not None else None
        pl.depth, pl.prior = depth.data_ptr(), float(prior)
        pl.row_layout = 1 if patch_rows else 0
        pl.depth_image, pl.depth_image_stride = None, 0
        assert sweep_xcd_chunk >= 0 and sweep_xcd_chunk % 4 == 0
        pl.sweep_xcd_chunk = int(sweep_xcd_chunk)
        if depth_image is not None:
            assert depth_image.dim() == 2 and depth_image.shape[0] == n_images
            _chk(depth_image, f32, depth_image.numel(), "depth_image")
            if n:       # every ray index is a valid entry of an image's map (checked once, here)
                lo, hi = int(ray_idxs.min()), int(ray_idxs.max())
                assert 0 <= lo and hi < depth_image.shape[1], (lo, hi, tuple(depth_image.shape))
            pl.depth_image, pl.depth_image_stride = depth_image.data_ptr(), int(depth_image.shape[1])
        # (the tensors the struct points into live as long as it does.  No `byref(pl)` is kept ON
        # pl: that is a reference cycle through a ctypes object the collector does not track --
        # every plan ever built, with its 7 GB of buffers, stayed allocated: round 6)
        pl._keepalive = (ray_idxs, feature_table, cameras, vox, rvc, Sr, msgs, acc0, acc1, depth,
                         acc_fixed, order, seg, depth_image)
        return pl

    def scene_run(self, plan, phases, iteration=0, image=-1):
        """rn_scene_run: the phases (a mask of _lib.RN_RUN_*) of one pass, one C call."""
        self._check(self.lib.rn_scene_run(self._h, ctypes.byref(plan), int(phases), int(iteration),
                                          int(image), _stream()))

    def count_voxels(self, ray_idxs, cameras):
        """-> int32 [n_images, n]: voxels crossed by every ray of ray_idxs in every reference
        image (rn_scene_count_voxels; cameras as for scene_prepare_all)."""
        n_images, n = int(cameras.shape[0]), len(ray_idxs)
        _chk(cameras, torch.float32, n_images * (12 * self.N + 16), "cameras")
        _chk(ray_idxs, torch.int32, n, "ray_idxs")
        out = torch.zeros((n_images, n), dtype=torch.int32, device=self.device)
        self._check(self.lib.rn_scene_count_voxels(self._h, n_images, n, _ptr(ray_idxs),
                                                   _ptr(cameras), _ptr(out), _stream()))
        return out

    def _chk_rows(self, Sr, vox, rvc, msgs, acc, part=None, part_dtype=torch.float32):
        n = len(rvc)
        f32, i32 = torch.float32, torch.int32
        ra = self._row_align
        _chk(Sr, f32, n * self.M, "Sr", align=ra); _chk(vox, i32, n * self.M, "vox", align=ra)
        _chk(rvc, i32, n, "rvc")
        _chk(msgs, f32, n * self.M, "msgs", align=ra)
        _chk(acc, f32, self.acc_size(), "accumulator", align=16)
        if part is not None:
            _chk(part, part_dtype, self.acc_size(), "partial accumulator", align=16)
        return n

    def scene_bp_sweep(self, Sr, vox, rvc, acc_in, msgs, acc_part, first_sweep=False,
                       patch_rows=False, uniform_acc=False):
        n = self._chk_rows(Sr, vox, rvc, msgs, acc_in, acc_part)
        self._drop_foreign_items(vox)
        self._check(self.lib.rn_scene_bp_sweep(self._h, n, _ptr(Sr), _ptr(vox), _ptr(rvc),
                                               _ptr(acc_in), _ptr(msgs), _ptr(acc_part),
                                               (1 if first_sweep else 0) | (2 if uniform_acc else 0),
                                               1 if patch_rows else 0, _stream()))

    def scene_bp_sweep_fixed(self, Sr, vox, rvc, acc_in, msgs, acc_part_fixed, first_sweep=False,
                             patch_rows=False, uniform_acc=False):
        n = self._chk_rows(Sr, vox, rvc, msgs, acc_in, acc_part_fixed, torch.int64)
        self._drop_foreign_items(vox)
        self._check(self.lib.rn_scene_bp_sweep_fixed(
            self._h, n, _ptr(Sr), _ptr(vox), _ptr(rvc), _ptr(acc_in), _ptr(msgs),
            _ptr(acc_part_fixed), (1 if first_sweep else 0) | (2 if uniform_acc else 0),
            1 if patch_rows else 0, _stream()))

    def acc_combine_fixed(self, acc_part_fixed, prior, acc_out):
        _chk(acc_part_fixed, torch.int64, acc_out.numel(), "partial accumulator", align=16)
        _chk(acc_out, torch.float32, 1, "accumulator", align=16)
        assert acc_out.numel() == self.acc_size()
        self._check(self.lib.rn_acc_combine_fixed(self._h, _ptr(acc_part_fixed), float(prior),
                                                  _ptr(acc_out), _stream()))

    def acc_combine_fixed_range(self, part_fixed, prior, out):
        """fixed -> float on any slab: out[i] = prior + part_fixed[i] * 2^-32; zeroes the slab."""
        n = part_fixed.numel()
        _chk(part_fixed, torch.int64, n, "fixed-point slab", align=8)
        _chk(out, torch.float32, n, "accumulator slab")
        self._check(self.lib.rn_acc_combine_fixed_range(self._h, _ptr(part_fixed), n, float(prior),
                                                        _ptr(out), _stream()))

    def acc_combine(self, acc_part, prior, acc_out):
        _chk(acc_part, torch.float32, self.acc_size(), "partial accumulator")
        _chk(acc_out, torch.float32, self.acc_size(), "accumulator")
        self._check(self.lib.rn_acc_combine(self._h, _ptr(acc_part), float(prior), _ptr(acc_out),
                                            _stream()))

    def acc_reduce_local(self, acc_part, acc_out):
        _chk(acc_part, torch.float32, self.acc_size(), "partial accumulator")
        _chk(acc_out, torch.float32, self.acc_size(), "accumulator")
        self._check(self.lib.rn_acc_reduce_local(self._h, _ptr(acc_part), _ptr(acc_out), _stream()))

    def acc_add_prior(self, acc, prior):
        _chk(acc, torch.float32, self.acc_size(), "accumulator")
        self._check(self.lib.rn_acc_add_prior(self._h, _ptr(acc), float(prior), _stream()))

    def scene_depth(self, Sr, vox, rvc, acc, msgs, center, S_new, depth_map, rays_per_center=0):
        n = self._chk_rows(Sr, vox, rvc, msgs, acc)
        _chk(S_new, torch.float32, n * self.M, "S_new", optional=True)
        _chk(depth_map, torch.float32, n, "depth_map", optional=True)
        groups = (n + rays_per_center - 1) // rays_per_center if rays_per_center > 0 else 1
        _chk(center, torch.float32, 4 * groups if rays_per_center > 0 else 3, "center",
             optional=depth_map is None, align=4)
        self._check(self.lib.rn_scene_depth(self._h, n, _ptr(Sr), _ptr(vox), _ptr(rvc),
                                            _ptr(acc), _ptr(msgs), _ptr(center),
                                            int(rays_per_center), _ptr(S_new), _ptr(depth_map),
                                            _stream()))
