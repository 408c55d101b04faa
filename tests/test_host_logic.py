"""Host-side pieces of the resident path that need no GPU: the patch row order, the lazily
cleared message container, ray sharding."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("H,W,tx,ty,along", [(480, 640, 16, 16, True), (37, 53, 16, 16, False),
                                              (33, 17, 8, 32, True), (16, 16, 16, 16, False)])
def test_tile_order_is_a_permutation_of_compact_patches(H, W, tx, ty, along):
    from raynet_amd.forward_pass import tile_order
    n = H * W
    order = tile_order(torch.arange(n, dtype=torch.int32), H, W, tx, ty, along_rows=along).numpy()
    assert order.dtype == np.int32 and np.array_equal(np.sort(order), np.arange(n))
    x, y = order // H, order % H
    # every run of tx*ty consecutive rows that starts on a patch boundary of a FULL patch
    # covers exactly that patch
    if H % ty == 0 and W % tx == 0:
        for start in range(0, n, tx * ty):
            px, py = x[start:start + tx * ty], y[start:start + tx * ty]
            assert px.max() - px.min() == tx - 1 and py.max() - py.min() == ty - 1
        # patches are enumerated along the requested direction
        first = order[::tx * ty]
        fx, fy = first // H // tx, first % H // ty
        key = fy * (W // tx) + fx if along else fx * (H // ty) + fy
        assert np.array_equal(key, np.arange(len(first)))
    # a subset of the rays (filter_out_rays) keeps the relative order of the full list
    rng = np.random.default_rng(0)
    keep = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32)
    sub = tile_order(torch.from_numpy(keep), H, W, tx, ty, along_rows=along).numpy()
    assert np.array_equal(sub, order[np.isin(order, keep)])


def test_messages_container_clears_tails_on_first_access():
    from raynet_amd.forward_pass import _Messages
    m = _Messages()
    raw = torch.full((5, 8), 7.0)
    counts = torch.tensor([0, 1, 3, 8, 5], dtype=torch.int32)
    m.put(2, raw, counts)
    assert 2 in m and float(raw.sum()) == 7.0 * 40          # nothing touched yet
    got = m[2]
    assert got is raw
    expect = np.full((5, 8), 7.0, np.float32)
    expect[0] = 0                      # count 0
    expect[1] = 0                      # count 1: such rays send no message
    expect[2, 3:] = 0
    expect[4, 5:] = 0
    assert np.array_equal(got.numpy(), expect)
    raw[3, 0] = 1.0
    assert float(m[2][3, 0]) == 1.0    # cleared once, not again
    m[9] = torch.ones((2, 2))          # plain assignment (the reference-schedule path)
    assert float(m[9].sum()) == 4.0


def test_shard_bounds_cover_the_list():
    from raynet_amd.forward_pass import shard_bounds
    for n in (0, 1, 7, 307200, 307201):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_bench_timeline_shares_split_concurrent_launches():
    """bench.py picks the dominant kernel family by its share of the timeline: launches that
    overlap on two streams split the time they share, sequential ones keep their durations."""
    import importlib.util
    import os
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # a: [0, 2), b: [1, 3) on another stream, c: [3, 4)
    launches = [("a", 0, 2.0), ("b", 0, 2.0), ("c", 0, 1.0)]
    shares = bench.timeline_shares(launches, [0.0, 1.0, 3.0])
    assert shares == pytest.approx({"a": 1.5, "b": 1.5, "c": 1.0})
    assert sum(shares.values()) == pytest.approx(4.0)          # the union of the intervals
    # same family twice, back to back: plain sum
    assert bench.timeline_shares([("a", 0, 1.0), ("a", 0, 2.0)], [0.0, 1.0]) == pytest.approx({"a": 3.0})
    # no start times: sums of durations
    assert bench.timeline_shares(launches, []) == pytest.approx({"a": 2.0, "b": 2.0, "c": 1.0})
    assert bench.timeline_shares([], []) == {}


def test_path_options_defaults_env_overrides_and_key():
    """One object for every knob; the environment only overrides defaults, parsed in one place."""
    from raynet_amd.hip_implementations.options import PathOptions, shard_alpha_for
    d = PathOptions()
    assert d.ray_tile == (16, 16) and d.overlap == 2 and d.plan_path
    # fixed-point sums: by the rank count unless said otherwise (a sharded run's
    # result is the one-rank result, bit for bit, by default)
    assert d.deterministic is None and not d.fixed_point(1) and d.fixed_point(2) and d.fixed_point(8)
    assert PathOptions(deterministic=False).fixed_point(8) is False
    assert PathOptions(deterministic=True).fixed_point(1) is True
    assert PathOptions.from_env({"RAYNET_DETERMINISTIC": "auto"}).deterministic is None
    assert PathOptions.from_env({"RAYNET_DETERMINISTIC": "0"}).deterministic is False
    assert d.context_options() == (-1, 0, 0, 2, 0, 0)
    assert PathOptions.from_env({"RAYNET_HIP_SWEEP_RAYS_PER_WAVE": "1"}).context_options()[5] == 1
    env = {"RAYNET_RAY_TILE": "0", "RAYNET_DETERMINISTIC": "1", "RAYNET_HIP_OVERLAP": "1",
           "RAYNET_SHARD_ALPHA": "0.45", "RAYNET_RESIDENT_GB": "1.5", "RAYNET_SLAB_BOXES": "0",
           "RAYNET_HIP_BOX_LEVEL": "1", "RAYNET_EXCHANGE": "reduce_scatter", "UNRELATED": "x"}
    o = PathOptions.from_env(env)
    assert o.ray_tile is None and o.deterministic and o.overlap == 1 and o.shard_alpha == 0.45
    assert o.resident_gb == 1.5 and not o.slab_boxes and o.box_level == 1
    assert o.exchange == "reduce_scatter"
    assert PathOptions.from_env({"RAYNET_RAY_TILE": "8x32"}).ray_tile == (8, 32)
    # explicit arguments beat the environment; None means "not given"
    assert PathOptions.from_env(env, deterministic=False, overlap=None).deterministic is False
    assert PathOptions.from_env(env, overlap=None).overlap == 1
    # a value that differs changes the key (plans are keyed on it), replace() leaves the source
    assert d.key() != o.key() and d.replace(box_pin=True).key() != d.key() and not d.box_pin
    assert set(d.as_dict()) == {f for f in PathOptions.__dataclass_fields__ if f != "ENV"}
    import json
    json.dumps(o.as_dict())                      # bench.py echoes it
    with pytest.raises(AssertionError):
        PathOptions(shard="tiles")
    # the balance constant follows the shape: plane-sweep work ~ N D per ray, BP work ~ voxels
    assert 0.3 < shard_alpha_for(5, 64, 137.4) < 0.45
    assert 0.55 < shard_alpha_for(9, 128, 270.0) < 0.8


def test_forward_pass_object_carries_its_options():
    from raynet_amd.common.generation_parameters import GenerationParameters
    from raynet_amd.forward_pass import RayNetForwardPass
    from raynet_amd.hip_implementations.options import PathOptions
    gp = GenerationParameters()
    fp = RayNetForwardPass(None, gp, "sample_in_bbox", (8, 8), 0, deterministic=True,
                           options=PathOptions(ray_tile=(8, 32)))
    assert fp.deterministic and fp.ray_tile == (8, 32)
    fp.ray_tile = None                           # tests / tools flip single knobs on the object
    assert fp.options.ray_tile is None and fp.options.deterministic
    fp.deterministic = False
    assert not fp.options.deterministic


def test_scatter_work_list_covers_every_live_chunk_once():
    """hip_implementations/context.py scatter_work_list: items `tile << 12 | first << 6 | chunks`
    cover every chunk below a tile's longest sending ray exactly once, none beyond it, none for
    tiles whose rays send nothing (count <= 1, mrf_np.py:300), longest items first."""
    import torch
    from raynet_amd.hip_implementations.context import scatter_work_list
    rng = np.random.default_rng(0)
    for rows, M, level in ((1000, 384, 0), (5000, 768, 1), (130, 96, 0)):
        rvc = torch.from_numpy(rng.integers(0, M + 30, rows).astype(np.int32))
        rvc[:300] = 1
        tile, steps = (128, 32) if level == 0 else (256, 16)
        c = np.minimum(rvc.numpy(), M)
        c = np.where(c <= 1, 0, c)
        c = np.concatenate([c, np.zeros((-rows) % tile, np.int64)])
        nch = (c.reshape(-1, tile).max(1) + steps - 1) // steps
        for target in (8, 64, 4096):
            items = scatter_work_list(rvc, M, level, target)
            if nch.sum() == 0:
                assert items is None
                continue
            it = items.numpy()
            cover = np.zeros((len(nch), 64), np.int64)
            for v in it:
                t, b, n = v >> 12, (v >> 6) & 63, v & 63
                assert n >= 1
                cover[t, b:b + n] += 1
            for t in range(len(nch)):
                assert (cover[t, :nch[t]] == 1).all() and (cover[t, nch[t]:] == 0).all()
            assert (np.diff(it & 63) <= 0).all()
            assert len(it) <= max(int(nch.sum()), 1)
    assert scatter_work_list(torch.ones(300, dtype=torch.int32), 96, 0) is None


def test_scatter_work_list_stops_where_an_item_cannot_name_the_tile():
    """An int32 item carries 19 bits of tile index: at the last representable plan the
    highest tile still comes out right, one tile more and there is no list (the scatter then deals
    tiles x chunks out itself; rn_scene_bind_scatter_items refuses such rows too)."""
    import torch
    from raynet_amd.hip_implementations.context import SCATTER_ITEM_TILES, scatter_work_list
    tiles = SCATTER_ITEM_TILES
    rvc = torch.zeros((tiles * 128,), dtype=torch.int32)
    rvc[-1] = 40                                     # only the LAST tile is live: chunks 0 and 1
    items = scatter_work_list(rvc, 96, 0, target_items=2048)
    assert items is not None and items.dtype == torch.int32
    got = sorted((int(v) >> 12, (int(v) >> 6) & 63, int(v) & 63) for v in items)
    assert all(int(v) >= 0 for v in items) and got == [(tiles - 1, 0, 1), (tiles - 1, 1, 1)]
    rvc = torch.cat([rvc, torch.full((128,), 40, dtype=torch.int32)])
    assert scatter_work_list(rvc, 96, 0, target_items=2048) is None


def test_leased_maps_return_their_set_when_the_last_view_dies():
    """forward_pass._take_set / _lease (PathOptions.maps = "lease"): an array handed out holds its
    set until it and every view of it are gone; a pass never gets a set somebody holds; beyond
    MAX_LEASED_SETS the caller gets the scratch set (copies).  No GPU needed: plain host tensors."""
    import gc
    from raynet_amd.forward_pass import RayNetForwardPass
    from raynet_amd.hip_implementations.options import PathOptions
    fp = RayNetForwardPass.__new__(RayNetForwardPass)
    fp.options = PathOptions()
    plan = dict(sets=[], scratch=None)
    V, HW = 3, 8
    k0, s0, leased = fp._take_set(plan, V, HW, False)
    assert (k0, leased) == (0, True) and len(s0["live"]) == 0
    s0["host"].copy_(__import__("torch").arange(V * HW, dtype=__import__("torch").float32).view(V, HW))
    a = [fp._lease(s0, k) for k in range(V)]
    assert len(s0["live"]) == 3 and a[1][2] == 10.0 and a[1].ctypes.data == s0["host"][1].data_ptr()
    view = a[2].reshape(4, 2).T[1:, 1:]
    del a
    gc.collect()
    assert len(s0["live"]) == 1                         # the slice of image 2 is still out there
    k1, s1, _ = fp._take_set(plan, V, HW, False)
    assert k1 == 1 and s1 is not s0
    del view
    gc.collect()
    assert len(s0["live"]) == 0 and fp._take_set(plan, V, HW, False)[0] == 0
    held = []
    for i in range(fp.MAX_LEASED_SETS + 2):
        k, st, leased = fp._take_set(plan, V, HW, False)
        if leased:
            held.append(fp._lease(st, 0))
        assert (k == -1) == (i >= fp.MAX_LEASED_SETS) and leased == (k >= 0)
    assert len(plan["sets"]) == fp.MAX_LEASED_SETS and plan["scratch"] is not None
    fp.options = PathOptions(maps="copy")
    assert fp._take_set(plan, V, HW, False)[0] == -1


def test_leases_taken_and_dropped_from_several_threads():
    """The lease book-keeping is a set of tokens (single add / discard operations), not a counter:
    eight threads leasing and dropping arrays of one set at once leave it free, and it is never
    seen free while an array is held."""
    import gc
    import threading
    from raynet_amd.forward_pass import RayNetForwardPass
    st = RayNetForwardPass._new_set(2, 16, False)
    keep = RayNetForwardPass._lease(st, 0)             # held throughout: the set is never free
    seen_free = []

    def worker():
        for i in range(2000):
            a = RayNetForwardPass._lease(st, i & 1)
            if not st["live"]:
                seen_free.append(i)
            del a
    threads = [threading.Thread(target=worker) for _ in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    gc.collect()
    assert not seen_free and len(st["live"]) == 1
    del keep
    gc.collect()
    assert not st["live"]


def test_the_committed_bench_line_keeps_the_drivers_contract():
    """The round's last bench line (profiles/, produced by `python bench.py` on an MI355X) carries what the
    driver's contract asks of the ONE JSON line -- metric / value / unit / n_gpus / steps / warmup /
    ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload -- plus the two
    objects of this tier, `roofline` and `cpu_baseline`, and its numbers agree with each other."""
    import glob
    import json
    import os
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = sorted(glob.glob(os.path.join(REPO, "profiles", "r06_r_bench.json")))
    assert lines, "no committed bench line"
    d = json.loads([l for l in open(lines[-1]) if l.startswith("{")][-1])
    for k, t in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int),
                 ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str),
                 ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)]:
        assert isinstance(d[k], t), (k, type(d[k]))
    assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert d["metric"].startswith("rays/sec") and d["unit"] == "rays/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = rays of a step / time of a step
    assert d["value"] == pytest.approx(d["config"]["rays_per_step"] / (d["ms_per_step"] * 1e-3), rel=1e-3)
    # every timed step's wall time is in the line; their mean is the step time (one rank: no barrier wait)
    each = d["step_ms"]["each"]
    assert len(each) == min(d["steps"], 32) and sum(each) / len(each) == pytest.approx(d["ms_per_step"], rel=0.02)
    r = d["roofline"]
    assert r["bound"] in ("hbm", "valu", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=1e-3) and 0 < r["frac"] < 1
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9,
                                          rel=1e-3)
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.5
    # the dominant kernel's launches fit into the steps they belong to
    assert r["avg_launch_ms"] * r["launches"] <= d["ms_per_step"] * d["steps"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "rays/s" and c["cores"] >= 1 and c["value"] > 0
    assert isinstance(c["sample"], str) and c["sample"]
    assert d["value"] > 100 * c["value"]                 # (a reported baseline, never the target)
    v = d["path_roofline"]["valu"]
    assert 0 < v["frac_of_step"] < 1 and v["busy_ms_per_step"] < d["ms_per_step"]
