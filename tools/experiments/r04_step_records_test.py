"""The test that went with tools/experiments/r04_step_records.patch (a function of
tests/test_forward_pass_gpu.py: it uses that module's _rank_main)."""


def test_step_records_give_the_scatter_the_same_lists(torch, tmp_path, monkeypatch):
    """RAYNET_HIP_STEP_LISTS=1 (experiment): k_traverse also leaves the lists
    as step records -- first voxel + 2-bit axis codes, 0.5 bytes per step -- and the box scatter
    decodes those instead of reading the 4-byte words.  The same lists, so in the fixed-point
    mode the same accumulator and maps, bit for bit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    runs = {}
    for tag, env in (("plain", None), ("steps", "1")):
        out = tmp_path / tag
        out.mkdir()
        if env is None:
            monkeypatch.delenv("RAYNET_HIP_STEP_LISTS", raising=False)
        else:
            monkeypatch.setenv("RAYNET_HIP_STEP_LISTS", env)
        p = ctx.Process(target=_rank_main, args=(0, 1, 0, str(out), True))
        p.start()
        p.join(300)
        assert p.exitcode == 0
        runs[tag] = np.load(str(out / "dw1_r0.npz"))
    assert np.array_equal(runs["plain"]["acc"], runs["steps"]["acc"])
    assert np.array_equal(runs["plain"]["depth"], runs["steps"]["depth"])


