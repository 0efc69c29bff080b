"""Host-side data-format logic: KITTI / nuScenes .bin readers and the
deterministic subsample / pad to a fixed point count.  Files are written by the test itself (the
reference ships no data)."""
import numpy as np
import pytest


def test_read_and_fix_size(tmp_path, sad):
    from sad_amd import io
    rng = np.random.default_rng(0)
    big = rng.normal(size=(5000, 4)).astype("<f4")
    small = rng.normal(size=(300, 4)).astype("<f4")
    nus = rng.normal(size=(700, 5)).astype("<f4")
    for name, arr in (("big.bin", big), ("small.bin", small), ("nus.pcd.bin", nus)):
        arr.tofile(tmp_path / name)
    np.testing.assert_array_equal(io.read_bin(str(tmp_path / "big.bin")), big)
    np.testing.assert_array_equal(io.read_bin(str(tmp_path / "nus.pcd.bin"), io.NUSCENES_COLS), nus)
    a = io.load_scene(str(tmp_path / "big.bin"), 1024, seed=3)
    b = io.load_scene(str(tmp_path / "big.bin"), 1024, seed=3)
    assert a.shape == (1024, 4) and a.dtype == np.float32 and a.flags.c_contiguous
    np.testing.assert_array_equal(a, b)                                   # deterministic
    assert not np.array_equal(a, io.load_scene(str(tmp_path / "big.bin"), 1024, seed=4))
    rows = {r.tobytes() for r in big}
    assert all(r.tobytes() in rows for r in a) and len({r.tobytes() for r in a}) == 1024   # a subset, no repeats
    p = io.load_scene(str(tmp_path / "small.bin"), 1024)
    np.testing.assert_array_equal(p[:300], small)                         # all points kept, then repeats
    assert all(r.tobytes() in {q.tobytes() for q in small} for r in p[300:])
    n = io.load_scene(str(tmp_path / "nus.pcd.bin"), 512, cols=io.NUSCENES_COLS, use_cols=4)
    assert n.shape == (512, 4)
    batch = io.load_batch([str(tmp_path / "big.bin"), str(tmp_path / "small.bin")], 256)
    assert batch.shape == (2, 256, 4)
    assert io.fix_size(np.zeros((0, 4), np.float32), 8).shape == (8, 4)


def test_crop_and_bad_file(tmp_path, sad):
    from sad_amd import io
    pts = np.array([[1, 0, 0, .5], [80, 0, 0, .5], [10, -50, 0, .5], [10, 5, -1, .2]], np.float32)
    got = io.crop_range(pts, (0, 70.4, -40, 40, -3, 1))
    np.testing.assert_array_equal(got, pts[[0, 3]])
    (tmp_path / "bad.bin").write_bytes(b"\0" * 10)
    with pytest.raises(ValueError):
        io.read_bin(str(tmp_path / "bad.bin"))


def test_fix_size_equals_oracle_subsample_pad(orc, sad):
    """SPEC.md §17: io.fix_size (numpy) and the C oracle pick the same rows for ragged scenes (more points than
    needed, fewer, exactly enough, one point, empty)."""
    from sad_amd import io
    rng = np.random.default_rng(17)
    sizes = [5000, 300, 1024, 1, 0, 1025, 40000]
    parts = [rng.normal(size=(n, 4)).astype(np.float32) for n in sizes]
    pts = np.concatenate(parts, 0)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    for seed in (0, 7, 123456789):
        got = orc.subsample_pad(pts, offs, 1024, seed)
        for b, p in enumerate(parts):
            np.testing.assert_array_equal(got[b], io.fix_size(p, 1024, seed, scene=b), err_msg=f"scene {b} seed {seed}")
    big = io.fix_size(parts[0], 1024, 3, scene=0)
    rows = {r.tobytes() for r in parts[0]}
    assert all(r.tobytes() in rows for r in big) and len({r.tobytes() for r in big}) == 1024      # a subset, no repeats
    sel = io.select_rows(5000, 1024, 3, 0)
    assert (np.diff(sel) > 0).all() and sel.max() - sel.min() > 4000                                # file order, spread over the file
