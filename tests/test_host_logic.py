"""CPU-only: host-side logic of the shim that mirrors reference behaviour without touching the GPU."""
import os

import numpy as np
import pytest


def test_mask_to_wh_rule():
    from pb3d.voxel_carving_utils import _mask_to_wh
    m = np.arange(12).reshape(3, 4)
    assert _mask_to_wh(m, 4, 3).shape == (4, 3) and np.array_equal(_mask_to_wh(m, 4, 3), m.T)   # (H,W) -> .T
    assert _mask_to_wh(m, 3, 4) is not None and np.array_equal(_mask_to_wh(m, 3, 4), m)       # already (W,H)
    sq = np.arange(9).reshape(3, 3)
    assert np.array_equal(_mask_to_wh(sq, 3, 3), sq.T)   # square: the (H,W) test wins -> always transposed
    with pytest.raises(ValueError, match=r"Mask shape \(3, 4\) incompatible with \(W,H\)=\(5,4\)"):
        _mask_to_wh(m, 5, 4)


def test_argument_errors_before_any_device_work():
    import pb3d
    g = np.zeros((4, 3, 2), np.uint8)
    with pytest.raises(ValueError, match="incompatible"):
        pb3d.carve_voxel_grid_with_masks(g, np.ones((5, 5), bool))
    with pytest.raises(ValueError, match="incompatible"):
        pb3d.process_voxel_grid(g, np.ones((5, 5), bool), 90)
    with pytest.raises(TypeError):
        pb3d.process_voxel_grid(g, np.ones((3, 4), bool), 45.0)
    with pytest.raises(ValueError, match="must not be zero"):
        pb3d.process_voxel_grid(g, np.ones((3, 4), bool), 0)
    with pytest.raises(TypeError, match="not supported"):          # a dtype np.where(mask, grid, 0) would change (or could not take): refused, not guessed
        pb3d.carve_voxel_grid_with_masks(np.array([[["a"]]]), np.ones((1, 1), bool))
    with pytest.raises(RuntimeError, match="data type not supported"):      # what SciPy's interpolation says to float16 (other dtypes: csrc/rotate_typed.hip)
        pb3d.process_voxel_grid(g.astype(np.float16), np.ones((3, 4), bool), 90)
    with pytest.raises(ValueError, match="broadcast"):
        pb3d.carve_voxel_grid_with_masks(np.zeros((4, 3, 2, 3), np.uint8), np.ones((4, 3, 3), np.uint8))
    # negative step: range(0, 91, -5) is empty and upstream returns the grid unchanged
    assert np.array_equal(pb3d.process_voxel_grid(g + 1, np.ones((3, 4), bool), -5), g + 1)
    # empty part list -> empty outputs without touching the device
    p, c = pb3d.get_voxel_points_by_parts(np.zeros((2, 2, 2, 3), np.uint8), pb3d.PART_COLORS, [])
    assert p.shape == (0, 3) and p.dtype == np.float32 and c.shape == (0, 3) and c.dtype == np.uint8


def test_look_at_matches_golden(golden):
    import json
    import os
    from conftest import GOLDEN
    from pb3d.camera_geometry import look_at_rotation
    g = golden("f7_projection")
    for mon in ("Akbar", "Charminar"):
        cams = json.load(open(os.path.join(GOLDEN, f"stored_{mon}_camera_params_final.json")))
        for view in ("front", "drone"):
            for mode, dt in (("f32", np.float32), ("f64", np.float64)):
                cp = np.array(cams[view]["cam_pos"], np.float32).astype(dt)
                tg = np.array(cams[view]["target"], np.float32).astype(dt)
                R = look_at_rotation(cp, tg)
                assert R.dtype == dt and np.array_equal(R, g[f"R_{mon}_{view}_{mode}"])
    # degenerate view direction (looking straight along +Y) switches the up vector
    R = look_at_rotation(np.zeros(3, np.float32), np.array([0, 5, 0], np.float32))
    assert np.array_equal(R[2], [0, 1, 0]) and np.all(np.isfinite(R))


def test_promotion_flags():
    from pb3d.projection_utils import _promotes_to_f64
    assert not _promotes_to_f64(1.5) and not _promotes_to_f64(3)
    assert _promotes_to_f64(np.float64(1.5)) and not _promotes_to_f64(np.float32(1.5))
    assert _promotes_to_f64(np.array(2.0)) and _promotes_to_f64(np.int64(3)) and not _promotes_to_f64(np.int16(3))


def test_slab_bounds_cover_and_balance():
    from pb3d.dist import equal_slabs, slab_bounds
    for n in (1, 7, 8, 1024, 1023, 355):
        for r in (1, 2, 3, 4, 8):
            b = [slab_bounds(n, k, r) for k in range(r)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[k][1] == b[k + 1][0] for k in range(r - 1))
            sizes = [x1 - x0 for x0, x1 in b]
            assert max(sizes) - min(sizes) <= 1
    assert equal_slabs(1024, 8) == 128
    with pytest.raises(ValueError):
        equal_slabs(1023, 8)
    with pytest.raises(ValueError):
        slab_bounds(8, 8, 8)


def test_install_patches_a_utils_like_package():
    import sys
    import types
    import pb3d
    pkg = types.ModuleType("fakeutils"); sub = types.ModuleType("fakeutils.voxel_carving_utils")
    sub.process_voxel_grid = lambda *a: "old"; sub.launch_smart_aligner = lambda *a: "kept"
    pkg.process_voxel_grid = sub.process_voxel_grid
    sys.modules["fakeutils"] = pkg; sys.modules["fakeutils.voxel_carving_utils"] = sub
    try:
        patched = pb3d.install(pkg)
        assert ("fakeutils.voxel_carving_utils", "process_voxel_grid") in patched and ("fakeutils", "process_voxel_grid") in patched
        assert sub.process_voxel_grid is pb3d.process_voxel_grid and sub.launch_smart_aligner() == "kept"
    finally:
        del sys.modules["fakeutils"], sys.modules["fakeutils.voxel_carving_utils"]


def test_mask_ingest_n3(golden, tmp_path):
    """row N3: PNG -> (semantic, exterior, binary) at max_dim without OpenCV.  The Taj masks at 512 are the ones with
    which the carve path reproduces the reference's stored results/1 grid position-exactly (test_results1_*), and the
    resized shape (139, 256) is the one notebook 1 prints for Taj at max_dim = 256."""
    import shutil
    from conftest import GOLDEN
    import pb3d
    for mon in ("Taj", "Akbar"):
        d = tmp_path / "data" / mon / "masks"; d.mkdir(parents=True)
        shutil.copyfile(os.path.join(GOLDEN, f"data_{mon}_front_mask.png"), d / f"{mon}_front_mask.png")
    root = str(tmp_path / "data")
    for mon, dim, fx in (("Taj", 512, "f9_Taj_512_masks"), ("Taj", 256, "f4_Taj_256_masks"), ("Akbar", 128, "f4_Akbar_128_masks")):
        sem, ext, binary = pb3d.load_and_prepare_masks(root, mon, "front", dim, pb3d.PART_COLORS_NP, pb3d.INTERIOR_PARTS)
        g = golden(fx)
        assert np.array_equal(sem, g["sem"]) and np.array_equal(ext, g["ext"]) and np.array_equal(binary, g["binary"])
        assert binary.dtype == np.uint8 and sem.dtype == np.uint8
    assert pb3d.load_and_prepare_masks(root, "Taj", "front", 256, pb3d.PART_COLORS_NP, pb3d.INTERIOR_PARTS)[0].shape == (139, 256, 3)
    full = pb3d.load_mask(root, "Taj", "front")
    assert full.shape == (660, 1214, 3)
    assert np.array_equal(pb3d.load_mask(root, "Taj", "front", 512), golden("f9_Taj_512_masks")["sem"])
    with pytest.raises(FileNotFoundError):
        pb3d.load_mask(root, "Bibi", "front")


def test_result_pool_reuses_only_released_memory():
    """pb3d._hostmem: the opt-in pool serves a result from the memory of an earlier one only after the caller dropped it and
    everything derived from it (views keep the backing buffer's reference count up); disabled = plain np.empty."""
    import gc
    from pb3d import _hostmem as hm
    before = hm._cap_bytes >> 20
    hm.set_result_pool(0)
    a = hm.empty((4, 1 << 20), np.uint8)
    assert a.flags.owndata                                         # pool off: ordinary arrays
    hm.set_result_pool(64)
    try:
        a = hm.empty((4, 1 << 20), np.uint8); a[:] = 7
        assert not a.flags.owndata and a.flags.writeable and a.shape == (4, 1 << 20)
        addr_a = a.ctypes.data
        b = hm.empty((4, 1 << 20), np.uint8)                       # a is alive: b must not alias it
        assert b.ctypes.data != addr_a
        view = a[1:3, ::2]                                          # a derived view keeps the buffer busy after `a` is gone
        del a; gc.collect()
        c = hm.empty((3, 1 << 20), np.uint8)
        assert c.ctypes.data != addr_a and (view == 7).all()
        del view; gc.collect()
        d = hm.empty((4, 1 << 20), np.float32).view(np.uint8)       # too large for the released 4 MiB buffer (> capacity rule) -> new
        e = hm.empty((1 << 20, 4), np.uint8)                         # fits the released buffer: reused
        assert e.ctypes.data == addr_a and e.shape == (1 << 20, 4)
        small = hm.empty((10, 10), np.uint8)
        assert small.flags.owndata                                   # below 1 MiB: never pooled
        big = hm.empty((80 << 20,), np.uint8)
        assert big.flags.owndata                                     # above the pool's capacity: plain allocation
        del b, c, d, e, small, big
    finally:
        hm.set_result_pool(0)
        hm.set_result_pool(before)


def test_search_loop_drivers_reproduce_the_reference_buttons(oracle):
    """Row N4's loops (pb3d.camera_estimation.random_search / coordinate_descent / powell_search, reference
    utils/camera_estimation.py:606-726) driven with the ORACLE's objective: the parameters each button of the reference's widget left
    behind (tools/gen_golden_n4_loops.py: headless drive with a seeded np.random), bit for bit -- the random trials around the base,
    the in-place stepping of the shared camera / target arrays in the coordinate loop, the first-improvement rule, Powell through the
    caller's minimiser.  (The GPU test runs the same drivers on CameraObjective.evaluate_batch.)"""
    from scipy.optimize import minimize
    from conftest import n4_loop_cases, n4_run_case
    cases, front, grid = n4_loop_cases()
    PC = oracle.PART_COLORS

    class Objective:
        def __init__(self, parts):
            self.pts, self.cols = oracle.get_voxel_points_by_parts(grid, PC, parts)
            self.seg = oracle.mask_parts_from_image(front, PC, parts)
            self.sel = {p: PC[p] for p in parts}

        def __call__(self, p):
            H, W = self.seg.shape[:2]
            return oracle.camera_objective(self.pts, self.cols, self.seg, self.sel, p, H, W)

        def evaluate_batch(self, ps):
            return [self(p) for p in ps]

    for case in cases:
        s1, s2, s3 = n4_run_case(case, Objective(case["parts"]), minimize)
        assert s1 == case["after_random"], (case["parts"], "random")
        assert s2 == case["after_coord"], (case["parts"], "coord")
        assert s3 == case["after_powell"], (case["parts"], "powell")


def test_handoff_formats_roundtrip(tmp_path):
    """Row N3, the files either side of the path: a grid saved like notebook 1 cell 9 is what notebook 2 cell 3 loads (key
    `voxel_grid`, stored (D,H,W,3) layout: the reference's own results/1 file loads through the package), camera parameters go out as
    notebook 2 cell 11 writes them and come back by notebook 3's to_numpy rule (lists -> float32 arrays, scalars stay Python floats)."""
    import json
    from pb3d import formats
    stored = os.path.join(os.path.dirname(__file__), "golden", "stored_Akbar_voxel_grid.npz")
    g = formats.load_voxel_grid(stored)
    assert g.dtype == np.uint8 and g.ndim == 4 and g.shape[3] == 3
    p = str(tmp_path / "Akbar_voxel_grid.npz")
    formats.save_voxel_grid(p, g)
    with np.load(p) as f:
        assert list(f.keys()) == ["voxel_grid"] and np.array_equal(f["voxel_grid"], g)
    cams = {"front": {"cam_pos": np.array([1.5, 2.0, -300.25]), "target": np.array([64.0, 60.0, 64.0], np.float32), "f": 450.5, "cx": 64.0, "cy": np.float64(61.5),
                      "H": 123, "W": 128}}
    q = str(tmp_path / "Akbar_camera_params_final.json")
    formats.save_camera_params(q, cams)
    raw = json.load(open(q))
    assert raw["front"]["cam_pos"] == [1.5, 2.0, -300.25] and raw["front"]["cy"] == 61.5
    back = formats.load_camera_params(q)
    assert back["front"]["cam_pos"].dtype == np.float32 and isinstance(back["front"]["f"], float) and isinstance(back["front"]["H"], int)
    assert np.array_equal(back["front"]["target"], cams["front"]["target"])
    ref = formats.load_camera_params(os.path.join(os.path.dirname(__file__), "golden", "stored_Akbar_camera_params_final.json"))
    assert ref["front"]["cam_pos"].dtype == np.float32 and ref["front"]["cam_pos"].shape == (3,)
