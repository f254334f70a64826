"""GPU parity tests (run with -m gpu on an MI355X): the hand-written HIP path, called through the
C-ABI by the NumPy shim, must be BIT-EXACT with (i) the golden vectors captured from the reference
and (ii) the CPU oracle on seeded inputs -- integer / byte / index work, so the bar is equality."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, PKG, ROOT

pytestmark = pytest.mark.gpu

JOBS_NB1 = [(["full_building"], 90), (["chhatris"], 90), (["plinth"], 90), (["front_minarets"], 90),
            (["small_minarets"], 90), (["dome"], 90)]
JOBS_MIXED = [(["full_building", "plinth"], 90), (["chhatris"], 45), (["dome"], 60), (["front_minarets", "small_minarets"], 90)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_native_library_is_loaded(pb3d_gpu):
    """The extension that runs is the in-tree libpb3d.so (no eager/CPU path exists to fall back to)."""
    maps = open("/proc/self/maps").read()
    assert "libpb3d.so" in maps
    info = __import__("pb3d").device.device_info()
    assert "gfx950" in info["name"], info


# ---- golden vectors from the reference ----------------------------------------------------------------
def test_carve_golden(pb3d_gpu, golden):
    g = golden("f2_carve")
    for i in range(int(g["n"])):
        out = pb3d_gpu.carve_voxel_grid_with_masks(g[f"grid_{i}"], g[f"mask_{i}"])
        assert out.dtype == np.uint8 and out.shape == g[f"out_{i}"].shape and np.array_equal(out, g[f"out_{i}"]), i


def test_rotate_step_golden_scipy(pb3d_gpu, golden):
    """single affine_transform steps against SciPy's own outputs (no carve: mask=None)."""
    import ctypes as C
    lib, L = pb3d_gpu._lib.load(), pb3d_gpu._lib
    g = golden("f3_affine")
    for i in range(int(g["n"])):
        x = np.ascontiguousarray(g[f"in_{i}"]); a = int(g[f"angle_{i}"])
        W, H, D = x.shape
        M = np.empty(9); off = np.empty(3)
        L.check(lib.pb3d_rotinv(a, L.p_dbl(M)))
        L.check(lib.pb3d_offset(L.p_dbl(M), (C.c_int64 * 3)(W, H, D), L.p_dbl(off)))
        out = np.empty_like(x)
        L.check(lib.pb3d_rotate_carve(L.ctx(), L.p_u8(x), W, H, D, L.p_dbl(M), L.p_dbl(off), None, L.p_u8(out)))
        assert np.array_equal(out, g[f"out_{i}"]), (i, x.shape, a, int((out != g[f"out_{i}"]).sum()))


def test_process_golden(pb3d_gpu, golden):
    g = golden("f3_process")
    for i in range(int(g["n"])):
        out = pb3d_gpu.process_voxel_grid(g[f"grid_{i}"], g[f"mask_{i}"], int(g[f"ai_{i}"]))
        want = g[f"out_{i}"]
        assert np.array_equal(out, want), (i, want.shape, int(g[f"ai_{i}"]), int((out != want).sum()))


@pytest.mark.parametrize("name", ["f4_Akbar_64", "f4_Bibi_64", "f4_Taj_96"])
def test_real_masks_golden(pb3d_gpu, golden, name):
    g = golden(name)
    gc = pb3d_gpu.global_carve(g["binary"], g["ext"], angle_interval=90)
    assert gc.shape == g["global_carve"].shape and np.array_equal(gc, g["global_carve"])
    assert np.array_equal(pb3d_gpu.global_carve(g["binary"], g["ext"], angle_interval=45), g["global_carve_45"])
    assert np.array_equal(pb3d_gpu.part_carve(gc, g["ext"], JOBS_NB1), g["part_carve_nb1"])
    assert np.array_equal(pb3d_gpu.part_carve(gc, g["ext"], JOBS_MIXED), g["part_carve_mixed"])
    from pb3d.voxel_carving_utils import _occupancy
    assert np.array_equal(_occupancy(gc), np.any(g["global_carve"] > 0, -1).astype(np.uint8))


def test_square_mask_double_transpose_golden(pb3d_gpu, golden):
    g = golden("f4_square_64")
    gc = pb3d_gpu.global_carve(g["binary"], g["ext"], 90)
    assert np.array_equal(gc, g["global_carve"])
    assert np.array_equal(pb3d_gpu.part_carve(gc, g["ext"], JOBS_NB1), g["part_carve_nb1"])


@pytest.mark.parametrize("key", ["Akbar_128", "Bibi_128", "Taj_256"])
def test_digests_configs_1_and_2(pb3d_gpu, golden, key):
    """BASELINE configs[0] (Bibi front mask, 128^3) and configs[1] (Taj front mask, 256^3)."""
    d = json.load(open(os.path.join(GOLDEN, "f4_digests.json")))[key]
    g = golden(f"f4_{key}_masks")
    gc = pb3d_gpu.global_carve(g["binary"], g["ext"], 90)
    assert list(gc.shape) == d["shape"] and sha(gc) == d["global_carve_sha256"]
    assert sha(pb3d_gpu.part_carve(gc, g["ext"], JOBS_NB1)) == d["part_carve_nb1_sha256"]


def test_results1_taj512_pinned_parts(pb3d_gpu, golden):
    """bit-exact with results/1.Orthographic_Voxel_Carving (Taj, 512): the parts the 90-degree path pins."""
    g = golden("f9_Taj_512_masks")
    stored = np.load(os.path.join(GOLDEN, "stored_Taj_voxel_grid.npz"))["voxel_grid"]
    gc = pb3d_gpu.global_carve(g["binary"], g["ext"], 90)
    pc = pb3d_gpu.part_carve(gc, g["ext"], JOBS_NB1)
    oriented = np.flip(pc.transpose(2, 1, 0, 3), axis=1)
    PC = pb3d_gpu.PART_COLORS
    eq = lambda grid, name: np.all(grid == np.array(PC[name], np.uint8), axis=-1)
    for part in ("plinth", "chhatris"):
        assert np.array_equal(eq(oriented, part), eq(stored, part)), part
    body = lambda grid: eq(grid, "full_building") | eq(grid, "main_door") | eq(grid, "windows")
    assert np.array_equal(body(oriented), body(stored))


def test_points_golden(pb3d_gpu, golden):
    grid = np.load(os.path.join(GOLDEN, "stored_Akbar_voxel_grid.npz"))["voxel_grid"]
    meta = json.load(open(os.path.join(GOLDEN, "f6_points_akbar.json")))
    PC = pb3d_gpu.PART_COLORS
    for key, m in meta.items():
        kind, arg = key.split(":")
        if kind == "parts":
            p, c = pb3d_gpu.get_voxel_points_by_parts(grid, PC, arg.split(","))
        else:
            p, c, shp = pb3d_gpu.voxel_grid_to_points(grid, stride=int(arg))
            assert list(shp) == m["shape"]
        assert p.dtype == np.float32 and c.dtype == np.uint8 and len(p) == m["n"], key
        assert sha(p) == m["pts_sha256"] and sha(c) == m["cols_sha256"], key


def _cams(mon):
    from pb3d.formats import load_camera_params          # the notebook-3 to_numpy rule (cell 3)
    return load_camera_params(os.path.join(GOLDEN, f"stored_{mon}_camera_params_final.json"))


@pytest.mark.parametrize("mon", ["Akbar", "Charminar"])
def test_projection_and_iou_golden(pb3d_gpu, golden, mon):
    """config 3 as the reference implements it: stored grid -> points -> pinhole projection with the stored
    front and aerial cameras -> per-part IoU."""
    g = golden("f7_projection")
    summ = json.load(open(os.path.join(GOLDEN, "f7_projection_summary.json")))
    grid = np.load(os.path.join(GOLDEN, f"stored_{mon}_voxel_grid.npz"))["voxel_grid"]
    PC = pb3d_gpu.PART_COLORS
    pts, col = pb3d_gpu.get_voxel_points_by_parts(grid, PC, list(PC))
    cams = _cams(mon)
    for view in ("front", "drone"):
        img = g[f"img_{mon}_{view}"]
        for mode in ("f32", "f64"):
            key = f"{mon}_{view}_{mode}"
            s = summ[key]
            cp, tg = cams[view]["cam_pos"], cams[view]["target"]
            if mode == "f64":
                cp, tg = cp.astype(np.float64), tg.astype(np.float64)
            proj = pb3d_gpu.project_colored_voxels(pts, col, cp, tg, cams[view]["f"], cams[view]["cx"], cams[view]["cy"], s["H"], s["W"])
            assert np.array_equal(proj, g[f"proj_{key}"]), (key, int((proj != g[f"proj_{key}"]).any(-1).sum()))
            per, mean = pb3d_gpu.compute_partwise_iou(proj, img, PC)
            assert {k: float(v) for k, v in per.items()} == s["iou"] and float(mean) == s["mean"]


def test_projection_synth_modes_golden(pb3d_gpu, golden):
    g = golden("f7_projection_synth")
    tmap = {"float": float, "float64": np.float64, "float32": np.float32}
    for i in range(int(g["n"])):
        f, cx, cy = (tmap[t](v) for t, v in zip(g[f"ftypes_{i}"], g[f"fcxcy_{i}"]))
        H, W = (int(v) for v in g[f"hw_{i}"])
        out = pb3d_gpu.project_colored_voxels(g[f"pts_{i}"], g[f"cols_{i}"], g[f"cam_{i}"], g[f"tgt_{i}"], f, cx, cy, H, W)
        assert np.array_equal(out, g[f"out_{i}"]), (i, int((out != g[f"out_{i}"]).any(-1).sum()))


# ---- seeded inputs against the oracle -----------------------------------------------------------------
def test_carve_vs_oracle_shapes(pb3d_gpu, oracle):
    rng = np.random.default_rng(7)
    shapes = [(1, 1, 1), (3, 2, 5), (17, 9, 16), (8, 5, 48), (64, 64, 64), (5, 7, 1024), (33, 31, 100), (2, 130, 16),
              (40, 3, 1040), (9, 9, 2741),
              # column sizes that are not multiples of 16 bytes: the flat-stream kernel (pieces straddling two columns)
              (7, 5, 17), (11, 4, 355), (2, 3, 21), (6, 5, 4099), (3, 2, 15), (37, 29, 123), (5, 5, 16), (1, 7, 33)]
    for (W, H, D) in shapes:
        for nd in (3, 4):
            grid = rng.integers(0, 256, (W, H, D) + ((3,) if nd == 4 else ()), dtype=np.uint8)
            for frac in (0.0, 0.5, 1.0):
                m = rng.random((H, W)) < frac
                assert np.array_equal(pb3d_gpu.carve_voxel_grid_with_masks(grid, m), oracle.carve_voxel_grid_with_masks(grid, m)), (W, H, D, nd, frac)
    # other dtypes / channel counts (upstream's np.where keeps the grid's dtype; SURVEY 8(b): the signature takes any array)
    for dt, tail in ((np.float32, ()), (np.int16, (3,)), (np.float64, (2,)), (np.uint8, (4,)), (np.int64, ()), (np.uint16, (3,))):
        grid = (rng.normal(size=(9, 7, 13) + tail) * 100).astype(dt)
        m = rng.random((7, 9)) < 0.5
        got = pb3d_gpu.carve_voxel_grid_with_masks(grid, m)
        want = np.where(m.T[:, :, None] if not tail else m.T[:, :, None, None], grid, 0)
        assert got.dtype == want.dtype == np.dtype(dt) and np.array_equal(got, want), dt
    gb = pb3d_gpu.carve_voxel_grid_with_masks(np.ones((3, 3, 3), bool), np.eye(3, dtype=bool))      # np.where(mask, bool_grid, 0): int64
    assert gb.dtype == np.int64 and np.array_equal(gb, np.where(np.eye(3, dtype=bool).T[:, :, None], np.ones((3, 3, 3), bool), 0))
    with pytest.raises(TypeError):
        pb3d_gpu.carve_voxel_grid_with_masks(np.array([[["a"]]]), np.ones((1, 1), bool))
    # empty grids
    for shp in [(0, 4, 4), (4, 0, 4), (4, 4, 0)]:
        e = np.zeros(shp, np.uint8)
        out = pb3d_gpu.carve_voxel_grid_with_masks(e, np.ones((shp[1], shp[0]), bool) if shp[0] != shp[1] else np.ones((shp[0], shp[1]), bool))
        assert out.shape == shp


def test_process_vs_oracle_angles(pb3d_gpu, oracle):
    rng = np.random.default_rng(11)
    for (W, H, D) in [(31, 6, 31), (40, 5, 28), (7, 3, 50), (96, 4, 96), (1, 2, 1), (130, 2, 260)]:
        for kind in ("bin", "full"):
            g = (rng.random((W, H, D)) < 0.5).astype(np.uint8) if kind == "bin" else rng.integers(0, 256, (W, H, D), dtype=np.uint8)
            m = rng.random((H, W)) < 0.9
            for ai in (90, 45, 30, 10 if W < 100 else 18, 7 if W < 40 else 23, 1 if W < 10 else 15, 91, 200):
                got = pb3d_gpu.process_voxel_grid(g, m, ai)
                want = oracle.process_voxel_grid(g, m, ai)
                assert np.array_equal(got, want), (W, H, D, kind, ai, int((got != want).sum()))


def test_color_apply_and_occupancy_vs_oracle(pb3d_gpu, oracle):
    rng = np.random.default_rng(13)
    from pb3d.voxel_carving_utils import _occupancy
    # D % 16 == 0: one column per 16-voxel group; other D >= 16: groups of the flat stream straddle two columns (+ voxel tail);
    # D < 16: voxel by voxel
    for (W, H, D) in [(16, 9, 16), (5, 4, 7), (32, 3, 64), (3, 3, 33), (7, 5, 17), (11, 6, 355), (9, 13, 123), (4, 3, 31), (1, 1, 16),
                      (2, 5, 19)]:
        carved = rng.integers(0, 3, (W, H, D), dtype=np.uint8)  # values 0,1,2: only == 1 is coloured
        rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        assert np.array_equal(pb3d_gpu.apply_colored_mask_to_voxel_grid(carved, rgb), oracle.apply_colored_mask_to_voxel_grid(carved, rgb))
        grid = rng.integers(0, 2, (W, H, D, 3), dtype=np.uint8) * rng.integers(0, 256, (W, H, D, 3), dtype=np.uint8)
        assert np.array_equal(_occupancy(grid), oracle.occupancy(grid))


def test_part_carve_vs_oracle_foreign_colours(pb3d_gpu, oracle, golden):
    """grids that hold colours the mask does not (kept by global_carve, dropped by part_carve)."""
    rng = np.random.default_rng(17)
    g = golden("f4_Bibi_64")
    colored = g["global_carve"].copy()
    k = rng.random(colored.shape[:3]) < 0.1
    colored[k] = rng.integers(0, 256, (int(k.sum()), 3), dtype=np.uint8)
    for jobs in (JOBS_NB1, JOBS_MIXED, [], [(["windows"], 90)]):
        assert np.array_equal(pb3d_gpu.part_carve(colored, g["ext"], jobs), oracle.part_carve(colored, g["ext"], jobs)), jobs


def test_points_vs_oracle(pb3d_gpu, oracle):
    rng = np.random.default_rng(19)
    pal = np.array(list(oracle.PART_COLORS.values()) + [(0, 0, 0)], np.uint8)
    for shp in [(1, 1, 1), (9, 7, 11), (40, 33, 29), (5, 3, 4200), (130, 2, 3)]:
        grid = pal[rng.integers(0, len(pal), shp)]
        for names in (["dome"], ["plinth", "windows", "background"], list(oracle.PART_COLORS)):
            gp, gc = pb3d_gpu.get_voxel_points_by_parts(grid, oracle.PART_COLORS, names)
            op, oc = oracle.get_voxel_points_by_parts(grid, oracle.PART_COLORS, names)
            assert np.array_equal(gp, op) and np.array_equal(gc, oc), (shp, names)
        for st in (1, 2, 3, 5):
            gp, gc, gs = pb3d_gpu.voxel_grid_to_points(grid, stride=st)
            op, oc, os_ = oracle.voxel_grid_to_points(grid, stride=st)
            assert np.array_equal(gp, op) and np.array_equal(gc, oc) and gs == os_, (shp, st)
    # all-empty and all-full
    z = np.zeros((6, 5, 4, 3), np.uint8)
    p, c, _ = pb3d_gpu.voxel_grid_to_points(z, stride=1)
    assert p.shape == (0, 3) and c.shape == (0, 3)
    p, c, _ = pb3d_gpu.voxel_grid_to_points(z + 9, stride=1)
    assert len(p) == 120 and np.array_equal(p[1], [1, 0, 0])


def test_projection_duplicates_last_writer_wins(pb3d_gpu, oracle):
    """many points on one pixel: the last in input order must win, run after run."""
    rng = np.random.default_rng(23)
    N = 200000
    pts = rng.integers(0, 8, (N, 3)).astype(np.float32)      # only 512 distinct positions -> massive collisions
    cols = rng.integers(1, 256, (N, 3), dtype=np.uint8)
    cam = np.array([4, 4, -30], np.float32); tgt = np.array([4, 4, 4], np.float32)
    want = oracle.project_colored_voxels(pts, cols, cam, tgt, 100.0, 32.0, 32.0, 64, 64)
    for _ in range(3):
        got = pb3d_gpu.project_colored_voxels(pts, cols, cam, tgt, 100.0, 32.0, 32.0, 64, 64)
        assert np.array_equal(got, want)
    # no points at all -> black image
    e = pb3d_gpu.project_colored_voxels(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8), cam, tgt, 100.0, 32.0, 32.0, 8, 8)
    assert e.shape == (8, 8, 3) and not e.any()


# ---- full-size properties (BASELINE config 4: synthetic 16-label mask, 1024^3) -----------------------
def test_full_size_carve_properties(pb3d_gpu):
    """1024^3 semantic carve (6.4 GB of traffic) checked through size-independent properties:
    idempotence, per-plane byte sums equal to the masked input sums, dropped columns all zero, and a slab
    of the volume bit-equal to NumPy's np.where on the same bytes."""
    import synth_host
    from pb3d import device as dev
    S = int(os.environ.get("PB3D_TEST_FULL_SIZE", "1024"))
    nvox = S * S * S
    d_in = dev.DeviceBuffer(nvox * 3); d_out = dev.DeviceBuffer(nvox * 3); d_out2 = dev.DeviceBuffer(nvox * 3)
    d_mwh = dev.DeviceBuffer(S * S)
    dev.synth_sem(0, S, S, S, 1, d_in)
    dev.synth_mask16(S, d_binary_wh=d_mwh)
    dev.carve_mask(d_in, S, S, S, 3, d_mwh, d_out)
    dev.carve_mask(d_out, S, S, S, 3, d_mwh, d_out2)       # idempotence
    dev.sync()
    lab, binary, rgb = synth_host.mask16(S)
    m_wh = np.ascontiguousarray(binary.T)
    assert np.array_equal(d_mwh.download((S, S)), m_wh)
    step = max(1, S // 8)
    planes = 2
    for x0 in list(range(0, S, step)) + [S - planes]:
        off = x0 * S * S * 3
        a = d_in.download((planes, S, S, 3), byte_offset=off)
        b = d_out.download((planes, S, S, 3), byte_offset=off)
        c = d_out2.download((planes, S, S, 3), byte_offset=off)
        assert np.array_equal(a, synth_host.sem_slab(x0, x0 + planes, S, S, 1))            # device generator == host formula
        want = np.where(m_wh[x0:x0 + planes, :, None, None].astype(bool), a, 0).astype(np.uint8)
        assert np.array_equal(b, want) and np.array_equal(c, b)
        assert int(b.astype(np.uint64).sum()) == int(a[m_wh[x0:x0 + planes].astype(bool)].astype(np.uint64).sum())
    for buf in (d_in, d_out, d_out2, d_mwh):
        buf.free()


def test_process90_tiled_permutation_sizes(pb3d_gpu, oracle):
    """the LDS-tiled 90-degree path (W + D even; any D, any row alignment: whole 16-byte pieces at arbitrary byte addresses,
    ragged pieces at row ends, W != D so the column offset c2 is non-zero and of either sign) on sizes that are not tile
    multiples, next to shapes that must fall back to the generic kernel (W + D odd -> half-integer coordinates)."""
    rng = np.random.default_rng(29)
    for (W, H, D) in [(100, 7, 100), (68, 5, 132), (132, 3, 68), (64, 2, 64), (4, 3, 4), (260, 4, 260), (200, 2, 72),
                      (130, 3, 62), (63, 4, 64), (65, 2, 65), (128, 3, 128), (256, 2, 256), (192, 5, 192),
                      (355, 6, 355), (123, 9, 123), (37, 5, 51), (51, 5, 37), (131, 7, 129), (1, 3, 1), (17, 4, 1), (1, 4, 17),
                      (150, 3, 200), (437, 2, 437), (15, 11, 15), (16, 3, 18),
                      # H * D % 128 == 0 with rows that are not whole lines: each x-row's stream tiled in whole lines (k_rot90_flat;
                      # segments that straddle two planes) ; knob rot90_flat = 1 keeps the row-wise tile kernel
                      (355, 128, 355), (131, 256, 131), (136, 128, 200), (200, 128, 136), (129, 384, 129), (437, 128, 437),
                      (130, 64, 130), (300, 32, 172), (141, 1024, 141)]:
        lines = D % 128 != 0 and (H * D) % 128 == 0 and D >= 128
        for kind in ("bin", "full"):
            g = (rng.random((W, H, D)) < 0.5).astype(np.uint8) if kind == "bin" else rng.integers(0, 256, (W, H, D), dtype=np.uint8)
            m = rng.random((H, W)) < 0.85
            want = oracle.process_voxel_grid(g, m, 90)
            for flat in ((0, 1) if lines else (0,)):        # 1: the row-wise tile kernel instead of the flat (stream) forms
                pb3d_gpu._lib.set_tuning("rot90_flat", flat)
                try:
                    got = pb3d_gpu.process_voxel_grid(g, m, 90)
                finally:
                    pb3d_gpu._lib.set_tuning("rot90_flat", 0)
                assert np.array_equal(got, want), (W, H, D, kind, flat, int((got != want).sum()))


def test_global_carve_slabs_device(pb3d_gpu, oracle):
    """fused global_carve: X-slabs computed independently (the sharded form) concatenate to the full result."""
    import synth_host
    from pb3d import device as dev
    for S in (64, 96, 80, 128):
        lab, binary, rgb = synth_host.mask16(S)
        want = oracle.global_carve(binary, rgb, 90)
        d_b = dev.from_numpy(binary); d_rgb = dev.from_numpy(rgb)
        full = dev.DeviceBuffer(S * S * S * 3)
        dev.global_carve(d_b, d_rgb, S, S, 90, full)
        assert np.array_equal(full.download((S, S, S, 3)), want), S
        for nr in (2, 4):
            parts = []
            for r in range(nr):
                x0, x1 = pb3d_gpu.dist.slab_bounds(S, r, nr)
                slab = dev.DeviceBuffer((x1 - x0) * S * S * 3)
                dev.global_carve(d_b, d_rgb, S, S, 90, slab, x0, x1)
                parts.append(slab.download((x1 - x0, S, S, 3)))
            assert np.array_equal(np.concatenate(parts, 0), want), (S, nr)
    # widths that are not multiples of 16 go through the byte-store variant; non-square images too
    # ... and, when the height is a multiple of 16, through the flat form (k_global_carve90f: the rows of a y-chunk as one stream of
    # aligned pieces, pieces that straddle two rows); (rounds 1-3 had a row-wise and a flat piece kernel; round 4's stream kernel has one form)
    for (h, w) in [(50, 50), (33, 70), (64, 40), (20, 144), (32, 355), (16, 37), (16, 17), (48, 131), (128, 141), (64, 16), (80, 22)]:
        lab, binary, rgb = synth_host.mask16(max(h, w))
        binary, rgb = np.ascontiguousarray(binary[:h, :w]), np.ascontiguousarray(rgb[:h, :w])
        want = oracle.global_carve(binary, rgb, 90)
        got = pb3d_gpu.global_carve(binary, rgb, 90)
        assert np.array_equal(got, want), (h, w)


def test_deformation_loop_f8(pb3d_gpu, oracle, golden):
    """BASELINE config 5: the notebook-3 deformation loop (deform -> bounds -> paint -> project -> IoU) against
    the headless drive of the reference's widget closures, and against the oracle on a random cloud."""
    g = golden("f8_deformation")
    meta = json.load(open(os.path.join(GOLDEN, "f8_deformation.json")))
    grid = np.load(os.path.join(GOLDEN, "stored_Akbar_voxel_grid.npz"))["voxel_grid"]
    PC = pb3d_gpu.PART_COLORS
    r = meta["rand"]
    got = pb3d_gpu.deform_coords(g["rand_pts"], r["image_shape"], r["voxel_shape"], r["deform"])
    assert got.dtype == np.int64 and np.array_equal(got, g["rand_coords"])
    cams = _cams("Akbar")["front"]
    saved = {}
    for part, c in meta["cases"].items():
        coords, _ = pb3d_gpu.get_voxel_points_by_parts(grid, PC, [part])
        cd = pb3d_gpu.deform_coords(coords, meta["image_shape"], meta["voxel_shape"], c["deform"])
        assert len(cd) == c["n_deformed"] and sha(cd) == c["coords_sha256"], part
        _, iou = pb3d_gpu.evaluate_part_deform(grid, PC, part, c["deform"], g["front_mask"], cams)
        assert iou == c["iou"], (part, iou, c["iou"])
        saved[part] = {"deform": c["deform"], "iou": iou}
    full = pb3d_gpu.build_deformed_grid(grid, PC, saved, meta["image_shape"])
    assert sha(full) == meta["deformed_grid_sha256"]
    rng = np.random.default_rng(31)
    for _ in range(4):
        pts = rng.integers(-5, 70, (int(rng.integers(1, 3000)), 3)).astype(np.float32)
        dv = dict(scale_y=float(rng.uniform(0.5, 2)), shift_y=float(rng.integers(-100, 100)), scale_xz=float(rng.uniform(0.5, 2)),
                  shift_xz=float(rng.integers(-100, 100)))
        a = pb3d_gpu.deform_coords(pts, (77, 131), (64, 70, 66), dv)
        b = oracle.deform_coords(pts, (77, 131), (64, 70, 66), dv)
        assert np.array_equal(a, b)
    with pytest.raises(pb3d_gpu._lib.Pb3dError, match="voxel indices"):
        pb3d_gpu.deform_coords(np.array([[0.5, 1, 2]], np.float32), (10, 10), (4, 4, 4), dict(scale_y=1.0, shift_y=0.0, scale_xz=1.0, shift_xz=0.0))


PART_SYMMETRY = {"dome": 5, "chhatris": 45, "front_minarets": 5, "small_minarets": 5}
EXTRUSION = {"main_door": 20, "windows": 10}


def test_connected_components_device(pb3d_gpu, golden, oracle):
    """device union-find labelling == scipy.ndimage.label numbering (fixture) incl. 590 salt-noise components."""
    from pb3d import device as dev
    from pb3d.voxel_carving_utils import _component_stats, _label
    g = golden("f5_label_noise")
    grid = np.ascontiguousarray(g["grid"]); A0, A1, A2, _ = grid.shape
    d_g = dev.from_numpy(grid); d_lab = dev.DeviceBuffer(A0 * A1 * A2 * 4)
    n = _label(d_g, (A0, A1, A2), np.array(pb3d_gpu.PART_COLORS["dome"], np.uint8), d_lab)
    lab = d_lab.download((A0, A1, A2), np.int32)
    assert n == int(g["n"]) and np.array_equal(lab, g["labels"])
    bbox, cnt, sums = _component_stats(d_lab, (A0, A1, A2), n)
    for i in (1, 2, n // 2, n):
        idx = np.argwhere(g["labels"] == i)
        assert np.array_equal(bbox[i - 1], np.concatenate([idx.min(0), idx.max(0) + 1])) and cnt[i - 1] == len(idx)
        assert np.array_equal(sums[i - 1], idx.sum(0))
    # one big snake-like component and an empty selection
    snake = np.zeros((40, 30, 50), bool); snake[::2] = True; snake[1::4, 0] = True; snake[3::4, -1] = True
    col = np.zeros(snake.shape + (3,), np.uint8); col[snake] = (9, 9, 9)
    d2 = dev.from_numpy(col); l2 = dev.DeviceBuffer(snake.size * 4)
    n2 = _label(d2, snake.shape, np.array((9, 9, 9), np.uint8), l2)
    want, nw = oracle.label6(snake)
    assert n2 == nw == 1 and np.array_equal(l2.download(snake.shape, np.int32), want)
    assert _label(d2, snake.shape, np.array((1, 2, 3), np.uint8), l2) == 0
    for b in (d_g, d_lab, d2, l2):
        b.free()


@pytest.mark.gpu
def test_connected_components_row_frame_shapes(pb3d_gpu, oracle):
    """The labelling works on 64-voxel windows of rows cut from a flat bit stream: rows shorter than, equal to and longer than a
    window, odd lengths (windows that straddle dwords of the stream), runs that cross windows, densities from salt noise to nearly
    full, plus the degenerate axes -- every label volume equals the oracle's scipy numbering."""
    from pb3d import device as dev
    from pb3d.voxel_carving_utils import _component_stats, _label
    rng = np.random.default_rng(77)
    col = np.array((200, 10, 30), np.uint8)
    shapes = [(1, 1, 1), (1, 1, 70), (3, 1, 64), (5, 7, 1), (4, 5, 63), (4, 5, 65), (6, 3, 128), (7, 9, 129), (9, 8, 191), (12, 10, 200),
              (33, 17, 355), (16, 16, 16), (2, 40, 31), (40, 2, 97), (21, 22, 23)]
    for shp in shapes:
        for dens in (0.05, 0.5, 0.9, 1.0):
            mask = rng.random(shp) < dens
            if dens == 0.9:                                     # long runs with a few cuts: runs cross windows, few unions per row
                mask = np.ones(shp, bool); mask[rng.random(shp) < 0.02] = False
            grid = rng.integers(0, 3, shp + (3,)).astype(np.uint8)          # other colours around the members
            grid[mask] = col
            mask = np.all(grid == col, axis=-1)
            d_g = dev.from_numpy(grid); d_lab = dev.DeviceBuffer(mask.size * 4)
            n = _label(d_g, shp, col, d_lab)
            want, nw = oracle.label6(mask)
            got = d_lab.download(shp, np.int32)
            assert n == nw and np.array_equal(got, want), (shp, dens)
            if n:
                bbox, cnt, sums = _component_stats(d_lab, shp, n)
                assert np.array_equal(cnt, np.bincount(want.ravel(), minlength=n + 1)[1:]), (shp, dens)
                # the statistics gathered by the labelling's own last pass (one round trip) == the separate pass; a capacity below the
                # component count falls back to it
                from pb3d.voxel_carving_utils import _label_stats
                for cap in (4096, 70, 2):
                    d_l2 = dev.DeviceBuffer(mask.size * 4)
                    n2, b2, c2, s2 = _label_stats(d_g, shp, col, d_l2, cap=cap)
                    assert n2 == n and np.array_equal(d_l2.download(shp, np.int32), want), (shp, dens, cap)
                    assert np.array_equal(b2, bbox) and np.array_equal(c2, cnt) and np.array_equal(s2, sums), (shp, dens, cap)
                    d_l2.free()
                # members_only: the zeros of the non-members are not written (poisoned buffer), the members' labels and the statistics are the same
                d_l3 = dev.from_numpy(np.full(shp, -7, np.int32))
                n3, b3, c3, s3 = _label_stats(d_g, shp, col, d_l3, cap=4096, members_only=True)
                got3 = d_l3.download(shp, np.int32)
                # (more components than the capacity: the call falls back to a FULL labelling -- zeros outside)
                # (the entries of non-members are unspecified: untouched, or zero where a 16-byte store of four voxels holds a member)
                assert n3 == n and np.array_equal(got3[mask], want[mask]) and np.all(np.isin(got3[~mask], (-7, 0) if n <= 4096 else (0,))), (shp, dens)
                assert np.array_equal(b3, bbox) and np.array_equal(c3, cnt) and np.array_equal(s3, sums), (shp, dens)
                d_l3.free()
            d_g.free(); d_lab.free()


@pytest.mark.gpu
def test_connected_components_several_colours_one_pass(pb3d_gpu, oracle):
    """N colours in ONE labelling sequence (pb3d_label_colors_stats_dev) == N single-colour calls == the oracle's scipy numbering, per
    colour: labels at the colour's voxels, component counts, boxes, voxel counts and coordinate sums.  Runs of different colours abut
    without a gap (up to 64 segments per window), rows shorter / longer than a window, 1..8 colours, RGB grids and 1-byte label volumes."""
    from pb3d import device as dev
    from pb3d.voxel_carving_utils import _label_stats, _label_stats_multi
    rng = np.random.default_rng(404)
    pal = np.array([(200, 10, 30), (1, 220, 5), (63, 138, 173), (190, 0, 255), (0, 0, 255), (5, 223, 223), (255, 180, 80), (180, 140, 255),
                    (255, 120, 230), (9, 9, 9)], np.uint8)
    cases = [((1, 1, 1), 1, 1.0), ((4, 5, 63), 2, 0.5), ((6, 3, 128), 3, 0.9), ((7, 9, 129), 4, 0.3), ((9, 8, 191), 5, 1.0), ((12, 10, 200), 8, 0.7),
             ((33, 17, 355), 4, 0.97), ((16, 16, 16), 8, 1.0), ((21, 22, 23), 2, 0.05), ((40, 2, 97), 7, 0.6), ((24, 20, 300), 4, 0.995)]
    for shp, K, dens in cases:
        for form in ("rgb", "label"):
            # blocky colour field (runs of the same colour cross windows) with noise on top
            idx = rng.integers(0, len(pal), tuple(max(1, (a + 5) // 6) for a in shp))
            idx = idx.repeat(6, 0).repeat(6, 1).repeat(6, 2)[:shp[0], :shp[1], :shp[2]]
            noise = rng.random(shp) > dens
            idx[noise] = rng.integers(0, len(pal), int(noise.sum()))
            if form == "rgb":
                grid = np.ascontiguousarray(pal[idx]); cols = [pal[k] for k in range(K)]
                masks = [np.all(grid == pal[k], axis=-1) for k in range(K)]
            else:
                grid = np.ascontiguousarray((idx + 1).astype(np.uint8)); cols = [int(k + 1) for k in range(K)]
                masks = [grid == k + 1 for k in range(K)]
            d_g = dev.from_numpy(grid)
            d_lab = dev.from_numpy(np.full(shp, -7, np.int32))
            res = _label_stats_multi(d_g, shp, cols, d_lab, cap=2048, members_only=True)
            got = d_lab.download(shp, np.int32)
            anym = np.zeros(shp, bool)
            for k in range(K):
                want, nw = oracle.label6(masks[k])
                anym |= masks[k]
                assert res[k] is not None and res[k][0] == nw, (shp, K, form, k)
                assert np.array_equal(got[masks[k]], want[masks[k]]), (shp, K, form, k)
                dev_l1 = dev.DeviceBuffer(int(np.prod(shp)) * 4)
                n1, b1, c1, s1 = _label_stats(d_g, shp, cols[k] if form == "label" else np.ascontiguousarray(cols[k]), dev_l1, cap=2048)
                assert n1 == nw and np.array_equal(dev_l1.download(shp, np.int32), want), (shp, K, form, k)
                assert np.array_equal(res[k][1], b1) and np.array_equal(res[k][2], c1) and np.array_equal(res[k][3], s1), (shp, K, form, k)
                assert np.array_equal(c1, np.bincount(want.ravel(), minlength=nw + 1)[1:]), (shp, K, form, k)
                dev_l1.free()
            assert np.all(np.isin(got[~anym], (-7, 0))), (shp, K, form)
            # a full label volume of several colours: zeros wherever no requested colour sits
            d_l2 = dev.from_numpy(np.full(shp, -7, np.int32))
            res2 = _label_stats_multi(d_g, shp, cols, d_l2, cap=2048, members_only=False)
            got2 = d_l2.download(shp, np.int32)
            assert np.all(got2[~anym] == 0) and np.array_equal(got2[anym], got[anym]) and [r[0] for r in res2] == [r[0] for r in res], (shp, K, form)
            # a capacity below a colour's component count: that colour comes back as None, the others stay valid
            if max(r[0] for r in res) > 3:
                res3 = _label_stats_multi(d_g, shp, cols, d_l2, cap=3, members_only=True)
                for k in range(K):
                    assert (res3[k] is None) == (res[k][0] > 3), (shp, K, form, k)
                    if res3[k] is not None:
                        assert np.array_equal(res3[k][1], res[k][1]) and np.array_equal(res3[k][2], res[k][2])
            for b in (d_g, d_lab, d_l2):
                b.free()


def test_guided_carve_fused_component_loop(pb3d_gpu, oracle):
    """N1 as one launch sequence (csrc/guided.hip): every component's 32-plane slices stay in LDS for all rotation steps, the grid is
    carved in place.  Against the oracle incl. the printed log: interlocking components whose boxes OVERLAP (a later crop sees -- and
    may restore -- an earlier component's voxels: batches + a copy), more components than one batch holds, tall crops (several plane
    groups), the angle steps of the notebook (5, 45) and an empty rotation loop (angle > 90)."""
    import contextlib
    import io
    rng = np.random.default_rng(77)
    col = np.array(pb3d_gpu.PART_COLORS["dome"], np.uint8)
    other = np.array(pb3d_gpu.PART_COLORS["plinth"], np.uint8)
    cases = []
    # (a) two interlocking L shapes + a far blob, boxes overlap
    W, H, D = 40, 70, 36
    g = np.zeros((W, H, D, 3), np.uint8)
    g[2:30, 3:60, 4:8] = col; g[2:6, 3:60, 4:30] = col                       # L one: box x 2..30, z 4..30
    g[10:28, 5:66, 12:28] = col                                              # block two, inside L one's box, not touching it
    g[12:20, 20:40, 14:20] = other                                           # foreign colour inside block two
    g[36:39, 10:20, 30:35] = col
    g[rng.random((W, H, D)) < 0.03] = 0
    cases.append((g, W, H, D))
    # (b) many small components (more than one batch), random boxes that overlap a lot
    W, H, D = 48, 40, 48
    g = np.zeros((W, H, D, 3), np.uint8)
    for _ in range(90):
        lo = [int(rng.integers(0, s - 2)) for s in (W, H, D)]
        hi = [min(s, l + int(rng.integers(1, 7))) for s, l in zip((W, H, D), lo)]
        g[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = col if rng.random() < 0.8 else other
    cases.append((g, W, H, D))
    # (c) one solid component, odd sizes
    W, H, D = 33, 45, 31
    g = np.zeros((W, H, D, 3), np.uint8); g[3:30, 2:44, 1:29] = col; g[rng.random((W, H, D)) < 0.1] = other
    cases.append((g, W, H, D))
    for (g, W, H, D) in cases:
        sem = np.zeros((H, W, 3), np.uint8)
        sem[rng.random((H, W)) < 0.85] = col
        for angle in (5, 45, 60, 120):
            b1, b2 = io.StringIO(), io.StringIO()
            with contextlib.redirect_stdout(b1):
                got = pb3d_gpu.left_right_guided_carve(g, sem, col, angle=angle)
            with contextlib.redirect_stdout(b2):
                want = oracle.left_right_guided_carve(g, sem, col, angle=angle)
            assert np.array_equal(got, want), (W, H, D, angle, int((got != want).any(-1).sum()))
            assert b1.getvalue() == b2.getvalue(), (W, H, D, angle)


@pytest.mark.parametrize("name", ["Taj_96", "Akbar_64", "Bibi_80"])
def test_partwise_stages_f5(pb3d_gpu, golden, name):
    """N1/N2: component-guided carve (incl. its printed log), extrusion, recolouring and the whole partwise_carve
    against the reference's stage outputs."""
    import contextlib
    import io
    g = golden(f"f5_{name}")
    meta = json.load(open(os.path.join(GOLDEN, "f5_meta.json")))[name]
    PCN = pb3d_gpu.PART_COLORS_NP
    gc = pb3d_gpu.global_carve(g["binary"], g["ext"], 90)
    pc = pb3d_gpu.part_carve(gc, g["ext"], JOBS_NB1)
    grid = pc
    for (part, angle), want_log in zip(PART_SYMMETRY.items(), meta["lrgc_stdout"]):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            grid = pb3d_gpu.left_right_guided_carve(grid, g["ext"], PCN[part], angle=angle)
        assert sha(grid) == meta["stages"][f"lrgc_{part}"], part
        assert buf.getvalue() == want_log
    for part, depth in EXTRUSION.items():
        mk = np.all(g["sem"] == PCN[part], axis=-1)
        for ax, dr in ((2, "+"), (2, "-"), (0, "+"), (0, "-")):
            grid = pb3d_gpu.extrude_from_surface(grid, mk, axis=ax, direction=dr, depth=depth, fill_color=PCN[part])
            assert sha(grid) == meta["stages"][f"extrude_{part}_{ax}{dr}"], (part, ax, dr)
    oriented = np.flip(grid.transpose(2, 1, 0, 3), axis=1)
    rec = pb3d_gpu.recolor_backward_components(oriented, PCN["front_minarets"], new_color=PCN["back_minarets"], k=2, sort_axis=0)
    assert np.array_equal(rec, g["after_recolor"]) and rec.flags["C_CONTIGUOUS"]
    with contextlib.redirect_stdout(io.StringIO()):
        full = pb3d_gpu.partwise_carve(gc, g["ext"], g["sem"], PCN, JOBS_NB1, PART_SYMMETRY, EXTRUSION)
    assert sha(full) == meta["partwise_sha256"] and list(full.shape) == meta["partwise_shape"]
    assert np.array_equal(pb3d_gpu.extrude_from_surface(pc, np.all(g["sem"] == PCN["full_building"], axis=-1), 2, "-", 3, None), g["extrude_none"])
    assert np.array_equal(pb3d_gpu.recolor_backward_components(pc, PCN["front_minarets"], PCN["windows"], k=1, sort_axis=2), g["recolor_k1_axis2"])


def test_rccl_allgather_single_rank(pb3d_gpu):
    """the RCCL path (dlopen, unique id, communicator, ncclAllGather on the context stream) with a 1-rank communicator:
    the in-place gather of the only slab must leave the volume untouched and a separate send buffer must be copied."""
    from pb3d import device as dev, dist
    rng = np.random.default_rng(37)
    slab = rng.integers(0, 256, (8, 16, 16, 3), dtype=np.uint8)
    d_full = dev.from_numpy(slab)
    uid = dist.new_unique_id()
    assert uid.shape == (128,) and uid.any()
    dist.comm_init(uid, 0, 1)
    try:
        dist.allgather(d_full, d_full, slab.nbytes)           # in place
        dev.sync()
        assert np.array_equal(d_full.download(slab.shape), slab)
        d_dst = dev.DeviceBuffer(slab.nbytes); d_dst.zero()
        dist.allgather(d_full, d_dst, slab.nbytes)
        dev.sync()
        assert np.array_equal(d_dst.download(slab.shape), slab)
    finally:
        dist.comm_destroy()


@pytest.mark.parametrize("mon", ["Akbar", "Charminar"])
def test_camera_objective_and_zbuffer_n45(pb3d_gpu, mon):
    """rows N4 / N5: the resident camera objective over perturbed cameras (float64 parameters, as the aligner's
    sliders / optimisers produce them) and the z-buffer visibility functions, against the reference's values."""
    g = np.load(os.path.join(GOLDEN, "n45_objective_zbuffer.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "n45_objective_zbuffer.json")))
    grid = np.load(os.path.join(GOLDEN, f"stored_{mon}_voxel_grid.npz"))["voxel_grid"]
    PC = pb3d_gpu.PART_COLORS
    m = meta[f"objective_{mon}"]
    front = np.load(os.path.join(GOLDEN, "f7_projection.npz"))[f"img_{mon}_front"]
    seg = pb3d_gpu.mask_parts_from_image(front, PC, m["parts"])
    assert np.array_equal(seg, g[f"seg_{mon}"])
    pts, cols = pb3d_gpu.get_voxel_points_by_parts(grid, PC, m["parts"])
    obj = pb3d_gpu.CameraObjective(pts, cols, seg, {p: PC[p] for p in m["parts"]})
    params = [{"cam_pos": np.array(t["cam_pos"]), "target": np.array(t["target"]), "f": t["f"], "cx": t["cx"], "cy": t["cy"],
               "H": m["H"], "W": m["W"]} for t in m["trials"]]
    got = obj.evaluate_batch(params)
    assert got == [t["neg_iou"] for t in m["trials"]]
    assert obj(params[0]) == m["trials"][0]["neg_iou"]          # re-evaluation on the resident cloud is stable
    obj.close()
    cams = _cams(mon)["front"]
    for mode in ("f32", "f64"):
        if f"zbuf_{mon}_{mode}" not in g.files:
            continue
        cam = {k: (v.astype(np.float64) if (mode == "f64" and isinstance(v, np.ndarray)) else v) for k, v in cams.items()}
        zbuf = pb3d_gpu.compute_global_depth_buffer(grid, cam, m["H"], m["W"])
        assert zbuf.dtype == np.float32 and np.array_equal(zbuf, g[f"zbuf_{mon}_{mode}"])
        ppts, _ = pb3d_gpu.get_voxel_points_by_parts(grid, PC, ["front_minarets"])
        vis = pb3d_gpu.project_part_visible(ppts, cam, zbuf, m["H"], m["W"])
        assert vis.dtype == bool and np.array_equal(vis, g[f"vis_{mon}_{mode}"])


def test_sharded_projection_keys(pb3d_gpu, oracle):
    """points partition: shards projected independently into key images, merged by max (what the RCCL
    all-reduce does), resolve == the unsharded projection; and the same through RCCL on a 1-rank communicator."""
    from pb3d import dist
    rng = np.random.default_rng(41)
    N = 150000
    pts = rng.integers(0, 24, (N, 3)).astype(np.float32)           # many collisions
    cols = rng.integers(0, 256, (N, 3), dtype=np.uint8)
    cam = np.array([12, 10, -60], np.float32); tgt = np.array([12, 12, 12], np.float32)
    args = (cam, tgt, 150.0, 48.0, 40.0, 80, 96)
    want = oracle.project_colored_voxels(pts, cols, *args)
    assert np.array_equal(pb3d_gpu.project_colored_voxels(pts, cols, *args), want)
    for nr in (2, 3, 8):
        merged = np.zeros((80, 96), np.uint64)
        for r in range(nr):
            i0, i1 = dist.point_shard_bounds(N, r, nr)
            img_r, keys_r = dist.project_colored_voxels_sharded(pts[i0:i1], cols[i0:i1], i0, *args, reduce=False)
            merged = np.maximum(merged, keys_r)
        assert np.array_equal(dist.resolve_keys(merged), want), nr
    dist.comm_init(dist.new_unique_id(), 0, 1)
    try:
        assert np.array_equal(dist.project_colored_voxels_sharded(pts, cols, 0, *args), want)
    finally:
        dist.comm_destroy()


def test_generic_angle_steps_large(pb3d_gpu, oracle):
    """generic-angle steps on larger grids: 0/1 data runs bit-sliced (csrc/sliced.hip; since round 4 also a SINGLE step, angle step 50),
    0..255 data -- or one voxel with a value > 1 -- must raise the slice pass's flag and be redone by the arithmetic kernel; both forms
    (and the pinned byte chain, tune sliced = 1) bit-exact.  Shapes: rows at arbitrary byte alignment, ragged row ends, fewer than 32
    planes, D % 16 == 0 but D % 32 != 0, two x-tiles."""
    rng = np.random.default_rng(43)
    for (W, H, D) in [(160, 90, 160), (176, 64, 192), (400, 20, 272), (200, 60, 180), (131, 128, 130), (133, 121, 129), (300, 20, 357),
                      (271, 33, 240), (355, 16, 355), (290, 40, 333)]:
        m = rng.random((H, W)) < 0.9
        g_bin = (rng.random((W, H, D)) < 0.5).astype(np.uint8)
        g_full = rng.integers(0, 256, (W, H, D), dtype=np.uint8)
        g_one_big = g_bin.copy(); g_one_big[W // 2, H // 2, D // 2] = 200      # a single value > 1 must switch the whole call
        for ai in (45, 30, 50):
            for g in (g_bin, g_full, g_one_big):
                want = oracle.process_voxel_grid(g, m, ai)
                for sliced in (0, 1):
                    pb3d_gpu._lib.set_tuning("sliced", sliced)
                    try:
                        got = pb3d_gpu.process_voxel_grid(g, m, ai)
                    finally:
                        pb3d_gpu._lib.set_tuning("sliced", 0)
                    assert np.array_equal(got, want), (W, H, D, ai, sliced, int((got != want).sum()))


def test_sliced_chain_equals_byte_chain_equals_oracle(pb3d_gpu, oracle):
    """chains of rotation steps stay bit-sliced between the steps (csrc/sliced.hip: 32 planes per dword, 1/4 B/voxel per middle
    step): sliced chain == byte chain == oracle on 0/1 grids, on 0..255 grids (the slice kernel's flag sends those to the byte
    chain) and on odd-sized grids; masks of every density; single-step calls with the sliced form forced."""
    rng = np.random.default_rng(31)
    shapes = [(64, 40, 64), (37, 11, 37), (130, 70, 131), (96, 33, 72), (200, 64, 256), (3, 5, 2), (1, 1, 1), (65, 32, 9), (16, 95, 257)]
    for (W, H, D) in shapes:
        for ai in (5, 45, 30, 7, 18):
            for kind in ("binary", "bytes", "dense"):
                if kind == "binary":
                    g = (rng.random((W, H, D)) < 0.6).astype(np.uint8)
                elif kind == "dense":
                    g = np.ones((W, H, D), np.uint8)
                else:
                    g = rng.integers(0, 256, (W, H, D), dtype=np.uint8)
                m = rng.random((H, W)) < (0.9 if kind != "dense" else 0.97)
                want = oracle.process_voxel_grid(g, m, ai)
                for sliced in (0, 1):
                    pb3d_gpu._lib.set_tuning("sliced", sliced)
                    try:
                        got = pb3d_gpu.process_voxel_grid(g, m, ai)
                    finally:
                        pb3d_gpu._lib.set_tuning("sliced", 0)
                    assert np.array_equal(got, want), (W, H, D, ai, kind, sliced, int((got != want).sum()))
    # one rotation step: 60 degrees and 90 degrees with W + D odd run through the sliced kernels, 90 degrees with W + D even on the
    # permutation kernels; rotate_carve with the caller's own matrix (no 0-degree carve in front, with and without a mask)
    import ctypes as C
    from pb3d import device as dev
    L = pb3d_gpu._lib
    for (W, H, D) in [(64, 40, 64), (37, 11, 38), (130, 70, 131), (128, 33, 128)]:
        g = (rng.random((W, H, D)) < 0.5).astype(np.uint8)
        m = rng.random((H, W)) < 0.9
        for ai in (90, 60):
            want = oracle.process_voxel_grid(g, m, ai)
            for sliced in (0, 1):
                L.set_tuning("sliced", sliced)
                try:
                    got = pb3d_gpu.process_voxel_grid(g, m, ai)
                finally:
                    L.set_tuning("sliced", 0)
                assert np.array_equal(got, want), (W, H, D, ai, sliced, int((got != want).sum()))
        d_g = dev.from_numpy(g); d_m = dev.from_numpy(np.ascontiguousarray(m.T).astype(np.uint8)); d_o = dev.DeviceBuffer(g.size)
        for ang in (33, 60, 90):
            M = pb3d_gpu.voxel_carving_utils._rotation_matrix_inv(ang)
            off = np.zeros(3); shape = (C.c_int64 * 3)(W, H, D)
            L.check(L.load().pb3d_offset(L.p_dbl(M), shape, L.p_dbl(off)))
            res = {}
            for masked in (True, False):
                for sliced in (0, 1):
                    L.set_tuning("sliced", sliced)
                    try:
                        dev.rotate_carve(d_g, W, H, D, M, off, d_m if masked else None, d_o)
                        res[(masked, sliced)] = d_o.download((W, H, D))
                    finally:
                        L.set_tuning("sliced", 0)
                assert np.array_equal(res[(masked, 0)], res[(masked, 1)]), (W, H, D, ang, masked)
            assert np.array_equal(res[(True, 0)], oracle.carve_voxel_grid_with_masks(res[(False, 0)], m)), (W, H, D, ang)
        for b in (d_g, d_m, d_o):
            b.free()


def test_odd_rows_with_dirty_slack(pb3d_gpu, oracle):
    """rows that are not multiples of 16 bytes: the slice pass reads 16-byte pieces at whatever alignment the rows have and the last piece of
    a row byte-wise -- whatever lies behind the volume in its allocation (here: 0xff) must neither change a voxel nor trip the 'value > 1'
    check."""
    from pb3d import device as dev
    rng = np.random.default_rng(8)
    for (W, H, D) in [(300, 16, 357), (355, 9, 355), (271, 40, 333)]:
        g = (rng.random((W, H, D)) < 0.5).astype(np.uint8)
        m = rng.random((H, W)) < 0.9
        n = g.size
        dirty = np.full(n + 64, 0xff, np.uint8); dirty[:n] = g.ravel()
        d_in = dev.from_numpy(dirty); d_m = dev.from_numpy(np.ascontiguousarray(m.T).astype(np.uint8))
        d_o = dev.DeviceBuffer(n); d_t = dev.DeviceBuffer(n)
        for ai in (45, 60):
            want = oracle.process_voxel_grid(g, m, ai)
            dev.process_grid(d_in, W, H, D, d_m, ai, d_o, d_t)
            got = d_o.download((W, H, D))
            assert np.array_equal(got, want), (W, H, D, ai, int((got != want).sum()))
        for b in (d_in, d_m, d_o, d_t):
            b.free()


def test_random_shapes_property(pb3d_gpu, oracle):
    """seeded random sweep over shapes / angle steps / mask densities / value ranges: HIP == oracle, byte for byte."""
    rng = np.random.default_rng(20261004)
    pal = np.array(list(oracle.PART_COLORS.values()), np.uint8)
    for trial in range(60):
        W, H, D = (int(v) for v in rng.integers(1, 90, 3))
        if trial % 3 == 0:
            D = W                                   # cubic in XZ -> the 90-degree permutation paths
        if trial % 6 == 0:
            W = D = int(rng.choice([16, 32, 48, 64, 80, 128]))
        ai = int(rng.choice([90, 90, 90, 60, 45, 30, 15, 10, 120]))
        g = (rng.random((W, H, D)) < rng.uniform(0.1, 0.9)).astype(np.uint8) if trial % 4 else rng.integers(0, 256, (W, H, D), dtype=np.uint8)
        m = rng.random((H, W)) < rng.uniform(0.3, 1.0)
        assert np.array_equal(pb3d_gpu.process_voxel_grid(g, m, ai), oracle.process_voxel_grid(g, m, ai)), (trial, W, H, D, ai)
        col = pal[rng.integers(0, 10, (W, H, D))] * (rng.random((W, H, D, 1)) < 0.7).astype(np.uint8)
        assert np.array_equal(pb3d_gpu.carve_voxel_grid_with_masks(col, m), oracle.carve_voxel_grid_with_masks(col, m)), (trial, "carve")
        if trial % 5 == 0:
            sem = pal[rng.integers(0, 10, (H // 3 + 1, W // 3 + 1))].repeat(3, 0).repeat(3, 1)[:H, :W]
            jobs = [(["full_building", "dome"], 90), (["plinth"], int(rng.choice([90, 45]))), (["chhatris", "windows"], 90)]
            assert np.array_equal(pb3d_gpu.part_carve(col, sem, jobs), oracle.part_carve(col, sem, jobs)), (trial, "part_carve", W, H, D)
            binary = (~np.all(sem == pal[9], axis=-1)).astype(np.uint8)
            assert np.array_equal(pb3d_gpu.global_carve(binary, sem, 90), oracle.global_carve(binary, sem, 90)), (trial, "global_carve", H, W)


def test_reprojection_iou_config3(pb3d_gpu):
    """BASELINE config 3 as the reference implements it: Charminar stored grid, front and aerial stored cameras,
    per-part points -> projection -> IoU, i.e. the numbers visualize_voxel_projection_iou puts in its plot titles."""
    summ = json.load(open(os.path.join(GOLDEN, "f7_projection_summary.json")))
    g = np.load(os.path.join(GOLDEN, "f7_projection.npz"))
    grid = np.load(os.path.join(GOLDEN, "stored_Charminar_voxel_grid.npz"))["voxel_grid"]
    cams = _cams("Charminar")
    for view in ("front", "drone"):
        per, combined = pb3d_gpu.projection_iou_by_part(grid, pb3d_gpu.PART_COLORS, g[f"img_Charminar_{view}"], cams[view])
        for part in ("full_building", "front_minarets", "back_minarets"):
            assert float(per[part]) == summ[f"Charminar_{view}_part_{part}"], (view, part)
        assert 0.0 < combined <= 1.0


@pytest.mark.gpu
def test_projection_f32_fast_path_adversarial(pb3d_gpu, oracle):
    """the all-float32 kernels (direct float32 arithmetic, 4 points per lane) against the oracle's generic evaluation:
    non-integer coordinates over 60 binades incl. float32 denormals, points behind / on the camera plane (Z clamp),
    quotients that overflow, N not a multiple of 4, and every N in 0..9 (ragged tail of the vector loop)."""
    rng = np.random.default_rng(77)
    N = 100003
    mag = np.exp2(rng.uniform(-30, 30, (N, 3))).astype(np.float32)
    pts = (mag * rng.choice([-1, 1], (N, 3))).astype(np.float32)
    pts[:2000] = rng.uniform(-40, 40, (2000, 3)).astype(np.float32)             # a dense cluster that lands in the image
    pts[2000:2100] = np.float32(1e-42) * rng.integers(-50, 50, (100, 3)).astype(np.float32)   # denormals
    pts[2100:2200, 2] = np.float32(-30.0)                                        # exactly on the camera plane: Z == 0 -> clamp
    pts[2200:2300] = rng.uniform(-3e38, 3e38, (100, 3)).astype(np.float32)       # overflow in the FMA chain -> inf/NaN
    cols = rng.integers(1, 256, (N, 3), dtype=np.uint8)
    cam = np.array([1.5, -2.25, -30], np.float32); tgt = np.array([0.5, 0.25, 4], np.float32)
    args = (cam, tgt, 123.456, 63.5, 47.25, 96, 128)
    with np.errstate(all="ignore"):
        want = oracle.project_colored_voxels(pts, cols, *args)
        assert want.any()
        assert np.array_equal(pb3d_gpu.project_colored_voxels(pts, cols, *args), want)
        for n in range(10):
            sub = slice(1000, 1000 + n)
            assert np.array_equal(pb3d_gpu.project_colored_voxels(pts[sub], cols[sub], *args),
                                  oracle.project_colored_voxels(pts[sub], cols[sub], *args)), n
        # the same cloud with a float64 camera goes through the generic kernel: both must agree with the oracle
        args64 = (cam.astype(np.float64), tgt.astype(np.float64)) + args[2:]
        assert np.array_equal(pb3d_gpu.project_colored_voxels(pts, cols, *args64), oracle.project_colored_voxels(pts, cols, *args64))


@pytest.mark.gpu
def test_points_colour_sets_hash_probe(pb3d_gpu, oracle):
    """the stride-1 compaction selects by one perfect-hash probe per voxel: random colour sets of 1..32 entries (with black,
    repeated entries, colours differing in one bit / one channel, voxel colours that hash like a member but are not one),
    dims that are powers of two, primes, 1, and longer than a block along the fast axis."""
    rng = np.random.default_rng(91)
    for trial, shp in enumerate([(3, 5, 4099), (64, 64, 64), (1, 1, 70000), (31, 1, 257), (7, 129, 33), (2, 4096, 3), (17, 16, 1024)]):
        ncol = [1, 2, 7, 13, 32, 31, 10][trial]
        members = rng.integers(0, 256, (ncol, 3), dtype=np.uint8)
        if trial % 2 == 0:
            members[0] = 0                                               # black is a member
        if ncol > 3:
            members[1] = members[2]                                      # a repeated entry
            members[3] = members[2] ^ np.array([0, 0, 1], np.uint8)      # one bit away
        others = np.concatenate([members ^ np.array([1, 0, 0], np.uint8), members[:, ::-1], rng.integers(0, 256, (40, 3), dtype=np.uint8),
                                 np.zeros((1, 3), np.uint8)])
        pal = np.concatenate([members, others])
        grid = pal[rng.integers(0, len(pal), shp)]
        pc = {f"p{k}": tuple(int(v) for v in members[k]) for k in range(ncol)}
        gp, gc = pb3d_gpu.get_voxel_points_by_parts(grid, pc, list(pc))
        op, oc = oracle.get_voxel_points_by_parts(grid, pc, list(pc))
        assert gp.dtype == np.float32 and np.array_equal(gp, op) and np.array_equal(gc, oc), (shp, ncol)
        gp, gc, _ = pb3d_gpu.voxel_grid_to_points(grid, stride=1)
        op, oc, _ = oracle.voxel_grid_to_points(grid, stride=1)
        assert np.array_equal(gp, op) and np.array_equal(gc, oc), shp


@pytest.mark.gpu
def test_points_wave_fill_and_block_fill(pb3d_gpu, oracle):
    """the fill pass of the count -> size -> fill protocol in its wave-private form (k_points_fillw, round 4) and in the block form it
    replaced (knob points_fill = 1), both against the oracle: rows shorter than a lane's four voxels (several row ends inside one
    lane), sizes that are not multiples of 4 / 16 / 1024 / 4096, densities from a handful of points per wave (head and tail bytes only)
    to full, RGB grids and 1-byte label volumes."""
    from pb3d import labels as L
    rng = np.random.default_rng(404)
    PC = oracle.PART_COLORS
    names = list(PC)
    pal = np.array([PC[n] for n in names], np.uint8)
    palette = L.Palette.from_part_colors(PC)
    shapes = [(1, 1, 1), (2, 3, 1), (7, 5, 2), (129, 2, 3), (5, 1000, 1), (3, 5, 4099), (1, 1, 70001), (64, 64, 64), (31, 7, 130), (9, 1023, 5), (2, 2, 8193)]
    for shp in shapes:
        for dens in (0.003, 0.35, 1.0):
            ids = rng.integers(0, len(pal), shp)
            grid = pal[ids] * (rng.random(shp) < dens)[..., None].astype(np.uint8)
            sel = [names[k] for k in rng.choice(len(names), size=int(rng.integers(1, len(names))), replace=False)]
            op, oc = oracle.get_voxel_points_by_parts(grid, PC, sel)
            ap, ac, _ = oracle.voxel_grid_to_points(grid, stride=1)
            lab = L.rgb_to_label(grid, palette)
            for knob in (0, 1):
                pb3d_gpu._lib.set_tuning("points_fill", knob)
                try:
                    gp, gc = pb3d_gpu.get_voxel_points_by_parts(grid, PC, sel)
                    assert np.array_equal(gp, op) and np.array_equal(gc, oc), (shp, dens, knob, "parts")
                    gp, gc, _ = pb3d_gpu.voxel_grid_to_points(grid, stride=1)
                    assert np.array_equal(gp, ap) and np.array_equal(gc, ac), (shp, dens, knob, "all")
                    gp, gc = L.get_voxel_points_by_parts_labels(lab, palette, sel)
                    assert np.array_equal(gp, op) and np.array_equal(gc, oc), (shp, dens, knob, "labels")
                finally:
                    pb3d_gpu._lib.set_tuning("points_fill", 0)


@pytest.mark.gpu
def test_part_carve_odd_shapes_w_ne_d(pb3d_gpu, oracle):
    """the fused 90-degree part_carve kernel on shapes the reference's own grids never have (W != D, odd D, non-zero column
    offset of either sign, rows at arbitrary byte alignment), foreign colours included; mixed-angle jobs beside it."""
    rng = np.random.default_rng(101)
    PC = oracle.PART_COLORS
    names = ["full_building", "chhatris", "plinth", "front_minarets", "small_minarets", "dome"]
    pal = np.array([PC[n] for n in names] + [(0, 0, 0), (9, 9, 9)], np.uint8)
    for (W, H, D) in [(37, 9, 51), (51, 5, 37), (130, 6, 62), (355, 4, 355), (129, 3, 131), (16, 7, 48), (200, 3, 72),
                      # H * D % 128 == 0 with rows that are not whole lines: the flat form (k_part90_flat: segments of each x-row's (y, z)
                      # stream, runs that straddle two planes); (round 4: k_part90_plane takes every shape)
                      (355, 128, 355), (131, 256, 131), (136, 128, 200), (200, 128, 136), (141, 384, 141), (300, 32, 172)]:
        sem = pal[rng.integers(0, 7, (H, W))]                              # (H,W,3) semantic mask: part colours + black
        colored = pal[rng.integers(0, len(pal), (W, H, D))]                # (W,H,D,3): part colours, black, a foreign colour
        colored[rng.random((W, H, D)) < 0.3] = 0
        lines = D % 128 != 0 and (H * D) % 128 == 0 and D >= 128
        for jobs in (JOBS_NB1, JOBS_MIXED):
            want = oracle.part_carve(colored, sem, jobs)
            got = pb3d_gpu.part_carve(colored, sem, jobs)
            assert np.array_equal(got, want), (W, H, D, len(jobs), int((got != want).sum()))
            if jobs is JOBS_MIXED:       # the jobs with other angles merged in one pass (default where the volume is whole 16-voxel groups) / job by job
                pb3d_gpu._lib.set_tuning("per_job", 1)
                try:
                    got = pb3d_gpu.part_carve(colored, sem, jobs)
                finally:
                    pb3d_gpu._lib.set_tuning("per_job", 0)
                assert np.array_equal(got, want), (W, H, D, "job by job", int((got != want).sum()))


@pytest.mark.gpu
def test_named_knobs(pb3d_gpu):
    """development knobs have names (include/pb3d.h: pb3d_set_tuning): an unknown name and a value out of range are refused with a
    message that says what the knob takes; PB3D_KNOBS is parsed by pb3d_create (a bad entry fails the creation loudly)."""
    import subprocess, sys
    L = pb3d_gpu._lib
    for name, bad in (("sliced", 2), ("rot90_flat", 3), ("per_job", -1), ("points_fill", 5)):
        with pytest.raises(ValueError, match=name):
            L.set_tuning(name, bad)
        L.set_tuning(name, 0)
    for name in ("misc0", "misc3", "no_such_knob"):
        with pytest.raises(ValueError, match="unknown knob"):
            L.set_tuning(name, 1)
    code = "import sys; sys.path.insert(0, %r); import pb3d; pb3d._lib.ctx(); print('created')" % PKG
    env = dict(os.environ, PB3D_KNOBS="sliced=1,per_job=1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "created" in r.stdout, r.stderr[-400:]
    env = dict(os.environ, PB3D_KNOBS="sliced=1,bogus=3")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "unknown knob" in (r.stderr + r.stdout), (r.returncode, r.stderr[-400:])


@pytest.mark.gpu
def test_extrude_from_surface_random_shapes(pb3d_gpu, oracle):
    """extrude_from_surface on seeded grids against the oracle: both axes and directions, widths from 1 to beyond eight 16-plane rounds of
    the x scan (k_extrude_x, round 4: lanes along z, rounds of doubling length split over eight waves), masks from a few pixels to full
    (the sparse and the dense regime of the scan), columns that hold nothing, a surface on the very first and the very last plane,
    depths that run out of the grid, clearing (fill_color None), RGB and label volumes."""
    rng = np.random.default_rng(909)
    PC = oracle.PART_COLORS
    names = list(PC)[:6]
    lpal = pb3d_gpu.Palette([PC[n] for n in names], names)
    tab = lpal.table()
    for (W, H) in [(1, 1), (7, 3), (16, 5), (17, 4), (64, 9), (100, 6), (130, 3), (300, 2), (65, 70)]:
        D = W                                                              # (axis 0 indexes the (H, W) mask with z: upstream needs D == W)
        for occ_dens, mask_dens in ((0.02, 0.05), (0.3, 0.5), (0.9, 1.0), (0.0, 0.7)):
            ids = (rng.integers(1, 7, (W, H, D)) * (rng.random((W, H, D)) < occ_dens)).astype(np.uint8)
            if W > 2 and occ_dens > 0:
                ids[0, :, :] = 3; ids[-1, :, 0] = 2; ids[:, :, -1] = 5          # surfaces on the first / last planes of both scans
                ids[W // 2] *= (rng.random((H, D)) < 0.5)
            grid = tab[ids]
            m2 = rng.random((H, W)) < mask_dens
            for axis in (0, 2):
                for dirn in ("+", "-"):
                    for depth, fc in ((1, PC["dome"]), (5, None), (W + 3, PC["plinth"]), (0, PC["dome"])):
                        want = oracle.extrude_from_surface(grid, m2, axis, direction=dirn, depth=depth, fill_color=None if fc is None else np.array(fc))
                        got = pb3d_gpu.extrude_from_surface(grid, m2, axis, direction=dirn, depth=depth, fill_color=None if fc is None else np.array(fc))
                        assert np.array_equal(got, want), (W, H, occ_dens, mask_dens, axis, dirn, depth, fc)
                        fl = None if fc is None else lpal.label_of([n for n in names if PC[n] == fc][0])
                        gl = pb3d_gpu.extrude_from_surface_labels(ids, m2, axis=axis, direction=dirn, depth=depth, fill_label=fl)
                        assert np.array_equal(tab[gl], want), (W, H, occ_dens, mask_dens, axis, dirn, depth, "labels")


@pytest.mark.gpu
def test_global_carve_90_stream_kernel_rgb_and_labels(pb3d_gpu, oracle):
    """global_carve(binary, image, 90) through the stream kernel of round 4 (k_global_carve90s), RGB image and 1-byte label image (row N3),
    against the oracle on widths that are / are not multiples of 16 (groups that straddle two columns, two pixel colours in one group),
    masks from nearly empty to full, odd heights; the label form also against its composed pipeline (ones -> process -> label apply)."""
    rng = np.random.default_rng(515)
    PC = oracle.PART_COLORS
    names = list(PC)[:6]
    lpal = pb3d_gpu.Palette([PC[n] for n in names], names)
    tab = lpal.table()
    for (h, w) in [(5, 16), (7, 17), (9, 33), (4, 48), (6, 100), (3, 131), (8, 160), (2, 355), (11, 64), (1, 19)]:
        for dens in (0.05, 0.6, 1.0):
            lab = (rng.integers(1, 7, (h, w)) * (rng.random((h, w)) < dens)).astype(np.uint8)
            sem = tab[lab]
            binary = (lab != 0).astype(np.uint8) * int(rng.integers(1, 255))
            want = oracle.global_carve(binary, sem, 90)
            assert np.array_equal(pb3d_gpu.global_carve(binary, sem, 90), want), (h, w, dens)
            lg = pb3d_gpu.global_carve_labels(binary, lab, 90)
            assert lg.dtype == np.uint8 and np.array_equal(tab[lg], want), (h, w, dens, "labels")
            pb3d_gpu._lib.set_tuning("per_job", 1)
            try:
                assert np.array_equal(pb3d_gpu.global_carve_labels(binary, lab, 90), lg), (h, w, dens, "labels, composed")
            finally:
                pb3d_gpu._lib.set_tuning("per_job", 0)


@pytest.mark.gpu
def test_part_carve_plane_kernel(pb3d_gpu, oracle):
    """part_carve with 90-degree jobs through the plane-local kernel of round 4 (k_part90_plane: occupancy BITS of a plane's source columns
    in LDS, 32 x 32 bit blocks transposed in registers, whole output rows) against the oracle: W != D with column offsets of either sign,
    rows of 16 .. 400 voxels at every alignment, more output rows than one 128-row workgroup, rows shorter than a 16-byte piece's six
    voxels apart, foreign colours, sparse and dense grids, skipped jobs and jobs whose masks overlap."""
    rng = np.random.default_rng(2024)
    PC = oracle.PART_COLORS
    names = ["full_building", "chhatris", "plinth", "front_minarets", "small_minarets", "dome"]
    pal = np.array([PC[n] for n in names] + [(0, 0, 0), (9, 9, 9)], np.uint8)
    lpal = pb3d_gpu.Palette([PC[n] for n in names], names)
    pal6 = lpal.table()                                                     # label k -> colour (label 0 = black)
    shapes = [(37, 9, 51), (51, 5, 37), (130, 6, 62), (129, 3, 131), (16, 7, 48), (200, 3, 72), (64, 5, 64), (96, 4, 32), (33, 3, 16), (17, 5, 17),
              (355, 4, 355), (131, 16, 131), (300, 3, 172), (172, 3, 300), (260, 2, 260), (128, 2, 128), (19, 1, 23), (400, 1, 400), (5, 3, 7), (9, 2, 3)]
    for (W, H, D) in shapes:
        for dens in (0.15, 0.8):
            sem = pal[rng.integers(0, 7, (H, W))]
            colored = pal[rng.integers(0, len(pal), (W, H, D))]
            colored[rng.random((W, H, D)) > dens] = 0
            for jobs in (JOBS_NB1, [(["dome", "plinth"], 90), (["dome"], 90)], [(["windows"], 90)]):
                want = oracle.part_carve(colored, sem, jobs)
                got = pb3d_gpu.part_carve(colored, sem, jobs)
                assert np.array_equal(got, want), (W, H, D, dens, len(jobs), int((got != want).sum()))
            # the same kernel on 1-byte label volumes (row N3: occupancy = label != 0, 16-voxel pieces) == the RGB result == the per-job label passes
            ids = rng.integers(0, 7, (W, H, D)).astype(np.uint8) * (rng.random((W, H, D)) < dens)
            lcol = pal6[ids]
            lsem = pal6[rng.integers(0, 7, (H, W))]
            lab_mask = lpal.mask_to_labels(lsem)
            want = oracle.part_carve(lcol, lsem, JOBS_NB1)
            got = pb3d_gpu.part_carve_labels(ids, lab_mask, JOBS_NB1, lpal)
            assert np.array_equal(pb3d_gpu.label_to_rgb(got, lpal), want), (W, H, D, dens, "labels")
            pb3d_gpu._lib.set_tuning("per_job", 1)
            try:
                assert np.array_equal(pb3d_gpu.part_carve_labels(ids, lab_mask, JOBS_NB1, lpal), got), (W, H, D, dens, "labels, per job")
            finally:
                pb3d_gpu._lib.set_tuning("per_job", 0)


@pytest.mark.gpu
def test_full_size_process_grid_45_against_oracle_slab(pb3d_gpu, oracle):
    """process_voxel_grid(occ, binary, 45) on the 1024^3 synthetic grid, device resident (the bit-sliced chain: slice with the 0-degree
    carve folded in, the 45-degree table step, the 90-degree step un-slicing in its own stores).  A rotation about Y never mixes
    Y-planes, so a slab of planes of the full result must equal the oracle run on that slab alone; and the byte chain (arithmetic
    kernel + permutation kernel), written independently, must agree on the whole volume."""
    import synth_host
    from pb3d import device as dev
    S = int(os.environ.get("PB3D_TEST_FULL_SIZE", "1024"))
    nvox = S * S * S
    d_occ = dev.DeviceBuffer(nvox); d_out = dev.DeviceBuffer(nvox); d_tmp = dev.DeviceBuffer(nvox); d_out64 = dev.DeviceBuffer(nvox)
    d_mwh = dev.DeviceBuffer(S * S)
    dev.synth_occ(0, S, S, S, 0, d_occ)
    dev.synth_mask16(S, d_binary_wh=d_mwh)
    lab, binary, rgb = synth_host.mask16(S)                  # (H,W) images
    dev.process_grid(d_occ, S, S, S, d_mwh, 45, d_out, d_tmp)
    # the byte chain (0-degree carve folded into the arithmetic kernel's 45-degree step, the 90-degree step on the permutation kernel),
    # written independently of the bit-sliced chain, must agree on the whole volume
    pb3d_gpu._lib.set_tuning("sliced", 1)
    try:
        dev.process_grid(d_occ, S, S, S, d_mwh, 45, d_out64, d_tmp)
    finally:
        pb3d_gpu._lib.set_tuning("sliced", 0)
    dev.sync()
    full = d_out.download((S, S, S))
    assert np.array_equal(full, d_out64.download((S, S, S)))
    assert full.max() <= 1 and 0 < int(full.sum()) < nvox
    occ = d_occ.download((S, S, S))
    for y0 in (0, S // 2 - 1, S - 3):
        ys = slice(y0, y0 + 3)
        want = oracle.process_voxel_grid(np.ascontiguousarray(occ[:, ys, :]), binary[ys, :], 45)
        assert np.array_equal(full[:, ys, :], want), y0
    for b in (d_occ, d_out, d_tmp, d_out64, d_mwh):
        b.free()


@pytest.mark.gpu
def test_y_slab_partition_gpu(pb3d_gpu, golden):
    """SURVEY 8(e) first row on the device path: an uneven 3-way Y-slab partition (18 | 17 | 17 planes of the real Taj masks
    at max_dim 96: a 95 x 52 x 95 grid) of global_carve, a chained 45-degree process_voxel_grid and the notebook-1 part_carve jobs,
    each rank's slab computed on its own with its mask rows, equals the unsharded result after concatenation."""
    from pb3d import dist
    g = golden("f4_Taj_96")
    ext, binary = g["ext"], g["binary"]
    H = binary.shape[0]
    full_gc = pb3d_gpu.global_carve(binary, ext, angle_interval=90)
    assert full_gc.shape[1] == H and full_gc.shape[0] != H            # (w, h, w, 3), non-square
    nr = 3
    parts = [pb3d_gpu.global_carve(dist.y_slab_image(binary, r, nr), dist.y_slab_image(ext, r, nr), angle_interval=90) for r in range(nr)]
    assert np.array_equal(dist.assemble_y_slabs(parts), full_gc)
    occ = (full_gc.any(-1)).astype(np.uint8)
    full_p = pb3d_gpu.process_voxel_grid(occ, binary, 45)
    parts = [pb3d_gpu.process_voxel_grid(dist.y_slab_grid(occ, r, nr), dist.y_slab_image(binary, r, nr), 45) for r in range(nr)]
    assert np.array_equal(dist.assemble_y_slabs(parts), full_p)
    for jobs in (JOBS_NB1, JOBS_MIXED):
        full_pc = pb3d_gpu.part_carve(full_gc, ext, jobs)
        parts = [pb3d_gpu.part_carve(dist.y_slab_grid(full_gc, r, nr), dist.y_slab_image(ext, r, nr), jobs) for r in range(nr)]
        assert np.array_equal(dist.assemble_y_slabs(parts), full_pc)


@pytest.mark.gpu
def test_large_odd_grid_against_oracle_slabs(pb3d_gpu, oracle):
    """a 1001 x 720 x 1001 grid (2.16 GB of RGB: byte offsets beyond 2^31, rows of 1001 bytes at every alignment, ragged
    tiles on all sides) through global_carve, part_carve, a chained 45-degree process_voxel_grid and the mask carve.
    Y-planes are independent, so slabs of planes of the full results must equal the oracle run on those slabs alone."""
    rng = np.random.default_rng(77)
    W, H = 1001, 720
    PC = oracle.PART_COLORS
    names = ["full_building", "chhatris", "plinth", "front_minarets", "small_minarets", "dome"]
    pal = np.array([PC[n] for n in names] + [(0, 0, 0)], np.uint8)
    sem = pal[rng.integers(0, len(pal), (H // 8 + 1, W // 8 + 1))].repeat(8, 0).repeat(8, 1)[:H, :W]        # blocky part image
    binary = sem.any(-1).astype(np.uint8)
    slabs = [(0, 2), (H // 2 - 1, H // 2 + 1), (H - 2, H)]
    gc = pb3d_gpu.global_carve(binary, sem, angle_interval=90)
    assert gc.shape == (W, H, W, 3) and gc.nbytes > 2 ** 31
    for y0, y1 in slabs:
        assert np.array_equal(gc[:, y0:y1], oracle.global_carve(binary[y0:y1], sem[y0:y1], 90)), ("global_carve", y0)
    pc = pb3d_gpu.part_carve(gc, sem, JOBS_NB1)
    for y0, y1 in slabs:
        assert np.array_equal(pc[:, y0:y1], oracle.part_carve(np.ascontiguousarray(gc[:, y0:y1]), sem[y0:y1], JOBS_NB1)), ("part_carve", y0)
    m = rng.random((H, W)) < 0.8
    cv = pb3d_gpu.carve_voxel_grid_with_masks(gc, m)
    assert np.array_equal(cv[::97], np.where(m.T[::97, :, None, None], gc[::97], 0))
    del pc, cv
    occ = np.ascontiguousarray(gc.any(-1)).view(np.uint8)
    del gc
    pv = pb3d_gpu.process_voxel_grid(occ, binary, 45)
    for y0, y1 in slabs:
        assert np.array_equal(pv[:, y0:y1], oracle.process_voxel_grid(np.ascontiguousarray(occ[:, y0:y1]), binary[y0:y1], 45)), ("process45", y0)


@pytest.mark.gpu
def test_plain_c_caller_of_the_cabi(pb3d_gpu, tmp_path):
    """examples/cabi_demo.c: a C99 program (no Python, no C++) carves through libpb3d.so and checks the result itself."""
    import subprocess
    libdir = os.path.dirname(pb3d_gpu._lib.LIB_PATH)
    exe = str(tmp_path / "cabi_demo")
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "cabi_demo.c"),
                    "-o", exe, "-L" + libdir, "-lpb3d", "-Wl,-rpath," + libdir], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "0 mismatching bytes" in r.stdout and " 0 outside the mask" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Taj_96", "Akbar_64"])
def test_device_resident_chain(pb3d_gpu, golden, name):
    """global_carve(on_device=True) -> part_carve -> partwise_carve on DeviceGrid handles (no upload or download of the
    volume in between) gives the same bytes as the NumPy-in / NumPy-out chain and as the reference's stage outputs."""
    import contextlib
    import io
    from pb3d.device import DeviceGrid
    g = golden(f"f5_{name}")
    meta = json.load(open(os.path.join(GOLDEN, "f5_meta.json")))[name]
    PCN = pb3d_gpu.PART_COLORS_NP
    d_gc = pb3d_gpu.global_carve(g["binary"], g["ext"], 90, on_device=True)
    assert isinstance(d_gc, DeviceGrid)
    host_gc = pb3d_gpu.global_carve(g["binary"], g["ext"], 90)
    assert np.array_equal(d_gc.numpy(), host_gc)
    d_pc = pb3d_gpu.part_carve(d_gc, g["ext"], JOBS_NB1)
    assert isinstance(d_pc, DeviceGrid) and np.array_equal(d_pc.numpy(), pb3d_gpu.part_carve(host_gc, g["ext"], JOBS_NB1))
    with contextlib.redirect_stdout(io.StringIO()):
        d_full = pb3d_gpu.partwise_carve(d_gc, g["ext"], g["sem"], PCN, JOBS_NB1, PART_SYMMETRY, EXTRUSION)
        d_plain = pb3d_gpu.partwise_carve(d_gc, g["ext"], g["sem"], PCN, JOBS_NB1, PART_SYMMETRY, EXTRUSION, recolor_back_minarets=False)
    full = d_full.numpy()
    assert sha(full) == meta["partwise_sha256"] and list(full.shape) == meta["partwise_shape"]
    assert d_plain.shape == d_gc.shape
    assert np.array_equal(d_gc.numpy(), host_gc)                       # the resident input is only read
    for d in (d_gc, d_pc, d_full, d_plain):
        d.free()


@pytest.mark.gpu
@pytest.mark.parametrize("mon", ["Akbar", "Bibi", "Charminar", "Itimad", "Taj"])
def test_five_monuments_deformation_loop_m5(pb3d_gpu, mon):
    """BASELINE configs[4]: the part-wise deformation re-projection loop on all five monuments (the reference's stored
    results/1 grids and results/2 final cameras) against digests captured from the reference's notebook-3 closures
    (tools/gen_golden_m5.py): point extraction, projection, deformed coordinates, per-part IoU, deformed grid."""
    meta = json.load(open(os.path.join(GOLDEN, "m5_five_monuments_deformation.json")))[mon]
    grid = np.load(os.path.join(GOLDEN, f"stored_{mon}_voxel_grid.npz"))["voxel_grid"]
    PC = pb3d_gpu.PART_COLORS
    cam = _cams(mon)["front"]
    names = list(PC.keys())
    pts, cols = pb3d_gpu.get_voxel_points_by_parts(grid, PC, names)
    assert len(pts) == meta["n_points"] and sha(pts) == meta["points_sha256"] and sha(cols) == meta["colors_sha256"]
    H, W = meta["image_shape"]
    image = pb3d_gpu.project_colored_voxels(pts, cols, cam["cam_pos"], cam["target"], cam["f"], cam["cx"], cam["cy"], H, W)
    assert sha(image) == meta["image_sha256"]
    saved = {}
    for part, c in meta["cases"].items():
        coords, _ = pb3d_gpu.get_voxel_points_by_parts(grid, PC, [part])
        cd = pb3d_gpu.deform_coords(coords, meta["image_shape"], meta["grid_shape"][:3], c["deform"])
        assert len(cd) == c["n_deformed"] and sha(cd) == c["coords_sha256"], part
        _, iou = pb3d_gpu.evaluate_part_deform(grid, PC, part, c["deform"], image, cam)
        assert iou == c["iou"], (part, iou, c["iou"])
        saved[part] = {"deform": c["deform"], "iou": iou}
    full = pb3d_gpu.build_deformed_grid(grid, PC, saved, meta["image_shape"])
    assert sha(full) == meta["deformed_grid_sha256"] and int(np.any(full > 0, -1).sum()) == meta["deformed_grid_occupied"]


@pytest.mark.gpu
@pytest.mark.parametrize("mon", ["Akbar", "Charminar"])
def test_camera_objective_batched_equals_one_at_a_time(pb3d_gpu, oracle, mon):
    """row N4: K cameras per launch (pb3d_project_iou_batch_dev + pb3d_look_at_batch) give exactly the values of the
    one-camera path -- float64 slider cameras, float32 JSON cameras, NumPy-scalar f / cx (NumPy-2 promotion), cameras behind
    the cloud, straight-down views -- and a few of them are pinned on the oracle's project + IoU."""
    meta = json.load(open(os.path.join(GOLDEN, "n45_objective_zbuffer.json")))
    grid = np.load(os.path.join(GOLDEN, f"stored_{mon}_voxel_grid.npz"))["voxel_grid"]
    PC = pb3d_gpu.PART_COLORS
    m = meta[f"objective_{mon}"]
    front = np.load(os.path.join(GOLDEN, "f7_projection.npz"))[f"img_{mon}_front"]
    seg = pb3d_gpu.mask_parts_from_image(front, PC, m["parts"])
    pts, cols = pb3d_gpu.get_voxel_points_by_parts(grid, PC, m["parts"])
    labels = {p: PC[p] for p in m["parts"]}
    obj = pb3d_gpu.CameraObjective(pts, cols, seg, labels)
    rng = np.random.default_rng(77)
    t0 = m["trials"][0]
    base = np.array(list(t0["cam_pos"]) + list(t0["target"]) + [t0["f"], t0["cx"], t0["cy"]], np.float64)
    params = []
    for k in range(150):
        x = base + rng.normal(size=9) * np.array([20, 20, 20, 5, 5, 5, 15, 4, 4]) * (k % 5 == 0 and 10 or 1)
        p = {"cam_pos": x[:3], "target": x[3:6], "f": x[6], "cx": x[7], "cy": x[8], "H": m["H"], "W": m["W"]}
        if k % 7 == 1:
            p["f"] = float(x[6]); p["cx"] = float(x[7]); p["cy"] = float(x[8])          # Python floats: weak scalars
        params.append(p)
    p32 = [{"cam_pos": p["cam_pos"].astype(np.float32), "target": p["target"].astype(np.float32), "f": float(p["f"]), "cx": float(p["cx"]),
            "cy": float(p["cy"]), "H": m["H"], "W": m["W"]} for p in params[:60]]                # the JSON path of notebooks 2/3
    p32[3]["f"] = np.float64(p32[3]["f"])                                                      # a float64 NumPy scalar widens the later stages
    p32[4]["cam_pos"] = p32[4]["target"] + np.array([0, 50, 0], np.float32)                    # looking straight down: the alternative up vector
    for plist in (params, p32, params[:1], []):
        got = obj.evaluate_batch(plist)
        assert got == [obj(p) for p in plist]
        assert all(type(a) is type(b) for a, b in zip(got, [obj(p) for p in plist[:3]]))
    mixed = [params[0], p32[0], params[1]]                                                      # mixed dtypes in one list: per-camera NumPy route
    assert obj.evaluate_batch(mixed) == [obj(p) for p in mixed]
    fixture = [{"cam_pos": np.array(t["cam_pos"]), "target": np.array(t["target"]), "f": t["f"], "cx": t["cx"], "cy": t["cy"]} for t in m["trials"]]
    assert obj.evaluate_batch(fixture) == [t["neg_iou"] for t in m["trials"]]                  # the reference's own values
    for p in (params[0], params[5], p32[2], p32[3]):                                            # the oracle's own projection + IoU
        proj = oracle.project_colored_voxels(pts, cols, p["cam_pos"], p["target"], p["f"], p["cx"], p["cy"], m["H"], m["W"])
        _, mean_iou = oracle.compute_partwise_iou(proj, seg, labels)
        assert obj.evaluate_batch([p])[0] == -mean_iou
    obj.close()
    # more than 8 parts: the mean takes NumPy's 8-accumulator path -- still the same bits as the single-camera route
    names = list(PC.keys())
    pts, cols = pb3d_gpu.get_voxel_points_by_parts(grid, PC, names)
    obj = pb3d_gpu.CameraObjective(pts, cols, front, {n: PC[n] for n in names})
    assert obj.evaluate_batch(params[:20]) == [obj(p) for p in params[:20]]
    obj.close()


@pytest.mark.gpu
def test_deform_tuples_batched_equals_one_at_a_time(pb3d_gpu):
    """row N4, deformation side: a grid of deform tuples through pb3d_deform_iou_batch_dev == evaluate_part_deform per tuple
    (unique + bounds filter + projection + IoU), including tuples that push the whole part out of the grid."""
    mon = "Akbar"
    meta = json.load(open(os.path.join(GOLDEN, "m5_five_monuments_deformation.json")))[mon]
    grid = np.load(os.path.join(GOLDEN, f"stored_{mon}_voxel_grid.npz"))["voxel_grid"]
    PC = pb3d_gpu.PART_COLORS
    cam = _cams(mon)["front"]
    pts, cols = pb3d_gpu.get_voxel_points_by_parts(grid, PC, list(PC.keys()))
    H, W = meta["image_shape"]
    image = pb3d_gpu.project_colored_voxels(pts, cols, cam["cam_pos"], cam["target"], cam["f"], cam["cx"], cam["cy"], H, W)
    for part, c in list(meta["cases"].items())[:2]:
        deforms = [c["deform"]]
        for sy in (0.8, 1.0, 1.2):
            for sxz in (0.9, 1.1):
                for dy in (-60.0, 0.0, 35.0):
                    for dxz in (-20.0, 0.0, 60.0):
                        deforms.append({"scale_y": sy, "shift_y": dy, "scale_xz": sxz, "shift_xz": dxz})
        deforms.append({"scale_y": 1.0, "shift_y": 100000.0, "scale_xz": 1.0, "shift_xz": 0.0})        # everything leaves the grid
        ious, nvalid = pb3d_gpu.evaluate_part_deform_batch(grid, PC, part, deforms, image, cam)
        assert ious[0] == c["iou"]                                                              # the reference's own value
        assert nvalid[-1] == 0 and ious[-1] == 0.0 and (nvalid > 0).sum() > 10
        for k in range(0, len(deforms) - 1, 5):
            if nvalid[k] == 0:                       # the whole part left the grid: upstream has nothing to project
                assert ious[k] == 0.0 and len(pb3d_gpu.deform_part(grid, PC, part, deforms[k], image.shape[:2])[0]) == 0
            else:
                assert ious[k] == pb3d_gpu.evaluate_part_deform(grid, PC, part, deforms[k], image, cam)[1], (part, deforms[k])


@pytest.mark.gpu
def test_label_form_expands_to_the_rgb_results(pb3d_gpu, oracle, golden):
    """row N3, device half: palette -> 1-byte labels, the carve ops on label volumes, and the expansion back: every result
    equals, byte for byte, what the RGB entry points (and the oracle) return."""
    PC = pb3d_gpu.PART_COLORS
    pal = pb3d_gpu.Palette.from_part_colors(PC)
    rng = np.random.default_rng(91)
    # conversions: every palette colour + black, ragged sizes, round trip; unknown colours / labels are refused
    for shape in ((5, 7, 9), (16, 3, 32), (1, 1, 1), (33, 2, 17)):
        lab = rng.integers(0, len(pal) + 1, shape).astype(np.uint8)
        rgb = pal.table()[lab]
        assert np.array_equal(pb3d_gpu.rgb_to_label(rgb, pal), lab)
        assert np.array_equal(pb3d_gpu.label_to_rgb(lab, pal), rgb)
    bad = pal.table()[rng.integers(0, len(pal) + 1, (4, 4, 4))].copy(); bad[1, 2, 3] = (1, 2, 3)
    with pytest.raises(ValueError, match="neither black nor in the palette"):
        pb3d_gpu.rgb_to_label(bad, pal)
    with pytest.raises(ValueError, match="exceeds the palette"):
        pb3d_gpu.label_to_rgb(np.full((3, 3), len(pal) + 1, np.uint8), pal)
    with pytest.raises(ValueError):
        pb3d_gpu.Palette([(1, 2, 3), (1, 2, 3)])
    # the real masks: global_carve and part_carve in label form == the RGB path == the oracle
    for name in ("f4_Akbar_64", "f4_Taj_96"):
        g = golden(name)
        sem, binary = g["ext"] if "ext" in g.files else g["sem"], g["binary"]
        lab_mask = pal.mask_to_labels(sem)
        assert lab_mask.shape == sem.shape[:2] and np.array_equal(pal.table()[lab_mask], sem)
        for ai in (90, 45):
            want = oracle.global_carve(binary, sem, ai)
            lg = pb3d_gpu.global_carve_labels(binary, lab_mask, ai)
            assert lg.dtype == np.uint8 and lg.shape == want.shape[:3]
            assert np.array_equal(pb3d_gpu.label_to_rgb(lg, pal), want), (name, ai)
        gc = oracle.global_carve(binary, sem, 90)
        lgc = pb3d_gpu.rgb_to_label(gc, pal)
        # carve_voxel_grid_with_masks takes the label volume as it is (C = 1)
        m2 = rng.random(binary.shape) < 0.7
        assert np.array_equal(pb3d_gpu.label_to_rgb(pb3d_gpu.carve_voxel_grid_with_masks(lgc, m2), pal), oracle.carve_voxel_grid_with_masks(gc, m2))
        present = [n for n in PC if np.all(sem == np.array(PC[n], np.uint8), axis=-1).any()]
        jobs = [([present[0]], 90), (present[1:3], 45), ([present[-1]], 90)]
        want = oracle.part_carve(gc, sem, jobs)
        assert np.array_equal(pb3d_gpu.part_carve(gc, sem, jobs), want)
        assert np.array_equal(pb3d_gpu.label_to_rgb(pb3d_gpu.part_carve_labels(lgc, lab_mask, jobs, pal), pal), want), name


@pytest.mark.gpu
def test_sharded_entry_points_single_rank_rccl(pb3d_gpu, oracle):
    """pb3d_carve_mask_sharded_dev / pb3d_global_carve_sharded_dev / pb3d_carve_labels_sharded_dev through RCCL on a 1-rank
    communicator (slab = the whole grid): the RGB gather, the label gather and its local expansion all equal the unsharded op;
    pb3d_comm_info reports what RCCL itself sees."""
    from pb3d import device as dev, dist
    import synth_host
    S = 64
    lab, binary, rgb = synth_host.mask16(S)
    pal16 = dev.synth_palette16()
    sem = synth_host.sem_slab(0, S, S, S, seed=4)
    m_wh = np.ascontiguousarray(binary.T)
    want = oracle.carve_voxel_grid_with_masks(sem, binary)          # square grid: the (H,W) reading of the mask wins
    d_in = dev.from_numpy(sem); d_m = dev.from_numpy(m_wh); d_full = dev.DeviceBuffer(sem.nbytes)
    d_lab_in = dev.DeviceBuffer(S ** 3); d_lab_full = dev.DeviceBuffer(S ** 3); d_rgb2 = dev.DeviceBuffer(sem.nbytes)
    d_b = dev.from_numpy(binary); d_rgbm = dev.from_numpy(rgb); d_gc = dev.DeviceBuffer(sem.nbytes)
    with pytest.raises(ValueError, match="pb3d_comm_init first"):
        dist.carve_mask_sharded(d_in, S, S, S, 3, d_m, d_full)
    dist.comm_init(dist.new_unique_id(), 0, 1)
    try:
        assert dist.comm_info() == (0, 1)
        dist.carve_mask_sharded(d_in, S, S, S, 3, d_m, d_full)
        dev.rgb_to_label(d_in, S ** 3, pal16, d_lab_in)
        dist.carve_labels_sharded(d_lab_in, S, S, S, d_m, d_lab_full, pal16, d_rgb2)
        dist.global_carve_sharded(d_b, d_rgbm, S, S, 90, d_gc)
        dev.sync()
        assert np.array_equal(d_full.download(sem.shape), want)
        assert np.array_equal(d_rgb2.download(sem.shape), want)                                   # label gather + local expansion == RGB gather
        labs = d_lab_full.download((S, S, S))
        assert labs.max() <= 16 and np.array_equal(np.concatenate([np.zeros((1, 3), np.uint8), pal16])[labs], want)
        assert np.array_equal(d_gc.download(sem.shape), oracle.global_carve(binary, rgb, 90))
    finally:
        dist.comm_destroy()
        for b in (d_in, d_m, d_full, d_lab_in, d_lab_full, d_rgb2, d_b, d_rgbm, d_gc):
            b.free()


def _structured_color_grid(pb3d_gpu, S):
    """global_carve(binary, rgb, 90) of the synthetic 16-label mask, resident in HBM: the realistic 1024^3 colour grid of opbench."""
    from pb3d import device as dev
    d_bhw = dev.DeviceBuffer(S * S); d_rgb = dev.DeviceBuffer(S * S * 3)
    dev.synth_mask16(S, d_binary_hw=d_bhw, d_rgb_hw3=d_rgb)
    d_col = dev.DeviceBuffer(S ** 3 * 3)
    dev.global_carve(d_bhw, d_rgb, S, S, 90, d_col)
    dev.sync()
    d_bhw.free(); d_rgb.free()
    return d_col


@pytest.mark.gpu
def test_full_size_points_and_projection_1024(pb3d_gpu, oracle):
    """M7 / M8 at the size opbench times them (1024^3 grid -> ~416 M points = 5 GB of float32, past the 2^32-byte mark):
    * the point count equals an INDEPENDENT device reduction (per-colour voxel counts from the IoU kernel);
    * whole X-planes of the point list -- first, middle, the one holding byte 2^32 of the buffer, last -- equal np.where on the
      matching plane of the grid (coordinates, colours, order);
    * the projection of all points equals the max-merge of 8 index-keyed shards, and on every pixel hit by one of the last two
      million points it equals the oracle's projection of those points alone (the highest index wins a pixel)."""
    import ctypes as C
    from pb3d import device as dev
    L, lib = pb3d_gpu._lib, pb3d_gpu._lib.load()
    S = int(os.environ.get("PB3D_TEST_FULL_SIZE", "1024"))
    nvox = S ** 3
    PC = pb3d_gpu.PART_COLORS
    cols = np.ascontiguousarray(np.array(list(PC.values()), np.uint8))
    d_col = _structured_color_grid(pb3d_gpu, S)

    def count(planes):
        n = C.c_int64(0)
        L.check(lib.pb3d_points_count_dev(L.ctx(), C.c_void_p(d_col.ptr), planes, S, S, 3, L.p_u8(cols), len(cols), 1, C.byref(n)))
        return n.value
    npts = count(S)
    inter = np.zeros(len(cols), np.int64); uni = np.zeros(len(cols), np.int64)
    L.check(lib.pb3d_partwise_iou_dev(L.ctx(), C.c_void_p(d_col.ptr), C.c_void_p(d_col.ptr), nvox, L.p_u8(cols), len(cols),
                                      inter.ctypes.data_as(L.i64p), uni.ctypes.data_as(L.i64p)))
    assert npts == int(inter.sum()) == int(uni.sum()) and npts > 0
    if S == 1024:
        assert npts * 12 > 2 ** 32                                           # the buffer really crosses the 4 GiB mark
    d_pts = dev.DeviceBuffer(npts * 12); d_pc = dev.DeviceBuffer(npts * 3)
    L.check(lib.pb3d_points_fill_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, 3, L.p_u8(cols), len(cols), 1, npts, C.c_void_p(d_pts.ptr),
                                     C.c_void_p(d_pc.ptr)))
    dev.sync()
    key = lambda a: a[..., 0].astype(np.uint32) | (a[..., 1].astype(np.uint32) << 8) | (a[..., 2].astype(np.uint32) << 16)
    ckeys = key(cols)
    # plane that holds point index 2^32 / 12 (bisection on the prefix counts), plus first / middle / last non-empty planes
    target = min(npts - 1, (2 ** 32) // 12)
    lo, hi = 0, S
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if count(mid) <= target: lo = mid
        else: hi = mid
    planes = sorted({lo, S // 2, S - 1} | {next(a for a in range(S) if count(a + 1) > 0)} | {max(a for a in range(S) if count(a + 1) > count(a))})
    for a in planes:
        c0, c1 = count(a), count(a + 1)
        plane = d_col.download((S, S, 3), byte_offset=a * S * S * 3)
        sel = np.isin(key(plane), ckeys)
        a1, a2 = np.nonzero(sel)
        assert c1 - c0 == len(a1), a
        if c1 == c0:
            continue
        got_p = d_pts.download((c1 - c0, 3), np.float32, byte_offset=c0 * 12)
        got_c = d_pc.download((c1 - c0, 3), byte_offset=c0 * 3)
        want_p = np.stack([a2, a1, np.full(len(a1), a)], axis=1).astype(np.float32)
        assert np.array_equal(got_p, want_p) and np.array_equal(got_c, plane[a1, a2]), a
    # ---- the one-pass form (decoupled look-back): same count, same rows, in the same order; too small a capacity is reported
    d_pts2 = dev.DeviceBuffer(npts * 12); d_pc2 = dev.DeviceBuffer(npts * 3)
    n2 = C.c_int64(0)
    L.check(lib.pb3d_points_extract_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, 3, L.p_u8(cols), len(cols), npts, C.c_void_p(d_pts2.ptr),
                                        C.c_void_p(d_pc2.ptr), C.byref(n2)))
    assert n2.value == npts
    for a in planes:
        c0, c1 = count(a), count(a + 1)
        if c1 > c0:
            assert np.array_equal(d_pts2.download((c1 - c0, 3), np.float32, byte_offset=c0 * 12), d_pts.download((c1 - c0, 3), np.float32, byte_offset=c0 * 12))
            assert np.array_equal(d_pc2.download((c1 - c0, 3), byte_offset=c0 * 3), d_pc.download((c1 - c0, 3), byte_offset=c0 * 3))
    L.check(lib.pb3d_points_extract_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, 3, L.p_u8(cols), len(cols), npts // 2, C.c_void_p(d_pts2.ptr),
                                        C.c_void_p(d_pc2.ptr), C.byref(n2)))
    assert n2.value == npts                                                  # the count is still exact; the caller sees n > capacity
    d_pts2.free(); d_pc2.free()
    # ---- M8 on the full list
    from pb3d.camera_geometry import look_at_rotation
    cam = np.array([S / 2, S / 2, -2.5 * S], np.float32); tgt = np.array([S / 2, S / 2, S / 2], np.float32)
    R = np.ascontiguousarray(look_at_rotation(cam, tgt), np.float64); cd = np.ascontiguousarray(cam, np.float64)
    f, cx, cy = float(1.2 * S), S / 2.0, S / 2.0
    prec = (C.c_int * 4)(0, 0, 0, 0)
    d_img = dev.DeviceBuffer(S * S * 3)
    L.check(lib.pb3d_project_dev(L.ctx(), C.c_void_p(d_pts.ptr), 0, C.c_void_p(d_pc.ptr), npts, L.p_dbl(R), L.p_dbl(cd), f, cx, cy, prec, S, S,
                                 C.c_void_p(d_img.ptr)))
    img = d_img.download((S, S, 3))
    d_keys = dev.DeviceBuffer(S * S * 8)
    merged = np.zeros((S, S), np.uint64)
    for r in range(8):
        i0, i1 = pb3d_gpu.dist.point_shard_bounds(npts, r, 8)
        L.check(lib.pb3d_project_keys_dev(L.ctx(), C.c_void_p(d_pts.ptr + i0 * 12), 0, C.c_void_p(d_pc.ptr + i0 * 3), i1 - i0, i0, L.p_dbl(R), L.p_dbl(cd),
                                          f, cx, cy, prec, S, S, C.c_void_p(d_keys.ptr)))
        merged = np.maximum(merged, d_keys.download((S, S), np.uint64))
    assert np.array_equal(pb3d_gpu.dist.resolve_keys(merged), img)
    ntail = min(npts, 2_000_000)
    tail_p = d_pts.download((ntail, 3), np.float32, byte_offset=(npts - ntail) * 12)
    tail_c = d_pc.download((ntail, 3), byte_offset=(npts - ntail) * 3)
    sub = oracle.project_colored_voxels(tail_p, tail_c, cam, tgt, f, cx, cy, S, S)
    hit = np.zeros((S, S), bool)
    one = oracle.project_colored_voxels(tail_p, np.full((ntail, 3), 255, np.uint8), cam, tgt, f, cx, cy, S, S)
    hit = one.any(-1)
    assert hit.any() and np.array_equal(img[hit], sub[hit])
    for b in (d_col, d_pts, d_pc, d_img, d_keys):
        b.free()


@pytest.mark.gpu
def test_full_size_connected_components_1024(pb3d_gpu, oracle):
    """CCL at the size opbench times it: the int32 label volume of one part colour of the 1024^3 structured grid (4 GiB) equals
    the oracle's scipy-numbered labelling voxel for voxel, and the per-component statistics equal bincounts of it."""
    import ctypes as C
    from pb3d import device as dev
    L, lib = pb3d_gpu._lib, pb3d_gpu._lib.load()
    S = int(os.environ.get("PB3D_TEST_FULL_SIZE", "1024"))
    nvox = S ** 3
    d_col = _structured_color_grid(pb3d_gpu, S)
    col = np.array(pb3d_gpu.PART_COLORS["full_building"], np.uint8)
    d_lab = dev.DeviceBuffer(nvox * 4)
    ncomp = C.c_int64(0)
    L.check(lib.pb3d_label_color_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, L.p_u8(col), C.c_void_p(d_lab.ptr), C.byref(ncomp)))
    n = ncomp.value
    bbox = (C.c_int64 * (6 * max(n, 1)))(); cnt = (C.c_int64 * max(n, 1))(); csum = (C.c_int64 * (3 * max(n, 1)))()
    L.check(lib.pb3d_component_stats_dev(L.ctx(), C.c_void_p(d_lab.ptr), S, S, S, n, bbox, cnt, csum))
    # the oracle labels the same mask plane block by plane block is not possible (components cross planes): whole volume at once
    mask = np.empty((S, S, S), np.uint8)
    step = max(1, S // 8)
    for x0 in range(0, S, step):
        g = d_col.download((step, S, S, 3), byte_offset=x0 * S * S * 3)
        mask[x0:x0 + step] = np.all(g == col, axis=-1)
    d_col.free()
    want, nw = oracle.label6(mask)
    assert n == nw and n > 0
    for x0 in range(0, S, step):                                             # 4 GiB of labels, compared in slabs
        got = d_lab.download((step, S, S), np.int32, byte_offset=x0 * S * S * 4)
        assert np.array_equal(got, want[x0:x0 + step]), x0
    d_lab.free()
    counts = np.bincount(want.ravel(), minlength=n + 1)[1:]
    assert np.array_equal(np.array(cnt[:n]), counts)
    xs = np.arange(S, dtype=np.int64)
    for axis in range(3):                                                    # coordinate sums per component along each axis
        sums = np.zeros(n + 1, np.int64)
        moved = np.moveaxis(want, axis, 0)
        for v in range(S):
            sums += np.bincount(moved[v].ravel(), minlength=n + 1) * v
        assert np.array_equal(np.array(csum[:3 * n]).reshape(n, 3)[:, axis], sums[1:]), axis


@pytest.mark.gpu
def test_global_carve_fused_chain_other_angles(pb3d_gpu, oracle, golden):
    """global_carve with angle steps other than 90: the fused chain (first step synthesised from the mask, colours written by the last
    step) and the composed pipeline (ones -> process_voxel_grid -> colour apply, knob global_composed = 1) both equal the oracle -- widths that
    suit the fused kernels (w % 16 == 0) and widths that fall back, one / two / three rotation steps, and the reference fixtures."""
    rng = np.random.default_rng(123)
    pal = np.array(list(pb3d_gpu.PART_COLORS.values()), np.uint8)
    for (h, w) in ((40, 64), (33, 48), (20, 128), (31, 50), (17, 37), (64, 160)):
        lab = rng.integers(0, len(pal), (h // 4 + 1, w // 4 + 1)).repeat(4, 0).repeat(4, 1)[:h, :w]
        sem = pal[lab]
        binary = (rng.random((h, w)) < 0.85) & (lab != len(pal) - 1)
        for ai in (45, 60, 30, 50, 89, 10):
            want = oracle.global_carve(binary, sem, ai)
            # (composed, sliced): the bit-sliced chain from the mask to the colours (default from two rotation steps up), the byte
            # chain with the fused first / last steps (tune sliced = 1), the composed pipeline with and without the sliced middle
            for composed, sliced in ((0, 0), (0, 1), (1, 0), (1, 1)):
                pb3d_gpu._lib.set_tuning("global_composed", composed); pb3d_gpu._lib.set_tuning("sliced", sliced)
                try:
                    got = pb3d_gpu.global_carve(binary, sem, ai)
                finally:
                    pb3d_gpu._lib.set_tuning("global_composed", 0); pb3d_gpu._lib.set_tuning("sliced", 0)
                assert np.array_equal(got, want), (h, w, ai, composed, sliced, int((got != want).sum()))
    for name in ("f4_Akbar_64", "f4_Bibi_64", "f4_Taj_96"):
        g = golden(name)
        assert np.array_equal(pb3d_gpu.global_carve(g["binary"], g["ext"], 45), g["global_carve_45"]), name


def test_search_loops_batched_equal_the_reference_buttons(pb3d_gpu):
    """Row N4, the loops: Random Search (all trials ONE launch), Coordinate Descent (a round's 18 trials ONE launch, first improvement
    wins) and Powell over the resident CameraObjective leave exactly the parameters the reference's buttons left (fixture: headless
    drive of launch_smart_aligner with a seeded np.random, tools/gen_golden_n4_loops.py)."""
    from scipy.optimize import minimize
    from conftest import n4_loop_cases, n4_run_case
    from pb3d.camera_estimation import CameraObjective
    cases, front, grid = n4_loop_cases()
    PC = pb3d_gpu.PART_COLORS
    for case in cases:
        parts = case["parts"]
        pts, cols = pb3d_gpu.get_voxel_points_by_parts(grid, PC, parts)
        seg = pb3d_gpu.mask_parts_from_image(front, PC, parts)
        obj = CameraObjective(pts, cols, seg, {p: PC[p] for p in parts})
        try:
            s1, s2, s3 = n4_run_case(case, obj, minimize)
        finally:
            obj.close()
        assert s1 == case["after_random"], (parts, "random")
        assert s2 == case["after_coord"], (parts, "coord")
        assert s3 == case["after_powell"], (parts, "powell")


def test_orient_kernels(pb3d_gpu):
    """the output orientation of partwise_carve (reference utils/voxel_carving_utils.py:384-385: flip(transpose(2,1,0,3), axis=1)) as
    ONE pass: the 128-pixel-tile kernel (x and z extents multiples of 128: whole 128-byte lines on both sides), the dword kernel
    (multiples of 4) and the byte kernel, against NumPy."""
    import ctypes as C
    from pb3d import device as dev
    L = pb3d_gpu._lib
    rng = np.random.default_rng(12)
    for (W, H, D) in [(128, 5, 128), (256, 3, 128), (128, 4, 384), (64, 3, 64), (37, 5, 41), (128, 2, 100), (132, 3, 128)]:
        g = rng.integers(0, 256, (W, H, D, 3), dtype=np.uint8)
        want = np.ascontiguousarray(np.flip(g.transpose(2, 1, 0, 3), axis=1))
        d_in = dev.from_numpy(g); d_out = dev.DeviceBuffer(g.size)
        for knob in (0, 1):                      # 1: without the 128-tile kernel
            L.set_tuning("orient_tile", knob)
            try:
                L.check(L.load().pb3d_orient_dev(L.ctx(), C.c_void_p(d_in.ptr), W, H, D, C.c_void_p(d_out.ptr)))
                got = d_out.download((D, H, W, 3))
            finally:
                L.set_tuning("orient_tile", 0)
            assert np.array_equal(got, want), (W, H, D, knob)
        d_in.free(); d_out.free()


@pytest.mark.parametrize("name", ["Taj_96", "Akbar_64"])
def test_label_chain_end_to_end(pb3d_gpu, golden, name):
    """Row N3: the WHOLE notebook-1 chain on 1-byte label volumes (global_carve -> part_carve -> component-guided carve -> extrusion
    -> orientation -> recolouring) expands to the reference's partwise_carve digest; its stages equal the RGB stages and print the
    RGB log; point extraction on label volumes (by parts, by occupancy with a stride) equals the RGB extraction of the expanded grid."""
    import contextlib
    import io
    g = golden(f"f5_{name}")
    meta = json.load(open(os.path.join(GOLDEN, "f5_meta.json")))[name]
    PC = pb3d_gpu.PART_COLORS; PCN = pb3d_gpu.PART_COLORS_NP
    pal = pb3d_gpu.Palette.from_part_colors(PC)
    lab_ext = pal.mask_to_labels(g["ext"]); lab_sem = pal.mask_to_labels(g["sem"])
    lgc = pb3d_gpu.global_carve_labels(g["binary"], lab_ext)
    lpc = pb3d_gpu.part_carve_labels(lgc, lab_ext, JOBS_NB1, pal)
    grid = lpc
    for (part, angle), want_log in zip(PART_SYMMETRY.items(), meta["lrgc_stdout"]):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            grid = pb3d_gpu.left_right_guided_carve_labels(grid, lab_ext, pal.label_of(part), angle=angle, log_color=PCN[part])
        assert sha(pb3d_gpu.label_to_rgb(grid, pal)) == meta["stages"][f"lrgc_{part}"], part
        assert buf.getvalue() == want_log
    for part, depth in EXTRUSION.items():
        mk = lab_sem == pal.label_of(part)
        for ax, dr in ((2, "+"), (2, "-"), (0, "+"), (0, "-")):
            grid = pb3d_gpu.extrude_from_surface_labels(grid, mk, axis=ax, direction=dr, depth=depth, fill_label=pal.label_of(part))
            assert sha(pb3d_gpu.label_to_rgb(grid, pal)) == meta["stages"][f"extrude_{part}_{ax}{dr}"], (part, ax, dr)
    oriented = np.flip(grid.transpose(2, 1, 0), axis=1)
    rec = pb3d_gpu.recolor_backward_components_labels(oriented, pal.label_of("front_minarets"), pal.label_of("back_minarets"), k=2, sort_axis=0)
    assert np.array_equal(pb3d_gpu.label_to_rgb(rec, pal), g["after_recolor"])
    with contextlib.redirect_stdout(io.StringIO()):
        full = pb3d_gpu.partwise_carve_labels(lgc, lab_ext, lab_sem, pal, JOBS_NB1, PART_SYMMETRY, EXTRUSION)
    rgb = pb3d_gpu.label_to_rgb(full, pal)
    assert sha(rgb) == meta["partwise_sha256"] and list(rgb.shape) == meta["partwise_shape"]
    # point extraction on the label volume == on the expanded grid
    for parts in (["dome", "plinth"], ["front_minarets"], list(PC.keys())):
        p1, c1 = pb3d_gpu.get_voxel_points_by_parts_labels(full, pal, parts)
        p2, c2 = pb3d_gpu.get_voxel_points_by_parts(rgb, PC, parts)
        assert np.array_equal(p1, p2) and np.array_equal(c1, c2), parts
    for st in (1, 2, 3):
        p1, c1, s1 = pb3d_gpu.voxel_grid_to_points_labels(full, pal, stride=st)
        p2, c2, s2 = pb3d_gpu.voxel_grid_to_points(rgb, stride=st)
        assert np.array_equal(p1, p2) and np.array_equal(c1, c2) and tuple(s1) == tuple(s2), st


def test_partwise_carve_host_waits_and_fallbacks(pb3d_gpu, golden, oracle):
    """partwise_carve (reference :302-400) queues every stage behind ONE labelling of all part colours: the resident chain waits for the
    device at most twice (pb3d_sync_count), the printed log and the grid are the oracle's.  The paths off the main road give the same
    result and log: a part colour named twice (second carve sees what the first left), a crop too large for the LDS-resident component
    loop (x-z box beyond 138 x 138: per-component entries, RGB and label form -- ADVICE r3), an empty angle loop (angle > 90)."""
    import contextlib
    import io
    from pb3d import device as dev
    PC = pb3d_gpu.PART_COLORS; PCN = pb3d_gpu.PART_COLORS_NP
    g = golden("f5_Taj_96")
    gc = pb3d_gpu.global_carve(g["binary"], g["ext"], 90)
    b1, b2 = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(b2):
        want = oracle.partwise_carve(gc, g["ext"], g["sem"], PCN, JOBS_NB1, PART_SYMMETRY, EXTRUSION)
    d_gc = dev.DeviceGrid(dev.from_numpy(gc), gc.shape)
    with contextlib.redirect_stdout(io.StringIO()):
        pb3d_gpu.partwise_carve(d_gc, g["ext"], g["sem"], PCN, JOBS_NB1, PART_SYMMETRY, EXTRUSION).free()        # (scratch buffers of this shape exist now)
    w0 = dev.sync_count()
    with contextlib.redirect_stdout(b1):
        d_res = pb3d_gpu.partwise_carve(d_gc, g["ext"], g["sem"], PCN, JOBS_NB1, PART_SYMMETRY, EXTRUSION)
    waits = dev.sync_count() - w0
    got = d_res.numpy(); d_res.free(); d_gc.free()
    assert np.array_equal(got, want) and b1.getvalue() == b2.getvalue()
    assert waits <= 2, waits
    # a colour named twice + an empty angle loop: one by one, same log
    PCN2 = dict(PCN); PCN2["dome_again"] = PCN["dome"]
    sym2 = {"dome": 45, "chhatris": 120, "dome_again": 30, "front_minarets": 5}
    b1, b2 = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(b1):
        got = pb3d_gpu.partwise_carve(gc, g["ext"], g["sem"], PCN2, JOBS_NB1, sym2, EXTRUSION)
    with contextlib.redirect_stdout(b2):
        want = oracle.partwise_carve(gc, g["ext"], g["sem"], PCN2, JOBS_NB1, sym2, EXTRUSION)
    assert np.array_equal(got, want) and b1.getvalue() == b2.getvalue()
    sym3 = {"dome": 45, "chhatris": 120, "front_minarets": 5}
    b1, b2 = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(b1):
        got = pb3d_gpu.partwise_carve(gc, g["ext"], g["sem"], PCN, JOBS_NB1, sym3, EXTRUSION, recolor_back_minarets=False)
    with contextlib.redirect_stdout(b2):
        want = oracle.partwise_carve(gc, g["ext"], g["sem"], PCN, JOBS_NB1, sym3, EXTRUSION, recolor_back_minarets=False)
    assert np.array_equal(got, want) and b1.getvalue() == b2.getvalue()
    # a crop whose 32-plane slice does not fit the LDS: a 150 x 150 slab of one colour (+ a small second part that still takes the fused loop
    # in the calls after it), RGB form and label form
    rng = np.random.default_rng(5)
    W, H, D = 160, 12, 160
    grid = np.zeros((W, H, D, 3), np.uint8)
    grid[4:154, 2:9, 5:155] = PCN["plinth"]; grid[60:80, 9:12, 60:90] = PCN["dome"]; grid[2:4, 0:2, 0:3] = PCN["chhatris"]
    grid[rng.random((W, H, D)) < 0.02] = 0
    ext = np.zeros((H, W, 3), np.uint8); ext[:] = PC["background"]
    ext[2:9, 4:154] = PC["plinth"]; ext[9:12, 55:85] = PC["dome"]; ext[0:2, 0:6] = PC["chhatris"]
    ext[rng.random((H, W)) < 0.05] = PC["background"]
    sym4 = {"chhatris": 45, "plinth": 30, "dome": 45}
    jobs4 = [(["plinth"], 90), (["dome"], 90), (["chhatris"], 90)]
    b1, b2 = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(b1):
        got = pb3d_gpu.partwise_carve(grid, ext, ext, PCN, jobs4, sym4, {}, recolor_back_minarets=False)
    with contextlib.redirect_stdout(b2):
        want = oracle.partwise_carve(grid, ext, ext, PCN, jobs4, sym4, {}, recolor_back_minarets=False)
    assert b1.getvalue() == b2.getvalue() and "bbox (5,2,7) → (154,9,155)" in b1.getvalue()
    assert np.array_equal(got, want), int((got != want).any(-1).sum())
    pal = pb3d_gpu.Palette.from_part_colors(PC)
    for part, angle in (("plinth", 30), ("dome", 45)):
        b1, b2 = io.StringIO(), io.StringIO()
        with contextlib.redirect_stdout(b1):
            got = pb3d_gpu.left_right_guided_carve(grid, ext, PCN[part], angle=angle)
        with contextlib.redirect_stdout(b2):
            want = oracle.left_right_guided_carve(grid, ext, PCN[part], angle=angle)
        assert np.array_equal(got, want) and b1.getvalue() == b2.getvalue(), part
        b3 = io.StringIO()
        with contextlib.redirect_stdout(b3):
            gl = pb3d_gpu.left_right_guided_carve_labels(pb3d_gpu.rgb_to_label(grid, pal), pal.mask_to_labels(ext), pal.label_of(part), angle=angle, log_color=PCN[part])
        assert np.array_equal(pb3d_gpu.label_to_rgb(gl, pal), want) and b3.getvalue() == b2.getvalue(), part


def test_recolor_backward_many_components(pb3d_gpu, oracle):
    """recolor_backward_components (reference :252-266) on scenes with more components than the labelling's statistics block holds
    (1024: the separate statistics pass regrows other scratch buffers -- ADVICE r3 medium) and more than the device-side keep decision
    takes (2048: host path), on a fresh order of calls; plus equal means (stable order of sorted())."""
    rng = np.random.default_rng(9)
    col = np.array(pb3d_gpu.PART_COLORS["front_minarets"], np.uint8); new = np.array(pb3d_gpu.PART_COLORS["back_minarets"], np.uint8)
    for shp, dens, k, axis in [((40, 36, 44), 0.08, 3, 0), ((64, 60, 50), 0.10, 700, 2), ((30, 20, 26), 0.5, 2, 1), ((8, 6, 40), 0.3, 4, 2)]:
        grid = np.zeros(shp + (3,), np.uint8)
        grid[rng.random(shp) < dens] = col
        grid[rng.random(shp) < 0.02] = (9, 9, 9)
        want = oracle.recolor_backward_components(grid, col, new, k=k, sort_axis=axis)
        got = pb3d_gpu.recolor_backward_components(grid, col, new, k=k, sort_axis=axis)
        assert np.array_equal(got, want), (shp, dens, k, axis)
    # equal means: mirrored blobs on the sort axis
    grid = np.zeros((20, 10, 20, 3), np.uint8)
    for z in (2, 8, 14):
        grid[5:8, 2:5, z:z + 3] = col
    want = oracle.recolor_backward_components(grid, col, new, k=2, sort_axis=0)
    assert np.array_equal(pb3d_gpu.recolor_backward_components(grid, col, new, k=2, sort_axis=0), want)


def test_rot90_wide_tile_kernel(pb3d_gpu, oracle):
    """the 256 x 256-tile form of the 90-degree step (grids from 256 x 256 planes up; tune rot90_wide = 2 pins the 128-tile kernel):
    both bit-exact on whole and clipped tiles."""
    rng = np.random.default_rng(41)
    for (W, H, D) in [(256, 5, 256), (512, 3, 256), (304, 4, 304), (128, 6, 128), (272, 3, 304), (64, 2, 64)]:
        for kind in ("binary", "bytes"):
            g = (rng.random((W, H, D)) < 0.5).astype(np.uint8) if kind == "binary" else rng.integers(0, 256, (W, H, D), dtype=np.uint8)
            m = rng.random((H, W)) < 0.85
            want = oracle.process_voxel_grid(g, m, 90)
            for wide in (0, 2):
                pb3d_gpu._lib.set_tuning("rot90_wide", wide)
                try:
                    got = pb3d_gpu.process_voxel_grid(g, m, 90)
                finally:
                    pb3d_gpu._lib.set_tuning("rot90_wide", 0)
                assert np.array_equal(got, want), (W, H, D, kind, wide, int((got != want).sum()))


def test_rot90_mask_blocks_random_shapes(pb3d_gpu, oracle):
    """the 90-degree kernels that keep a workgroup's mask bytes / job bits in LDS (k_rot90w, k_rot90wf, k_part90), on seeded random
    mid-size shapes: several 8-plane windows per workgroup, heights that end inside a window, W != D (a column offset), mask bytes
    other than 0 / 1, empty mask rows -- block on and off (knob rot90_mask_block), against the oracle."""
    from pb3d import device as dev
    rng = np.random.default_rng(77)
    shapes = [(256, 37, 256), (512, 20, 256), (272, 19, 304), (355, 48, 355), (200, 36, 204), (300, 24, 250), (437, 16, 437), (161, 64, 161),
              (384, 9, 384), (600, 4, 600), (320, 2, 328), (161, 16, 129), (641, 16, 131), (1001, 16, 1001)]      # (three x-tiles; a segment over three planes; a large column offset)
    for (W, H, D) in shapes:
        g = (rng.random((W, H, D)) < 0.5).astype(np.uint8)
        m = (rng.random((H, W)) < 0.8)
        m[:, rng.integers(0, W, 7)] = False; m[rng.integers(0, H, 2), :] = False
        want = oracle.process_voxel_grid(g, m, 90)
        # the device entry takes the (W, H) byte image: any non-zero byte keeps
        mb = np.ascontiguousarray(m.T).astype(np.uint8) * rng.integers(1, 256, (W, H), dtype=np.uint8)
        d_g = dev.from_numpy(g); d_m = dev.from_numpy(mb); d_o = dev.DeviceBuffer(g.size); d_t = dev.DeviceBuffer(g.size)
        for mblk in (0, 1):            # 1: mask bytes per plane / segment instead of the workgroup's LDS block
            pb3d_gpu._lib.set_tuning("rot90_mask_block", mblk)
            try:
                dev.process_grid(d_g, W, H, D, d_m, 90, d_o, d_t)
                got = d_o.download((W, H, D))
            finally:
                pb3d_gpu._lib.set_tuning("rot90_mask_block", 0)
            assert np.array_equal(got, want), (W, H, D, mblk, int((got != want).sum()))
        for b in (d_g, d_m, d_o, d_t):
            b.free()
    pal = np.array(list(pb3d_gpu.PART_COLORS.values()), np.uint8)
    for (W, H, D) in [(256, 37, 256), (384, 21, 384), (304, 40, 304)]:
        lab = rng.integers(1, len(pal), (H // 4 + 1, W // 4 + 1)).repeat(4, 0).repeat(4, 1)[:H, :W]
        sem = pal[lab]
        col = pal[rng.integers(1, len(pal), (W, H, D))] * (rng.random((W, H, D, 1)) < 0.6).astype(np.uint8)
        want = oracle.part_carve(col, sem, JOBS_NB1)
        for mblk in (0, 1):            # 1: mask bytes per plane / segment instead of the workgroup's LDS block
            pb3d_gpu._lib.set_tuning("rot90_mask_block", mblk)
            try:
                got = pb3d_gpu.part_carve(col, sem, JOBS_NB1)
            finally:
                pb3d_gpu._lib.set_tuning("rot90_mask_block", 0)
            assert np.array_equal(got, want), (W, H, D, mblk, int((got != want).any(-1).sum()))


def test_process_typed_non_finite_values_are_carried_not_spread(pb3d_gpu):
    """inf / NaN in a float grid (outside the parity contract, csrc/rotate_typed.hip: SciPy multiplies its zero-weight taps too and turns the
    neighbourhood into NaN): the typed kernel skips zero-weight taps, so a step whose weights are exactly 0 / 1 (the 0-degree step) moves
    non-finite values like any other value -- the result equals the run on the same grid with finite stand-ins, the stand-ins put back.
    (At 90 degrees cos = 6e-17: the second taps carry tiny non-zero weights and a non-finite neighbour does reach the cell, there as in SciPy.)"""
    rng = np.random.default_rng(77)
    W, H, D = 24, 5, 24
    m = rng.random((H, W)) < 0.85
    for dt in ("float32", "float64"):
        a = (rng.random((W, H, D)) * 100 - 50).astype(dt)
        b = a.copy()
        marks = {1234.5: np.inf, -2345.5: -np.inf, 3456.5: np.nan}
        for sent, val in marks.items():
            idx = (rng.integers(0, W, 6), rng.integers(0, H, 6), rng.integers(0, D, 6))
            a[idx] = sent; b[idx] = val
        for ang in (91, 200):           # any step beyond 90: the 0-degree step alone
            want = pb3d_gpu.process_voxel_grid(a, m, ang)
            got = pb3d_gpu.process_voxel_grid(b, m, ang)
            for sent, val in marks.items():
                want = np.where(want == np.asarray(sent, dt), np.asarray(val, dt), want)
            assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), (dt, ang)
            assert np.isfinite(got).sum() == np.isfinite(want).sum()


def test_process_voxel_grid_other_dtypes(pb3d_gpu, oracle, golden):
    """process_voxel_grid on grids that are not uint8 (csrc/rotate_typed.hip): the reference's own outputs for every dtype SciPy's
    interpolation takes (fixture f12), then larger seeded grids against the restatement; float16 is refused as SciPy refuses it; a bool
    grid comes back as int64 (upstream's np.where(mask, grid, 0)), from carve_voxel_grid_with_masks too."""
    g = golden("f12_process_typed")
    for k in g["cases"]:
        k = str(k)
        got = pb3d_gpu.process_voxel_grid(g[k + "_in"], g[k + "_mask"], int(k.split("_")[-1]))
        want = g[k + "_out"]
        assert got.dtype == want.dtype and np.array_equal(got.view(np.uint8), want.view(np.uint8)), k
    rng = np.random.default_rng(91)
    for dt in ("bool", "int8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64", "complex64", "complex128"):
        for (W, H, D), ang in (((70, 9, 65), 5), ((64, 5, 64), 45), ((33, 4, 300), 90), ((1, 1, 1), 30)):
            if dt == "bool":
                a = rng.random((W, H, D)) < 0.5
            elif dt.startswith("complex"):
                a = ((rng.random((W, H, D)) * 400 - 200) + 1j * (rng.random((W, H, D)) * 10 - 5)).astype(dt)
            elif dt.startswith("float"):
                a = (rng.random((W, H, D)) * 400 - 200).astype(dt)
            elif dt.startswith("u"):
                a = (rng.random((W, H, D)) * min(float(np.iinfo(dt).max), 2.0 ** 45)).astype(dt)
            else:
                a = ((rng.random((W, H, D)) - 0.5) * min(float(np.iinfo(dt).max), 2.0 ** 45) * 2).astype(dt)
            m = rng.random((H, W)) < 0.8
            want = oracle.process_voxel_grid_typed(a, m, ang)
            got = pb3d_gpu.process_voxel_grid(a, m, ang)
            assert got.dtype == want.dtype and np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dt, W, H, D, ang, int((got != want).sum()))
    with pytest.raises(RuntimeError, match="data type not supported"):
        pb3d_gpu.process_voxel_grid(np.zeros((4, 3, 4), np.float16), np.ones((3, 4), bool), 45)
    b = rng.random((6, 5, 7)) < 0.5
    mb = rng.random((5, 6)) < 0.6
    got = pb3d_gpu.carve_voxel_grid_with_masks(b, mb)
    want = np.where(mb.T[:, :, None], b, 0)
    assert got.dtype == want.dtype == np.int64 and np.array_equal(got, want)
    assert pb3d_gpu.process_voxel_grid(b, mb, -3).dtype == np.bool_        # the empty angle loop returns the grid as it came


def test_recolour_entries_agree(pb3d_gpu, oracle):
    """pb3d_recolor_components_dev (scans the whole label volume) and pb3d_recolor_last_labelled_dev (walks the labelling's membership
    bits) write the same grid; the second refuses a label buffer that is not the last labelled one."""
    import ctypes as C
    from pb3d import device as dev
    from pb3d.voxel_carving_utils import _label_stats
    L = pb3d_gpu._lib
    rng = np.random.default_rng(5)
    col = np.array(pb3d_gpu.PART_COLORS["front_minarets"], np.uint8); new = np.array(pb3d_gpu.PART_COLORS["back_minarets"], np.uint8)
    for shp in [(40, 33, 70), (17, 9, 130), (64, 64, 64)]:
        grid = np.zeros(shp + (3,), np.uint8)
        for _ in range(12):
            lo = [int(rng.integers(0, s - 1)) for s in shp]; hi = [min(s, l + int(rng.integers(1, 9))) for s, l in zip(shp, lo)]
            grid[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = col
        want = oracle.recolor_backward_components(grid, col, new, k=3, sort_axis=1)
        assert np.array_equal(pb3d_gpu.recolor_backward_components(grid, col, new, k=3, sort_axis=1), want), shp
        d_g = dev.from_numpy(grid); d_a = dev.from_numpy(grid); d_b = dev.from_numpy(grid); d_lab = dev.DeviceBuffer(grid.size // 3 * 4)
        n, _, cnt, sums = _label_stats(d_g, shp, col, d_lab)
        flags = (rng.random(n) < 0.5).astype(np.uint8)
        nvox = grid.size // 3
        L.check(L.load().pb3d_recolor_components_dev(L.ctx(), C.c_void_p(d_lab.ptr), nvox, L.p_u8(flags), n, L.p_u8(new), C.c_void_p(d_a.ptr)))
        L.check(L.load().pb3d_recolor_last_labelled_dev(L.ctx(), C.c_void_p(d_lab.ptr), nvox, L.p_u8(flags), n, L.p_u8(new), C.c_void_p(d_b.ptr), 3))
        assert np.array_equal(d_a.download(grid.shape), d_b.download(grid.shape)), shp
        with pytest.raises(ValueError, match="last pb3d_label"):
            L.check(L.load().pb3d_recolor_last_labelled_dev(L.ctx(), C.c_void_p(d_a.ptr), nvox, L.p_u8(flags), n, L.p_u8(new), C.c_void_p(d_b.ptr), 3))
        for b in (d_g, d_a, d_b, d_lab):
            b.free()


def test_rot90_flat_ragged_streams(pb3d_gpu, oracle):
    """the flat 90-degree kernel on x-row streams that are whole 16-byte pieces but not whole lines (H * D % 16 == 0, % 128 != 0, e.g.
    500 x 400 x 500): ragged last segment, rows of odd x starting mid-line -- against the tile kernel (tune misc2 = 4) and the oracle."""
    rng = np.random.default_rng(43)
    for (W, H, D) in [(200, 12, 204), (260, 10, 136), (131, 4, 140), (300, 8, 250), (150, 24, 202)]:
        assert (H * D) % 16 == 0 and (H * D) % 128 != 0 and D >= 128
        for kind in ("binary", "bytes"):
            g = (rng.random((W, H, D)) < 0.5).astype(np.uint8) if kind == "binary" else rng.integers(0, 256, (W, H, D), dtype=np.uint8)
            m = rng.random((H, W)) < 0.85
            want = oracle.process_voxel_grid(g, m, 90)
            for knob, wide, mblk in ((0, 0, 0), (0, 0, 1), (0, 2, 0), (2, 0, 0)):   # 256-byte segments (k_rot90wf; mask block on / off), 128-byte segments (k_rot90_flat), tile kernel
                pb3d_gpu._lib.set_tuning("rot90_flat", knob); pb3d_gpu._lib.set_tuning("rot90_wide", wide); pb3d_gpu._lib.set_tuning("rot90_mask_block", mblk)
                try:
                    got = pb3d_gpu.process_voxel_grid(g, m, 90)
                finally:
                    pb3d_gpu._lib.set_tuning("rot90_flat", 0); pb3d_gpu._lib.set_tuning("rot90_wide", 0); pb3d_gpu._lib.set_tuning("rot90_mask_block", 0)
                assert np.array_equal(got, want), (W, H, D, kind, knob, wide, mblk, int((got != want).sum()))
    # whole-line streams of odd row length (H * D % 128 == 0): both segment widths
    for (W, H, D) in [(355, 128, 355), (259, 128, 259), (437, 128, 437), (180, 128, 182)]:
        g = (rng.random((W, H, D)) < 0.5).astype(np.uint8)
        m = rng.random((H, W)) < 0.85
        want = oracle.process_voxel_grid(g, m, 90)
        for wide, mblk in ((0, 0), (0, 1), (2, 0)):          # masks from the workgroup's LDS block / fetched per segment; 128-byte segments
            pb3d_gpu._lib.set_tuning("rot90_wide", wide); pb3d_gpu._lib.set_tuning("rot90_mask_block", mblk)
            try:
                got = pb3d_gpu.process_voxel_grid(g, m, 90)
            finally:
                pb3d_gpu._lib.set_tuning("rot90_wide", 0); pb3d_gpu._lib.set_tuning("rot90_mask_block", 0)
            assert np.array_equal(got, want), (W, H, D, wide, mblk, int((got != want).sum()))
    # the same streams through the six-job part_carve sweep (k_part90_flat)
    pal = np.array(list(pb3d_gpu.PART_COLORS.values()), np.uint8)
    for (W, H, D) in [(200, 12, 204), (150, 24, 202), (131, 4, 140)]:
        lab = rng.integers(1, len(pal), (H // 4 + 1, W // 4 + 1)).repeat(4, 0).repeat(4, 1)[:H, :W]
        sem = pal[lab]
        col = pal[rng.integers(1, len(pal), (W, H, D))] * (rng.random((W, H, D, 1)) < 0.6).astype(np.uint8)
        want = oracle.part_carve(col, sem, JOBS_NB1)
        got = pb3d_gpu.part_carve(col, sem, JOBS_NB1)
        assert np.array_equal(got, want), (W, H, D, int((got != want).any(-1).sum()))
