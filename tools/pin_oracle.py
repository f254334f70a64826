"""Pin the CPU oracle against the LIVE reference (THIS CONTAINER ONLY).

Runs the reference's own functions (imported through tools/ref_import.py) and SciPy
itself next to oracle/oracle.py on randomised inputs and asserts bit equality.  This is
the wide, slow counterpart of tests/test_oracle_golden.py (which replays a committed
subset as fixtures and needs no reference).  Usage: python tools/pin_oracle.py [--quick]
"""
import os
import sys
import time

import numpy as np
import scipy.ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402
from oracle import oracle as orc  # noqa: E402

vc, vu, pu, cg, ce, cfg = ref_import.load_reference()
import tqdm as _tqdm  # noqa: E402
vc.tqdm = lambda it, **k: it  # silence progress bars

quick = "--quick" in sys.argv
rng = np.random.default_rng(20261004)
nchk = 0


def eq(a, b, what):
    global nchk
    nchk += 1
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert a.dtype == b.dtype, (what, a.dtype, b.dtype)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError(f"{what}: {len(bad)} mismatches, first at {bad[0]}: {a[tuple(bad[0])]} vs {b[tuple(bad[0])]}")


t0 = time.time()
# ---- A3 / offsets ---------------------------------------------------------------------
for a in range(91):
    eq(orc.rotation_matrix_inv(a).view(np.uint64), vc._rotation_matrix_inv(a).view(np.uint64), f"rotinv {a}")
shapes = [(16, 5, 16), (37, 11, 37), (33, 7, 33), (21, 5, 34), (64, 9, 64), (128, 3, 128), (99, 116, 99),
          (35, 50, 35), (31, 189, 31), (1, 1, 1), (2, 3, 1), (1, 4, 7), (256, 139, 256), (512, 278, 512),
          (1024, 1024, 1024), (355, 512, 355), (7, 7, 7), (10, 3, 10)]
for sh in shapes:
    c = np.array(sh) / 2
    for a in range(91):
        M = vc._rotation_matrix_inv(a)
        eq(orc.affine_offset(M, sh).view(np.uint64), (c - M @ c).view(np.uint64), f"offset {sh} {a}")
print("rotinv/offset ok", nchk)

# ---- affine_transform vs SciPy --------------------------------------------------------
aff_shapes = [(16, 5, 16), (37, 11, 37), (33, 7, 33), (21, 5, 34), (64, 9, 64), (1, 1, 1), (2, 3, 1), (1, 4, 7),
              (5, 1, 9), (10, 3, 10), (128, 2, 128)]
angles = [0, 1, 5, 10, 30, 45, 60, 85, 89, 90]
for sh in aff_shapes:
    for kind in ("bin", "full", "ones"):
        if kind == "bin":
            g = (rng.random(sh) < 0.5).astype(np.uint8)
        elif kind == "full":
            g = rng.integers(0, 256, sh, dtype=np.uint8)
        else:
            g = np.ones(sh, np.uint8)
        for a in angles:
            M = vc._rotation_matrix_inv(a)
            off = np.array(sh) / 2 - M @ (np.array(sh) / 2)
            ref = scipy.ndimage.affine_transform(g, M, offset=off, order=1, mode="constant", cval=0)
            eq(orc.affine_transform_u8(g, M, off), ref, f"affine {sh} {kind} {a}")
print("affine ok", nchk)

# ---- A4 carve -------------------------------------------------------------------------
for (W, H, D) in [(8, 5, 6), (7, 7, 3), (16, 9, 16), (1, 1, 1), (3, 2, 5)]:
    for nd in (3, 4):
        g = rng.integers(0, 256, (W, H, D) + ((3,) if nd == 4 else ()), dtype=np.uint8)
        for mshape in {(H, W), (W, H)}:
            for dt in (bool, np.uint8, np.float32):
                m = (rng.random(mshape) < 0.6)
                m = m.astype(dt) * (3 if dt != bool else 1)
                eq(orc.carve_voxel_grid_with_masks(g, m), vc.carve_voxel_grid_with_masks(g, m), f"carve {W,H,D} {nd} {mshape} {dt}")
            if nd == 4 and W > 1 and H > 1:
                # the reference's RGB-mask branch (:90-95) cannot broadcast for any non-degenerate
                # shape: it always ends in ValueError; the oracle front end mirrors that.
                m3 = rng.integers(0, 2, mshape + (3,), dtype=np.uint8) * 200
                for f in (orc.carve_voxel_grid_with_masks, vc.carve_voxel_grid_with_masks):
                    try:
                        f(g, m3)
                        raise SystemExit("expected ValueError for RGB mask")
                    except ValueError:
                        nchk += 1
    for bad in [(H + 1, W), (W, H + 2)]:
        for f in (orc.carve_voxel_grid_with_masks, vc.carve_voxel_grid_with_masks):
            try:
                f(np.zeros((W, H, D), np.uint8), np.zeros(bad, np.uint8))
                raise SystemExit("expected ValueError")
            except ValueError:
                pass
print("carve ok", nchk)

# ---- A5 process_voxel_grid ------------------------------------------------------------
pshapes = [(16, 5, 16), (37, 11, 37), (33, 7, 33), (21, 5, 34), (64, 9, 64), (12, 12, 12), (9, 9, 5)]
if not quick:
    pshapes += [(128, 3, 128), (99, 20, 99), (31, 40, 31)]
for sh in pshapes:
    W, H, D = sh
    for kind in ("bin", "full", "ones"):
        g = {"bin": (rng.random(sh) < 0.5).astype(np.uint8), "full": rng.integers(0, 256, sh, dtype=np.uint8),
             "ones": np.ones(sh, np.uint8)}[kind]
        for ai in (90, 60, 45, 30, 5, 7, 91, 100):
            for mshape in {(H, W), (W, H)}:
                m = rng.random(mshape) < 0.8
                eq(orc.process_voxel_grid(g, m, ai), vc.process_voxel_grid(g, m, ai), f"process {sh} {kind} {ai} {mshape}")
print("process ok", nchk)

# ---- A2/A6/A7/A8 ----------------------------------------------------------------------
pal = np.array(list(cfg.PART_COLORS.values()), np.uint8)


def rand_sem(h, w, nlab=10, extra=False):
    lab = rng.integers(0, nlab, (h // 4 + 1, w // 4 + 1))
    lab = np.kron(lab, np.ones((4, 4), int))[:h, :w]
    sem = pal[lab]
    if extra:  # sprinkle non-palette colours
        k = rng.random((h, w)) < 0.05
        sem[k] = rng.integers(0, 256, (int(k.sum()), 3), dtype=np.uint8)
    return sem


bg = np.array(cfg.PART_COLORS["background"], np.uint8)
gc_sizes = [(9, 16), (16, 16), (21, 13), (40, 64)] + ([] if quick else [(79, 128)])
for (h, w) in gc_sizes:
    for extra in (False, True):
        sem = rand_sem(h, w, extra=extra)
        binary = (~np.all(sem == bg, axis=-1)).astype(np.uint8)
        for ai in (90, 45):
            ref = vc.global_carve(binary, sem, angle_interval=ai)
            got = orc.global_carve(binary, sem, angle_interval=ai)
            eq(got, ref, f"global_carve {h,w} {ai} {extra}")
        eq(orc.occupancy(ref), vc._occupancy(ref), "occupancy")
        carved = (rng.random((w, h, w)) < 0.5).astype(np.uint8) * rng.integers(1, 3, (w, h, w), dtype=np.uint8)
        eq(orc.apply_colored_mask_to_voxel_grid(carved, sem), vc.apply_colored_mask_to_voxel_grid(carved, sem), "color_apply")
        jobs = [(["full_building"], 90), (["chhatris", "dome"], 90), (["plinth"], 60), (["front_minarets"], 90),
                (["small_minarets"], 45), (["windows"], 90)]
        # a grid with arbitrary colours (not only the mask's) exercises the `sub`/`carved` frames
        colored = ref.copy()
        k = rng.random(colored.shape[:3]) < 0.1
        colored[k] = rng.integers(0, 256, (int(k.sum()), 3), dtype=np.uint8)
        eq(orc.part_carve(colored, sem, jobs), vc.part_carve(colored, sem, jobs), f"part_carve {h,w}")
print("global/part carve ok", nchk)

# ---- A15/A16 --------------------------------------------------------------------------
for sh in [(9, 7, 11), (16, 16, 16), (1, 1, 1), (5, 6, 2)]:
    lab = rng.integers(0, 12, sh)
    pal12 = np.vstack([pal, [[0, 0, 0], [9, 9, 9]]]).astype(np.uint8)
    grid = pal12[lab]
    for names in (["dome"], ["dome", "plinth", "windows"], list(cfg.PART_COLORS), []):
        rp, rc = vu.get_voxel_points_by_parts(grid, cfg.PART_COLORS, names)
        op, oc = orc.get_voxel_points_by_parts(grid, cfg.PART_COLORS, names)
        if len(names):
            eq(op, rp, f"points pts {sh} {names}"); eq(oc, rc, "points cols")
        else:
            assert rp.shape == op.shape == (0, 3)
    for st in (1, 2, 3, 4):
        rp, rc, rs = vu.voxel_grid_to_points(grid, stride=st)
        op, oc, os_ = orc.voxel_grid_to_points(grid, stride=st)
        eq(op, rp, f"vg2p pts {sh} {st}"); eq(oc, rc, "vg2p cols"); assert rs == os_
print("points ok", nchk)

# ---- A13/A14/A17 ----------------------------------------------------------------------
for trial in range(6 if quick else 20):
    N = int(rng.integers(2000, 30000))
    A = int(rng.integers(32, 128))
    pts = rng.integers(0, A, (N, 3)).astype(np.float32)
    cols = pal[rng.integers(0, 10, N)]
    Himg, Wimg = int(rng.integers(40, 200)), int(rng.integers(40, 200))
    for mode in ("f32", "f64", "mixed_f", "mixed_c"):
        cam = np.array([A / 2 + rng.normal() * 5, A / 2 + rng.normal() * 5, -2.0 * A + rng.normal() * 10])
        tgt = np.array([A / 2, A / 2, A / 2]) + rng.normal(size=3)
        f, cx, cy = float(1.5 * Wimg + rng.normal()), Wimg / 2 + float(rng.normal()), Himg / 2 + float(rng.normal())
        if mode == "f32":
            cam = cam.astype(np.float32); tgt = tgt.astype(np.float32)
        elif mode == "mixed_f":
            cam = cam.astype(np.float32); tgt = tgt.astype(np.float32); f = np.float64(f)
        elif mode == "mixed_c":
            cam = cam.astype(np.float32); tgt = tgt.astype(np.float32); cx = np.float64(cx); cy = np.float32(cy)
        eq(orc.look_at_rotation(cam, tgt), cg.look_at_rotation(cam.copy(), tgt.copy()), "look_at")
        ref = pu.project_colored_voxels(pts, cols, cam.copy(), tgt.copy(), f, cx, cy, Himg, Wimg)
        got = orc.project_colored_voxels(pts, cols, cam, tgt, f, cx, cy, Himg, Wimg)
        eq(got, ref, f"project {mode} N={N}")
        img2 = pal[rng.integers(0, 10, (Himg, Wimg))]
        rper, rmean = ce.compute_partwise_iou(ref, img2, cfg.PART_COLORS)
        oper, omean = orc.compute_partwise_iou(got, img2, cfg.PART_COLORS)
        assert rper == oper and rmean == omean, "iou"
        nchk += 1
print("project/iou ok", nchk)
print("ALL PINNED: %d checks in %.1fs" % (nchk, time.time() - t0))
