"""Capture golden input/output vectors from the LIVE reference (THIS CONTAINER ONLY).

Imports the reference's own Python through tools/ref_import.py, runs its hot-path
functions on seeded inputs and on masks derived from the reference's data files, and
writes the inputs + expected outputs as small .npz fixtures under tests/golden/.  The
fixtures are data only; no reference source or bytecode is stored.  tests/ replay them
against the CPU oracle (not gpu) and against the HIP path (gpu) without the reference.

    python tools/gen_golden.py            # regenerates everything (about a minute)
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np
import scipy.ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mask_ingest  # noqa: E402
import ref_import  # noqa: E402

vc, vu, pu, cg, ce, cfg = ref_import.load_reference()
vc.tqdm = lambda it, **k: it
PC = cfg.PART_COLORS
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(1791073527)
REF = ref_import.REFERENCE_ROOT


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB  %d arrays" % (name, os.path.getsize(path) / 1024, len(arrays)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---- F1: Rinv table + offsets -----------------------------------------------------------
shapes = np.array([(16, 5, 16), (37, 11, 37), (33, 7, 33), (21, 5, 34), (64, 9, 64), (128, 3, 128), (99, 116, 99),
                   (35, 50, 35), (31, 189, 31), (1, 1, 1), (2, 3, 1), (256, 139, 256), (512, 278, 512),
                   (1024, 1024, 1024), (355, 512, 355), (128, 79, 128), (128, 123, 128), (7, 7, 7)], np.int64)
rot = np.stack([vc._rotation_matrix_inv(a).ravel() for a in range(91)])
offs = np.stack([[np.array(s) / 2 - vc._rotation_matrix_inv(a) @ (np.array(s) / 2) for a in range(91)] for s in shapes])
save("f1_rotinv_offsets", rotinv_bits=rot.view(np.uint64), shapes=shapes, offsets_bits=offs.view(np.uint64))

# ---- affine_transform (SciPy as called at voxel_carving_utils.py:116-123) ---------------
d = {}
i = 0
for sh in [(16, 5, 16), (21, 5, 34), (33, 7, 33), (2, 3, 1), (1, 4, 7), (10, 3, 10)]:
    for kind in ("bin", "full"):
        g = (rng.random(sh) < 0.5).astype(np.uint8) if kind == "bin" else rng.integers(0, 256, sh, dtype=np.uint8)
        for a in (0, 5, 45, 60, 90):
            M = vc._rotation_matrix_inv(a)
            off = np.array(sh) / 2 - M @ (np.array(sh) / 2)
            d[f"in_{i}"] = g; d[f"angle_{i}"] = np.int64(a)
            d[f"out_{i}"] = scipy.ndimage.affine_transform(g, M, offset=off, order=1, mode="constant", cval=0)
            i += 1
d["n"] = np.int64(i)
save("f3_affine", **d)

# ---- F2: carve_voxel_grid_with_masks ----------------------------------------------------
d = {}
i = 0
for (W, H, D) in [(8, 5, 6), (7, 7, 3), (16, 9, 16), (1, 1, 1), (3, 2, 5), (12, 12, 4)]:
    for nd in (3, 4):
        g = rng.integers(0, 256, (W, H, D) + ((3,) if nd == 4 else ()), dtype=np.uint8)
        for mshape in sorted({(H, W), (W, H)}):
            for dt in ("bool", "uint8"):
                m = rng.random(mshape) < 0.6
                m = m if dt == "bool" else m.astype(np.uint8) * 7
                d[f"grid_{i}"] = g; d[f"mask_{i}"] = m
                d[f"out_{i}"] = vc.carve_voxel_grid_with_masks(g, m)
                i += 1
d["n"] = np.int64(i)
save("f2_carve", **d)

# ---- F3: process_voxel_grid -------------------------------------------------------------
d = {}
i = 0
for sh in [(16, 5, 16), (37, 11, 37), (33, 7, 33), (21, 5, 34), (64, 9, 64), (128, 3, 128), (12, 12, 12), (9, 9, 5)]:
    W, H, D = sh
    for kind in ("bin", "full", "ones"):
        g = {"bin": (rng.random(sh) < 0.5).astype(np.uint8), "full": rng.integers(0, 256, sh, dtype=np.uint8),
             "ones": np.ones(sh, np.uint8)}[kind]
        for ai in (90, 60, 45, 5):
            if kind == "full" and ai == 5 and W > 40:
                continue
            mshape = (H, W) if (i % 3) else (W, H)
            m = (rng.random(mshape) < 0.85)
            m = m if i % 2 else m.astype(np.uint8)
            d[f"grid_{i}"] = g; d[f"mask_{i}"] = m; d[f"ai_{i}"] = np.int64(ai)
            d[f"out_{i}"] = vc.process_voxel_grid(g, m, ai)
            i += 1
d["n"] = np.int64(i)
save("f3_process", **d)

# ---- F4: real masks -> global_carve / part_carve ----------------------------------------
jobs_nb1 = [(["full_building"], 90), (["chhatris"], 90), (["plinth"], 90), (["front_minarets"], 90),
            (["small_minarets"], 90), (["dome"], 90)]
jobs_mixed = [(["full_building", "plinth"], 90), (["chhatris"], 45), (["dome"], 60), (["front_minarets", "small_minarets"], 90)]
digests = {}
for mon, dim in [("Akbar", 64), ("Bibi", 64), ("Taj", 96)]:
    sem, ext, binary = mask_ingest.load_and_prepare(mon, dim, PC)
    gc = vc.global_carve(binary, ext, angle_interval=90)
    pc1 = vc.part_carve(gc, ext, jobs_nb1)
    pc2 = vc.part_carve(gc, ext, jobs_mixed)
    gc45 = vc.global_carve(binary, ext, angle_interval=45)
    save(f"f4_{mon}_{dim}", sem=sem, ext=ext, binary=binary, global_carve=gc, part_carve_nb1=pc1,
         part_carve_mixed=pc2, global_carve_45=gc45)
# a square case (W == H): exercises the double transpose inside part_carve (:24 wins when W == H)
sem, ext, binary = mask_ingest.load_and_prepare("Akbar", 64, PC)
sq = np.full((64, 64, 3), PC["background"], np.uint8)
sq[: ext.shape[0], : ext.shape[1]] = ext[:64, :64]
sqb = (~np.all(sq == np.array(PC["background"]), axis=-1)).astype(np.uint8)
gsq = vc.global_carve(sqb, sq, angle_interval=90)
save("f4_square_64", ext=sq, binary=sqb, global_carve=gsq, part_carve_nb1=vc.part_carve(gsq, sq, jobs_nb1))
for mon, dim in [("Akbar", 128), ("Bibi", 128), ("Taj", 256)]:
    sem, ext, binary = mask_ingest.load_and_prepare(mon, dim, PC)
    gc = vc.global_carve(binary, ext, angle_interval=90)
    pc1 = vc.part_carve(gc, ext, jobs_nb1)
    digests[f"{mon}_{dim}"] = {"shape": list(gc.shape), "global_carve_sha256": sha(gc), "part_carve_nb1_sha256": sha(pc1),
                               "occupied_global": int(np.any(gc > 0, -1).sum()), "occupied_part": int(np.any(pc1 > 0, -1).sum())}
    save(f"f4_{mon}_{dim}_masks", sem=sem, ext=ext, binary=binary)
# Taj@512 masks for the results/1 acceptance case (outputs are the stored artefact itself)
sem, ext, binary = mask_ingest.load_and_prepare("Taj", 512, PC)
save("f9_Taj_512_masks", sem=sem, ext=ext, binary=binary)
for fn in ("Akbar_voxel_grid.npz", "Taj_voxel_grid.npz", "Charminar_voxel_grid.npz"):
    shutil.copyfile(os.path.join(REF, "results", "1.Orthographic_Voxel_Carving", fn), os.path.join(OUT, "stored_" + fn))
for mon in ("Akbar", "Taj", "Charminar"):
    for tag in ("init", "kp", "final"):
        fn = f"{mon}_camera_params_{tag}.json"
        shutil.copyfile(os.path.join(REF, "results", "2.Perspective_Camera_Estimation", fn), os.path.join(OUT, "stored_" + fn))
with open(os.path.join(OUT, "f4_digests.json"), "w") as f:
    json.dump(digests, f, indent=1)

# ---- F6: points -------------------------------------------------------------------------
akbar = np.load(os.path.join(OUT, "stored_Akbar_voxel_grid.npz"))["voxel_grid"]
d = {}
pd = {}
for names in (["front_minarets"], ["full_building", "chhatris"], list(PC)):
    p, c = vu.get_voxel_points_by_parts(akbar, PC, names)
    pd["parts:" + ",".join(names)] = {"n": int(len(p)), "pts_sha256": sha(p), "cols_sha256": sha(c),
                                       "pts_dtype": str(p.dtype), "cols_dtype": str(c.dtype)}
for st in (1, 2, 3, 4):
    p, c, s = vu.voxel_grid_to_points(akbar, stride=st)
    pd[f"stride:{st}"] = {"n": int(len(p)), "pts_sha256": sha(p), "cols_sha256": sha(c), "shape": [int(v) for v in s]}
    if st == 4:  # one case kept in full (order + values visible in the fixture)
        d["s4_pts"] = p; d["s4_cols"] = c
p, c = vu.get_voxel_points_by_parts(akbar, PC, ["chhatris"])
d["chhatris_pts"] = p; d["chhatris_cols"] = c
with open(os.path.join(OUT, "f6_points_akbar.json"), "w") as f:
    json.dump(pd, f, indent=1)
save("f6_points_akbar", **d)

# ---- F7: projection + IoU with the stored cameras ---------------------------------------


def to_numpy(obj):  # notebook 3 cell 3 rule: lists -> float32 arrays, scalars stay Python floats
    if isinstance(obj, list):
        return np.array(obj, dtype=np.float32)
    if isinstance(obj, dict):
        return {k: to_numpy(v) for k, v in obj.items()}
    return obj


d = {}
i = 0
summary = {}
for mon in ("Akbar", "Charminar"):
    grid = np.load(os.path.join(OUT, f"stored_{mon}_voxel_grid.npz"))["voxel_grid"]
    cams = to_numpy(json.load(open(os.path.join(OUT, f"stored_{mon}_camera_params_final.json"))))
    max_dim = int(np.max(grid.shape))
    front = mask_ingest.nearest_resize(mask_ingest.load_rgb(mon, "front"), max_dim)
    drone = mask_ingest.load_rgb(mon, "drone")
    for view, img in (("front", front), ("drone", drone)):
        cam = cams[view]
        H, W = img.shape[:2]
        pts, col = vu.get_voxel_points_by_parts(grid, PC, list(PC))
        for mode in ("f32", "f64"):
            cp, tg = cam["cam_pos"].copy(), cam["target"].copy()
            f, cx, cy = cam["f"], cam["cx"], cam["cy"]
            if mode == "f64":
                cp, tg = cp.astype(np.float64), tg.astype(np.float64)
            proj = pu.project_colored_voxels(pts, col, cp.copy(), tg.copy(), f, cx, cy, H, W)
            per, mean = ce.compute_partwise_iou(proj, img, PC)
            key = f"{mon}_{view}_{mode}"
            d[f"proj_{key}"] = proj
            d[f"R_{key}"] = cg.look_at_rotation(cp.copy(), tg.copy())
            summary[key] = {"iou": {k: float(v) for k, v in per.items()}, "mean": float(mean), "H": H, "W": W,
                            "f": float(f), "cx": float(cx), "cy": float(cy), "npts": int(len(pts))}
        d[f"img_{mon}_{view}"] = img
        # per-part IoU as visualize_voxel_projection_iou computes it (camera_estimation.py:381-403)
        for part in ("full_building", "front_minarets", "back_minarets"):
            p1, c1 = vu.get_voxel_points_by_parts(grid, PC, [part])
            if len(p1) == 0:
                continue
            pj = pu.project_colored_voxels(p1, c1, cam["cam_pos"].copy(), cam["target"].copy(), cam["f"], cam["cx"], cam["cy"], H, W)
            a = np.all(img == PC[part], -1); b = np.all(pj == PC[part], -1)
            summary[f"{mon}_{view}_part_{part}"] = float((a & b).sum() / (a | b).sum()) if (a | b).sum() else 0.0
with open(os.path.join(OUT, "f7_projection_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
save("f7_projection", **d)

# synthetic projection cases incl. NumPy-2 mixed promotion, duplicates and out-of-frustum points
d = {}
i = 0
pal = np.array(list(PC.values()), np.uint8)
for mode in ("f32", "f64", "mixed_f", "mixed_c", "behind"):
    N = 20000
    A = 96
    pts = rng.integers(0, A, (N, 3)).astype(np.float32)
    cols = pal[rng.integers(0, 10, N)]
    Himg, Wimg = 90, 120
    cam = np.array([A / 2 + 3.0, A / 2 - 2.0, -2.0 * A]); tgt = np.array([A / 2, A / 2, A / 2 + 1.0])
    f, cx, cy = 170.5, 60.25, 44.75
    if mode != "f64":
        cam = cam.astype(np.float32); tgt = tgt.astype(np.float32)
    if mode == "mixed_f":
        f = np.float64(f)
    if mode == "mixed_c":
        cx = np.float64(cx); cy = np.float32(cy)
    if mode == "behind":
        cam = np.array([A / 2, A / 2, A / 2], np.float32); tgt = np.array([A / 2 + 1, A / 2, A], np.float32)
    d[f"pts_{i}"] = pts; d[f"cols_{i}"] = cols; d[f"cam_{i}"] = cam; d[f"tgt_{i}"] = tgt
    d[f"fcxcy_{i}"] = np.array([float(f), float(cx), float(cy)])
    d[f"ftypes_{i}"] = np.array([type(f).__name__, type(cx).__name__, type(cy).__name__])
    d[f"hw_{i}"] = np.array([Himg, Wimg])
    d[f"out_{i}"] = pu.project_colored_voxels(pts, cols, cam.copy(), tgt.copy(), f, cx, cy, Himg, Wimg)
    i += 1
d["n"] = np.int64(i)
save("f7_projection_synth", **d)
print("done")
