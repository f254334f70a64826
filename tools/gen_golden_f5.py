"""F5: golden vectors for the component-guided / extrusion / recolour stages and the whole partwise_carve
(reference utils/voxel_carving_utils.py:163-266, :302-400), captured from the live reference (THIS CONTAINER ONLY)."""
import contextlib
import hashlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mask_ingest  # noqa: E402
import ref_import  # noqa: E402

vc, vu, pu, cg, ce, cfg = ref_import.load_reference()
vc.tqdm = lambda it, **k: it
PC, PCN = cfg.PART_COLORS, cfg.PART_COLORS_NP
OUT = os.path.join(ROOT, "tests", "golden")
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

group_jobs = [(["full_building"], 90), (["chhatris"], 90), (["plinth"], 90), (["front_minarets"], 90), (["small_minarets"], 90), (["dome"], 90)]
part_symmetry = {"dome": 5, "chhatris": 45, "front_minarets": 5, "small_minarets": 5}
extrusion_depths = {"main_door": 20, "windows": 10}

meta = {}
for mon, dim in [("Taj", 96), ("Akbar", 64), ("Bibi", 80)]:
    sem, ext, binary = mask_ingest.load_and_prepare(mon, dim, PC)
    gc = vc.global_carve(binary, ext, 90)
    pc = vc.part_carve(gc, ext, group_jobs)
    d = {"sem": sem, "ext": ext, "binary": binary}
    m = {"stages": {}}
    grid = pc
    logs = []
    for part, angle in part_symmetry.items():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            grid = vc.left_right_guided_carve(grid, ext, PCN[part], angle=angle)
        logs.append(buf.getvalue())
        m["stages"][f"lrgc_{part}"] = sha(grid)
    d["after_lrgc"] = grid
    m["lrgc_stdout"] = logs
    # extrusion, the four directions, door then windows (reference :356-373)
    for part, depth in extrusion_depths.items():
        mk = np.all(sem == PCN[part], axis=-1)
        for ax, dr in [(2, "+"), (2, "-"), (0, "+"), (0, "-")]:
            grid = vc.extrude_from_surface(grid, mk, axis=ax, direction=dr, depth=depth, fill_color=PCN[part])
            m["stages"][f"extrude_{part}_{ax}{dr}"] = sha(grid)
    d["after_extrude"] = grid
    oriented = np.flip(grid.transpose(2, 1, 0, 3), axis=1)
    rec = vc.recolor_backward_components(oriented, PCN["front_minarets"], new_color=PCN["back_minarets"], k=2, sort_axis=0)
    d["after_recolor"] = rec
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        full = vc.partwise_carve(gc, ext, sem, PCN, group_jobs, part_symmetry, extrusion_depths)
    assert np.array_equal(full, rec)
    m["partwise_sha256"] = sha(full); m["partwise_shape"] = list(full.shape)
    # extra single-function cases
    d["extrude_none"] = vc.extrude_from_surface(pc, np.all(sem == PCN["full_building"], axis=-1), axis=2, direction="-", depth=3, fill_color=None)
    d["recolor_k1_axis2"] = vc.recolor_backward_components(pc, PCN["front_minarets"], new_color=PCN["windows"], k=1, sort_axis=2)
    lab, n = __import__("scipy.ndimage").ndimage.label(np.all(pc == PCN["front_minarets"], axis=-1))
    m["label_front_minarets"] = {"n": int(n), "sha256": sha(lab.astype(np.int32))}
    lab, n = __import__("scipy.ndimage").ndimage.label(np.all(pc == PCN["full_building"], axis=-1))
    m["label_full_building"] = {"n": int(n), "sha256": sha(lab.astype(np.int32))}
    np.savez_compressed(os.path.join(OUT, f"f5_{mon}_{dim}.npz"), **d)
    meta[f"{mon}_{dim}"] = m
    print(mon, dim, full.shape, os.path.getsize(os.path.join(OUT, f"f5_{mon}_{dim}.npz")) // 1024, "KB", m["label_front_minarets"], [len(x) for x in logs])
# a synthetic label case with many small components (salt noise) incl. touching ones
rng = np.random.default_rng(3)
g = np.zeros((24, 20, 28, 3), np.uint8)
g[rng.random((24, 20, 28)) < 0.35] = PC["dome"]
import scipy.ndimage  # noqa: E402
lab, n = scipy.ndimage.label(np.all(g == PCN["dome"], axis=-1))
np.savez_compressed(os.path.join(OUT, "f5_label_noise.npz"), grid=g, labels=lab.astype(np.int32), n=np.int64(n))
json.dump(meta, open(os.path.join(OUT, "f5_meta.json"), "w"), indent=1)
print("noise comps", n)
