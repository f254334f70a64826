"""End-to-end timing of notebook 1 (global_carve + partwise_carve, notebook parameters) through the NumPy-signature
host API -- i.e. INCLUDING H2D/D2H of every call (PCIe-inclusive; never the bench.py `value`).  Taj at max_dim 512 is the
case the reference's stored artefact results/1 pins; its CPU timings are in BASELINE.md (23.9 s + 149.8 s)."""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd"))
import pb3d  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
group_jobs = [(["full_building"], 90), (["chhatris"], 90), (["plinth"], 90), (["front_minarets"], 90), (["small_minarets"], 90), (["dome"], 90)]
part_symmetry = {"dome": 5, "chhatris": 45, "front_minarets": 5, "small_minarets": 5}
extrusion_depths = {"main_door": 20, "windows": 10}

g = {k: v for k, v in np.load(os.path.join(GOLDEN, "f9_Taj_512_masks.npz")).items()}     # materialised once (NpzFile re-reads per access)
stored = np.load(os.path.join(GOLDEN, "stored_Taj_voxel_grid.npz"))["voxel_grid"]
PCN = pb3d.PART_COLORS_NP
pb3d.global_carve(g["binary"][:64, :64].copy(), g["ext"][:64, :64].copy(), 90)   # context + kernels warm
res = {}
pool_mb = pb3d._hostmem._cap_bytes >> 20      # result pool (pb3d/_hostmem.py; PB3D_RESULT_POOL_MB=0 turns it off): steady state from the 3rd run
for rep in range(4 if pool_mb else 2):
    t0 = time.perf_counter()
    gc = pb3d.global_carve(g["binary"], g["ext"], angle_interval=90)
    t1 = time.perf_counter()
    pc = pb3d.part_carve(gc, g["ext"], group_jobs)
    t2 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        full = pb3d.partwise_carve(gc, g["ext"], g["sem"], PCN, group_jobs, part_symmetry, extrusion_depths)
    t3 = time.perf_counter()
    res = {"grid": list(gc.shape), "Mvoxel": round(gc.size / 3 / 1e6, 1), "global_carve_s": round(t1 - t0, 3), "part_carve_s": round(t2 - t1, 3),
           "partwise_carve_s": round(t3 - t2, 3), "result_pool_mb": pool_mb, "reference_cpu_s": {"global_carve": 23.9, "part_carve": 122.7, "partwise_carve": 149.8}}
# the same chain on device-resident handles: no upload or download of the volume between the stages, one download at the end
from pb3d import device as dev  # noqa: E402
chain_walls = []
for rep in range(10):                 # steady state: ten chains back to back, every one ends with a device sync; ONE download at the end
    dev.sync(); t0 = time.perf_counter()
    d_gc = pb3d.global_carve(g["binary"], g["ext"], angle_interval=90, on_device=True)
    d_pc = pb3d.part_carve(d_gc, g["ext"], group_jobs)
    with contextlib.redirect_stdout(io.StringIO()):
        d_full = pb3d.partwise_carve(d_gc, g["ext"], g["sem"], PCN, group_jobs, part_symmetry, extrusion_depths)
    dev.sync(); t1 = time.perf_counter()
    chain_walls.append(t1 - t0)
    if rep == 9:
        full_r = d_full.numpy()
        t2 = time.perf_counter()
        same = bool(np.array_equal(full_r, full))
    for d in (d_gc, d_pc, d_full):
        d.free()
cw = sorted(chain_walls[2:])
res["resident_chain"] = {"global+part+partwise_ms_median": round(cw[len(cw) // 2] * 1e3, 3), "global+part+partwise_ms_min": round(cw[0] * 1e3, 3),
                         "first_chain_ms": round(chain_walls[0] * 1e3, 3), "final_download_s": round(t2 - t1, 4), "equals_host_chain": same}
# partwise_carve alone on a resident grid: wall time per call and the number of times the host waits for the device inside it
d_gc = pb3d.global_carve(g["binary"], g["ext"], angle_interval=90, on_device=True)
walls, waits = [], []
for rep in range(12):
    dev.sync(); w0 = dev.sync_count(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        d_full = pb3d.partwise_carve(d_gc, g["ext"], g["sem"], PCN, group_jobs, part_symmetry, extrusion_depths)
    w1 = dev.sync_count(); dev.sync(); walls.append(time.perf_counter() - t0); waits.append(w1 - w0)
    d_full.free()
d_gc.free()
walls = sorted(walls[2:])
res["resident_partwise_carve"] = {"wall_ms_median": round(walls[len(walls) // 2] * 1e3, 3), "wall_ms_min": round(walls[0] * 1e3, 3), "host_waits_per_call": waits[-1]}
oriented = np.flip(pc.transpose(2, 1, 0, 3), axis=1)
eq = lambda grid, name: np.all(grid == np.array(pb3d.PART_COLORS[name], np.uint8), axis=-1)
res["results1_pinned_parts_exact"] = bool(all(np.array_equal(eq(oriented, p), eq(stored, p)) for p in ("plinth", "chhatris")))
res["final_shape"] = list(full.shape)
print(json.dumps(res))
