"""Fixture for process_voxel_grid on grids that are not uint8: the REFERENCE's own function (utils/voxel_carving_utils.py:104-126, which hands
whatever dtype it gets to scipy.ndimage.affine_transform and np.where) run on small seeded grids of every dtype SciPy's interpolation takes.
THIS CONTAINER ONLY (imports the reference).  -> tests/golden/f12_process_typed.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402

vc = ref_import.load_reference()[0]
import tqdm  # noqa: E402
vc.tqdm = lambda it, **k: it           # (no progress bars in the log)

rng = np.random.default_rng(12)
out = {}
DT = ("bool", "int8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64", "complex64", "complex128")
cases = []
for dt in DT:
    for (W, H, D), ang in (((13, 5, 13), 30), ((16, 3, 12), 45), ((9, 4, 17), 90), ((11, 2, 11), 7)):
        if dt == "bool":
            g = rng.random((W, H, D)) < 0.55
        elif dt.startswith("complex"):
            g = ((rng.random((W, H, D)) * 500 - 250) + 1j * (rng.random((W, H, D)) * 8 - 4)).astype(dt)
        elif dt.startswith("float"):
            g = (rng.random((W, H, D)) * 500 - 250).astype(dt)
        elif dt.startswith("u"):
            g = (rng.random((W, H, D)) * min(float(np.iinfo(dt).max), 2.0 ** 45)).astype(dt)
        else:
            g = ((rng.random((W, H, D)) - 0.5) * 2 * min(float(np.iinfo(dt).max), 2.0 ** 45)).astype(dt)
        g[rng.random((W, H, D)) < 0.3] = 0
        m = rng.random((H, W)) < 0.8
        want = vc.process_voxel_grid(g.copy(), m, ang)
        assert want.dtype == (np.int64 if dt == "bool" else g.dtype), (dt, want.dtype)
        k = f"{dt}_{W}x{H}x{D}_{ang}"
        out[k + "_in"] = g; out[k + "_mask"] = m; out[k + "_out"] = want
        cases.append(k)
out["cases"] = np.array(cases)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "f12_process_typed.npz"), **out)
print("written", len(cases), "cases")
