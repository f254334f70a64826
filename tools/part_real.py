"""part_carve (six 90-degree jobs) on the reference's REAL masks (Taj at max_dim 512), device resident, under development knobs:
python tools/part_real.py --variants ";part90_inflight=11;part90_inflight=22" """
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd"))
import numpy as np  # noqa: E402
import pb3d  # noqa: E402
from pb3d import device as dev  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default=";")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
g = {k: v for k, v in np.load(os.path.join(ROOT, "tests", "golden", "f9_Taj_512_masks.npz")).items()}
jobs = [(["full_building"], 90), (["chhatris"], 90), (["plinth"], 90), (["front_minarets"], 90), (["small_minarets"], 90), (["dome"], 90)]
d_gc = pb3d.global_carve(g["binary"], g["ext"], angle_interval=90, on_device=True)
names = [v for v in a.variants.split(";")]
sets = [dict(kv.split("=") for kv in v.split(",") if kv) for v in names]
keys = sorted({k for st in sets for k in st})
res = {}
for r in range(a.rounds):
    for v, st in zip(names, sets):
        for k in keys:
            pb3d._lib.set_tuning(k, int(st.get(k, 0)))
        pb3d.part_carve(d_gc, g["ext"], jobs); dev.sync()
        e0, e1 = dev.Event(), dev.Event()
        e0.record()
        for _ in range(a.reps):
            pb3d.part_carve(d_gc, g["ext"], jobs)
        e1.record(); dev.sync()
        res.setdefault(v or "default", []).append(round(e1.elapsed_ms_since(e0) / a.reps, 4))
for k in keys:
    pb3d._lib.set_tuning(k, 0)
print(json.dumps({"grid": list(d_gc.shape), "ms_by_variant": res}))
