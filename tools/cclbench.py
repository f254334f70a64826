"""Labelling + statistics of the part colours of the notebook-1 chain on the carved Taj @ 512 grid (what left_right_guided_carve does
first): per colour, and all four in ONE labelling sequence (pb3d_label_colors_stats_dev); knob ccl_blocks = workgroups per CU of the
labelling's last pass (0: eight)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd"))
import pb3d  # noqa: E402
from pb3d import device as dev  # noqa: E402
from pb3d.voxel_carving_utils import _label_stats, _label, _component_stats, _label_stats_multi  # noqa: E402
g = {k: v for k, v in np.load(os.path.join(ROOT, "tests", "golden", "f9_Taj_512_masks.npz")).items()}
jobs = [(["full_building"], 90), (["chhatris"], 90), (["plinth"], 90), (["front_minarets"], 90), (["small_minarets"], 90), (["dome"], 90)]
d_gc = pb3d.global_carve(g["binary"], g["ext"], angle_interval=90, on_device=True)
d_pc = pb3d.part_carve(d_gc, g["ext"], jobs)
W, H, D, _ = d_pc.shape
d_lab = dev.DeviceBuffer(W * H * D * 4)


def t(fn, reps=10):
    fn(); dev.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dev.sync(); return round((time.perf_counter() - t0) / reps * 1e3, 4)


parts = ("dome", "chhatris", "front_minarets", "plinth")
for part in parts:
    col = np.array(pb3d.PART_COLORS[part], np.uint8)
    row = {"part": part}
    row["label+stats_ms"] = t(lambda: _label_stats(d_pc.buf, (W, H, D), col, d_lab))
    row["label+stats_members_only_ms"] = t(lambda: _label_stats(d_pc.buf, (W, H, D), col, d_lab, members_only=True))
    n = _label(d_pc.buf, (W, H, D), col, d_lab)
    row["components"] = n
    row["label_only_ms"] = t(lambda: _label(d_pc.buf, (W, H, D), col, d_lab))
    row["separate_stats_ms"] = t(lambda: _component_stats(d_lab, (W, H, D), n)) if n else None
    print(json.dumps(row), flush=True)
cols = [np.array(pb3d.PART_COLORS[p], np.uint8) for p in parts]
row = {"part": "+".join(parts), "colours_in_one_labelling": len(parts)}
for blocks in (0, 2, 4, 16):
    pb3d._lib.set_tuning("ccl_blocks", blocks)
    row[f"label+stats_members_only_ms_blocks{blocks}"] = t(lambda: _label_stats_multi(d_pc.buf, (W, H, D), cols, d_lab, members_only=True))
pb3d._lib.set_tuning("ccl_blocks", 0)
pb3d._lib.set_tuning("ccl_merge", 1)
row["label+stats_members_only_ms_pairwise_merge"] = t(lambda: _label_stats_multi(d_pc.buf, (W, H, D), cols, d_lab, members_only=True))
pb3d._lib.set_tuning("ccl_merge", 0)
row["components"] = [r[0] for r in _label_stats_multi(d_pc.buf, (W, H, D), cols, d_lab, members_only=True)]
print(json.dumps(row), flush=True)
