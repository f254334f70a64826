"""Where the WALL time of the resident notebook-1 chain goes on the host: cProfile of global_carve + part_carve + partwise_carve on
device-resident handles (Taj @ 512).  Blocking ctypes calls show up under their own names."""
import contextlib, cProfile, io, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd"))
import pb3d  # noqa: E402
from pb3d import device as dev  # noqa: E402
GOLDEN = os.path.join(ROOT, "tests", "golden")
group_jobs = [(["full_building"], 90), (["chhatris"], 90), (["plinth"], 90), (["front_minarets"], 90), (["small_minarets"], 90), (["dome"], 90)]
part_symmetry = {"dome": 5, "chhatris": 45, "front_minarets": 5, "small_minarets": 5}
extrusion_depths = {"main_door": 20, "windows": 10}
g = {k: v for k, v in np.load(os.path.join(GOLDEN, "f9_Taj_512_masks.npz")).items()}
PCN = pb3d.PART_COLORS_NP


def chain():
    d_gc = pb3d.global_carve(g["binary"], g["ext"], angle_interval=90, on_device=True)
    d_pc = pb3d.part_carve(d_gc, g["ext"], group_jobs)
    with contextlib.redirect_stdout(io.StringIO()):
        d_full = pb3d.partwise_carve(d_gc, g["ext"], g["sem"], PCN, group_jobs, part_symmetry, extrusion_depths)
    dev.sync()
    for d in (d_gc, d_pc, d_full):
        d.free()


for _ in range(3):
    chain()
t0 = time.perf_counter()
for _ in range(5):
    chain()
print("chain ms:", round((time.perf_counter() - t0) / 5 * 1e3, 2))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    chain()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
