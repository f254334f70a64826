"""Wall time of the NumPy-signature host API (PCIe-inclusive): what a notebook user of the drop-in sees.
python tools/hostbench.py [--size 512]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pb3d  # noqa: E402


def wall(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=512); a = ap.parse_args()
    S = a.size
    rng = np.random.default_rng(3)
    sem = rng.integers(0, 255, (S, S, S, 3), dtype=np.uint8)
    occ = (sem[..., 0] > 100).astype(np.uint8)
    m = rng.random((S, S)) < 0.8
    rgb = rng.integers(1, 255, (S, S, 3), dtype=np.uint8)
    res = {}
    res["carve(sem)"] = wall(lambda: pb3d.carve_voxel_grid_with_masks(sem, m))
    res["carve(occ)"] = wall(lambda: pb3d.carve_voxel_grid_with_masks(occ, m))
    res["process_voxel_grid(occ,90)"] = wall(lambda: pb3d.process_voxel_grid(occ, m, 90))
    res["global_carve(90)"] = wall(lambda: pb3d.global_carve(m, rgb, angle_interval=90))
    res["memcpy 2x grid (numpy copy sem)"] = wall(lambda: sem.copy())
    for k, v in res.items():
        nb = {"carve(sem)": 6, "carve(occ)": 2, "process_voxel_grid(occ,90)": 2, "global_carve(90)": 3, "memcpy 2x grid (numpy copy sem)": 6}[k] * S ** 3
        print(json.dumps({"size": S, "op": k, "wall_s": round(v, 4), "host_GB_s": round(nb / v / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    main()
