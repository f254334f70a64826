"""Device-resident timing of the 90-degree ops on the reference's real grid shapes (odd / unaligned dims).
python tools/shapebench.py  -> one JSON line per (shape, op)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pb3d  # noqa: E402
from pb3d import device as dev  # noqa: E402


def timeit(fn, reps=20, rounds=3):
    """ms per call: `reps` calls back to back between two events (at 30 - 60 us a call, five were not enough for the host's launch
    rate to drop out), best of `rounds`."""
    fn(); dev.sync()
    best = None
    for _ in range(rounds):
        e0, e1 = dev.Event(), dev.Event()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); dev.sync()
        ms = e1.elapsed_ms_since(e0) / reps
        best = ms if best is None or ms < best else best
    return best


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="128x123x128,512x278x512,355x512x355,512x512x512,437x512x437,500x400x500")
    ap.add_argument("--tune", default="", help="development knobs, e.g. rot90_flat=1,sliced=1 (pb3d_set_tuning)")
    a = ap.parse_args()
    tune = dict(kv.split("=") for kv in a.tune.split(",") if kv)
    for k, v in tune.items():
        pb3d._lib.set_tuning(k, int(v))
    rng = np.random.default_rng(5)
    shapes = [tuple(int(v) for v in sh.split("x")) for sh in a.shapes.split(",")]
    for (W, H, D) in shapes:
        nvox = W * H * D
        m_hw = (rng.random((H, W)) < 0.8)
        rgb = rng.integers(1, 255, (H, W, 3), dtype=np.uint8)
        d_mwh = dev.from_numpy(np.ascontiguousarray(m_hw.T).view(np.uint8))
        d_bhw = dev.from_numpy(np.ascontiguousarray(m_hw).view(np.uint8))
        d_rgb = dev.from_numpy(rgb)
        occ = (rng.random((W, H, D)) < 0.6).astype(np.uint8)
        d_occ = dev.from_numpy(occ); d_o = dev.DeviceBuffer(nvox); d_t = dev.DeviceBuffer(nvox); d_col = dev.DeviceBuffer(nvox * 3)
        sem = rng.integers(0, 255, (W, H, D, 3), dtype=np.uint8)
        d_sem = dev.from_numpy(sem); d_sem2 = dev.DeviceBuffer(nvox * 3)
        rows0 = {"carve_voxel_grid_with_masks(sem)": (timeit(lambda: dev.carve_mask(d_sem, W, H, D, 3, d_mwh, d_sem2)), 6),
                 "carve_voxel_grid_with_masks(occ)": (timeit(lambda: dev.carve_mask(d_occ, W, H, D, 1, d_mwh, d_o)), 2),
                 "_occupancy": (timeit(lambda: dev.occupancy(d_sem, nvox, d_o)), 4),
                 "apply_colored_mask": (timeit(lambda: dev.color_apply(d_occ, W, H, D, d_rgb, d_sem2)), 4)}
        d_sem.free(); d_sem2.free()
        rows = {"process_voxel_grid(occ,90)": (timeit(lambda: dev.process_grid(d_occ, W, H, D, d_mwh, 90, d_o, d_t)), 2),
                "process_voxel_grid(occ,45) per step": (timeit(lambda: dev.process_grid(d_occ, W, H, D, d_mwh, 45, d_o, d_t)) / 2, 2),
                "global_carve(90)": (timeit(lambda: dev.global_carve(d_bhw, d_rgb, H, W, 90, d_col)), 3)}
        # part_carve with six 90-degree jobs: label image -> six disjoint sub-masks (the (W,H) images the C-ABI takes)
        import ctypes as C
        lab = rng.integers(0, 7, (W, H), dtype=np.uint8) * m_hw.T
        msub = np.stack([(lab == j + 1).astype(np.uint8) for j in range(6)])
        d_ms = dev.from_numpy(msub); d_mc = dev.from_numpy(msub)
        dev.global_carve(d_bhw, d_rgb, H, W, 90, d_col)
        d_pout = dev.DeviceBuffer(nvox * 3)
        L, lib = pb3d._lib, pb3d._lib.load()
        ang = (C.c_int * 6)(*([90] * 6)); skip = (C.c_int * 6)(*([0] * 6))
        rows0["part_carve(6 x 90 deg)"] = (timeit(lambda: L.check(lib.pb3d_part_carve_dev(L.ctx(), C.c_void_p(d_col.ptr), W, H, D, C.c_void_p(d_ms.ptr),
                                                                                        C.c_void_p(d_mc.ptr), ang, skip, 6, C.c_void_p(d_pout.ptr))), 10), 6)   # 6 B/voxel: what the ONE fused sweep moves (PMC)
        ang2 = (C.c_int * 6)(90, 45, 90, 60, 90, 90)
        rows0["part_carve(4 x 90, 45, 60 deg)"] = (timeit(lambda: L.check(lib.pb3d_part_carve_dev(L.ctx(), C.c_void_p(d_col.ptr), W, H, D, C.c_void_p(d_ms.ptr),
                                                                                                 C.c_void_p(d_mc.ptr), ang2, skip, 6, C.c_void_p(d_pout.ptr))), 5), 6)
        d_ms.free(); d_mc.free(); d_pout.free()
        rows.update(rows0)
        for name, (ms, bpv) in rows.items():
            print(json.dumps({"shape": [W, H, D], "op": name, "ms": round(ms, 4), "Mvoxel_s": round(nvox / ms / 1e3, 1),
                              "alg_GB_s": round(bpv * nvox / ms / 1e6, 1), **({"tune": tune} if tune else {})}), flush=True)
        for b in (d_mwh, d_bhw, d_rgb, d_occ, d_o, d_t, d_col):
            b.free()


if __name__ == "__main__":
    main()
