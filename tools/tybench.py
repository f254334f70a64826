"""Development A/B of the planes-per-workgroup rule of the 90-degree kernels: process_voxel_grid(occ, 90) under knob rot90_fill
(0 = the built-in rule), variants interleaved and repeated so that clock / box drift shows up as spread instead of as a winner.
python tools/tybench.py --shapes 1024x1024x1024,512x512x512 --fills 0,4,6,8,12 --rounds 3"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd"))
import numpy as np  # noqa: E402
import pb3d  # noqa: E402
from pb3d import device as dev  # noqa: E402


def timeit(fn, reps):
    fn(); dev.sync()
    e0, e1 = dev.Event(), dev.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); dev.sync()
    return e1.elapsed_ms_since(e0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="1024x1024x1024,512x512x512,437x512x437,355x512x355")
    ap.add_argument("--fills", default="0,4,6,8,12,16")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--wide", default="0", help="comma list of tune rot90_wide values to interleave with the fills (0 = the 256 x 256-tile kernel where it applies, 2 = the 128-tile kernel)")
    ap.add_argument("--op", default="process", help="process: process_voxel_grid(occ, 90); part: part_carve with six 90-degree jobs (needs --variants)")
    ap.add_argument("--variants", default="", help="instead of fills x wide: ';'-separated tuning sets, each 'k=v,k=v' ('' = defaults), interleaved")
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    for sh in a.shapes.split(","):
        W, H, D = (int(v) for v in sh.split("x"))
        nvox = W * H * D
        d_mwh = dev.from_numpy((rng.random((W, H)) < 0.8).astype(np.uint8))
        d_occ = dev.DeviceBuffer(nvox); d_o = dev.DeviceBuffer(nvox); d_t = dev.DeviceBuffer(nvox)
        dev.synth_occ(0, W, H, D, 0, d_occ)
        res = {}
        fn = lambda: dev.process_grid(d_occ, W, H, D, d_mwh, 90, d_o, d_t)
        if a.op == "part":
            import ctypes as C
            m_hw = rng.random((H, W)) < 0.8
            lab = rng.integers(0, 7, (W, H), dtype=np.uint8) * m_hw.T
            msub = np.stack([(lab == j + 1).astype(np.uint8) for j in range(6)])
            d_ms = dev.from_numpy(msub); d_col = dev.DeviceBuffer(nvox * 3); d_pout = dev.DeviceBuffer(nvox * 3)
            dev.global_carve(dev.from_numpy(np.ascontiguousarray(m_hw).view(np.uint8)), dev.from_numpy(rng.integers(1, 255, (H, W, 3), dtype=np.uint8)), H, W, 90, d_col)
            L, lib = pb3d._lib, pb3d._lib.load()
            ang = (C.c_int * 6)(*([90] * 6)); skip = (C.c_int * 6)(*([0] * 6))
            fn = lambda: L.check(lib.pb3d_part_carve_dev(L.ctx(), C.c_void_p(d_col.ptr), W, H, D, C.c_void_p(d_ms.ptr), C.c_void_p(d_ms.ptr), ang, skip, 6,
                                                         C.c_void_p(d_pout.ptr)))
        if a.op == "partlabel":      # the same six jobs on a 1-byte label volume (row N3)
            import ctypes as C
            m_hw = rng.random((H, W)) < 0.8
            lab = rng.integers(0, 7, (W, H), dtype=np.uint8) * m_hw.T
            msub = np.stack([(lab == j + 1).astype(np.uint8) for j in range(6)])
            d_ms = dev.from_numpy(msub); d_lab = dev.DeviceBuffer(nvox); d_lout = dev.DeviceBuffer(nvox)
            dev.synth_occ(0, W, H, D, 0, d_lab)                                  # 0 / 1 bytes: labels 0 and 1
            L, lib = pb3d._lib, pb3d._lib.load()
            ang = (C.c_int * 6)(*([90] * 6)); skip = (C.c_int * 6)(*([0] * 6))
            fn = lambda: L.check(lib.pb3d_part_carve_label_dev(L.ctx(), C.c_void_p(d_lab.ptr), W, H, D, C.c_void_p(d_ms.ptr), C.c_void_p(d_ms.ptr), ang, skip, 6,
                                                               C.c_void_p(d_lout.ptr)))
        if a.variants:
            sets = [dict(kv.split("=") for kv in v.split(",") if kv) for v in a.variants.split(";")]
            keys = sorted({k for st in sets for k in st})
            for r in range(a.rounds):
                for v, st in zip(a.variants.split(";"), sets):
                    for k in keys:
                        pb3d._lib.set_tuning(k, int(st.get(k, 0)))
                    res.setdefault(v or "default", []).append(round(timeit(fn, a.reps), 4))
            for k in keys:
                pb3d._lib.set_tuning(k, 0)
            print(json.dumps({"shape": [W, H, D], "ms_by_variant": res}), flush=True)
            for b in (d_mwh, d_occ, d_o, d_t):
                b.free()
            continue
        for r in range(a.rounds):
            for wd in a.wide.split(","):
                pb3d._lib.set_tuning("rot90_wide", int(wd))
                for f in a.fills.split(","):
                    pb3d._lib.set_tuning("rot90_fill", int(f))
                    res.setdefault(f"tile256:{f}" if wd == "0" else f"tile128:{f}", []).append(round(timeit(lambda: dev.process_grid(d_occ, W, H, D, d_mwh, 90, d_o, d_t), a.reps), 4))
        pb3d._lib.set_tuning("rot90_fill", 0); pb3d._lib.set_tuning("rot90_wide", 0)
        print(json.dumps({"shape": [W, H, D], "ms_by_fill": res}), flush=True)
        for b in (d_mwh, d_occ, d_o, d_t):
            b.free()


if __name__ == "__main__":
    main()
