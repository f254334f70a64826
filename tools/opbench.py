"""Device-resident timing of every op of the path (M1..M8 of SURVEY.md 8(d)) on synthetic 1024^3 inputs.
Development/measurement tool: python tools/opbench.py [--size 1024] [--ops M1,M3,...]; one JSON line per op."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd"))
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402

import pb3d  # noqa: E402
from pb3d import device as dev  # noqa: E402

PEAK = 8000.0


def timeit(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    dev.sync()
    e0, e1 = dev.Event(), dev.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    return e1.elapsed_ms_since(e0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--ops", default="M1,M2,M3,M4,M5,M6,M7,M8")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    S = a.size
    nvox = S ** 3
    ops = a.ops.split(",")
    lib, L = pb3d._lib.load(), pb3d._lib
    d_mwh = dev.DeviceBuffer(S * S); d_bhw = dev.DeviceBuffer(S * S); d_rgb = dev.DeviceBuffer(S * S * 3)
    dev.synth_mask16(S, d_binary_hw=d_bhw, d_rgb_hw3=d_rgb, d_binary_wh=d_mwh)
    res = []

    def report(op, name, ms, bpv, extra=None):
        r = {"op": op, "name": name, "size": S, "ms": round(ms, 4), "Mvoxel_s": round(nvox / ms / 1e3, 1),
             "alg_B_per_voxel": bpv, "alg_GB_s": round(bpv * nvox / ms / 1e6, 1), "frac_of_8TBs": round(bpv * nvox / ms / 1e6 / PEAK, 4)}
        if extra:
            r.update(extra)
        print(json.dumps(r), flush=True)
        res.append(r)

    if "M1" in ops:
        d_in = dev.DeviceBuffer(nvox * 3); d_out = dev.DeviceBuffer(nvox * 3)
        dev.synth_sem(0, S, S, S, 1, d_in)
        report("M1", "carve_voxel_grid_with_masks(sem,binary)", timeit(lambda: dev.carve_mask(d_in, S, S, S, 3, d_mwh, d_out), 20), 6)
        d_in.free(); d_out.free()
    d_occ = dev.DeviceBuffer(nvox); d_o1 = dev.DeviceBuffer(nvox); d_tmp = dev.DeviceBuffer(nvox)
    dev.synth_occ(0, S, S, S, 0, d_occ)
    if "M2" in ops:
        report("M2", "carve_voxel_grid_with_masks(occ,binary)", timeit(lambda: dev.carve_mask(d_occ, S, S, S, 1, d_mwh, d_o1), 20), 2)
    if "M3" in ops:
        report("M3", "process_voxel_grid(occ,binary,90)", timeit(lambda: dev.process_grid(d_occ, S, S, S, d_mwh, 90, d_o1, d_tmp), a.reps), 2)
    if "M4" in ops:
        M = np.empty(9); off = np.empty(3)
        for ang in (45, 5):
            L.check(lib.pb3d_rotinv(ang, L.p_dbl(M))); L.check(lib.pb3d_offset(L.p_dbl(M), (C.c_int64 * 3)(S, S, S), L.p_dbl(off)))
            report("M4", f"one rotate+carve step, {ang} deg", timeit(lambda: dev.rotate_carve(d_occ, S, S, S, M, off, d_mwh, d_o1), a.reps), 2)
    if "M5" in ops:
        d_out = dev.DeviceBuffer(nvox * 3)
        report("M5", "global_carve(binary,rgb,90)", timeit(lambda: dev.global_carve(d_bhw, d_rgb, S, S, 90, d_out), a.reps), 3)
        d_out.free()
    d_occ.free(); d_o1.free(); d_tmp.free()
    return res


if __name__ == "__main__":
    main()
