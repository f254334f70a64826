"""A/B timing of the generic-angle rotate+carve step (M4) across the tile kernels and their knobs, with a full-volume equality
check between kernels.  python tools/m4bench.py [--size 1024] [--angles 45,5] [--variants 128,256:8,256:16,256:32]
A variant is  tile[:rot8_ty[:misc0[:misc1 ...]]] ; one JSON line per (angle, variant)."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd"))
import numpy as np  # noqa: E402

import pb3d  # noqa: E402
from pb3d import device as dev  # noqa: E402


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
    ap.add_argument("--size", default="1024")
    ap.add_argument("--angles", default="45,5")
    ap.add_argument("--variants", default="128,256:8,256:16,256:32,256:64")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    dims = [int(v) for v in a.size.split("x")]
    W, H, D = dims if len(dims) == 3 else (dims[0],) * 3
    nvox = W * H * D
    lib, L = pb3d._lib.load(), pb3d._lib
    rng = np.random.default_rng(3)
    d_occ = dev.DeviceBuffer(nvox); d_o = dev.DeviceBuffer(nvox)
    if W == H == D:
        d_mwh = dev.DeviceBuffer(W * H)
        dev.synth_mask16(W, d_binary_wh=d_mwh)
    else:
        d_mwh = dev.from_numpy((rng.random((W, H)) < 0.8).astype(np.uint8))
    dev.synth_occ(0, W, H, D, 0, d_occ)
    M = np.empty(9); off = np.empty(3)
    for ang in [int(v) for v in a.angles.split(",")]:
        L.check(lib.pb3d_rotinv(ang, L.p_dbl(M))); L.check(lib.pb3d_offset(L.p_dbl(M), (C.c_int64 * 3)(W, H, D), L.p_dbl(off)))
        ref = None
        for var in a.variants.split(","):
            f = [int(v) for v in var.split(":")]
            L.set_tuning("rotate_tile", f[0])
            L.set_tuning("rot8_ty", f[1] if len(f) > 1 else 0)
            for k in range(5):                  # misc5 is left to the environment (PB3D_TUNE5: one-off diagnostics)
                L.set_tuning(f"misc{k}", f[2 + k] if len(f) > 2 + k else 0)
            ms = timeit(lambda: dev.rotate_carve(d_occ, W, H, D, M, off, d_mwh, d_o), a.reps)
            same = None
            if not a.no_check:
                got = d_o.download((W, H, D))
                if ref is None:
                    ref = got
                else:
                    same = bool(np.array_equal(ref, got))
            print(json.dumps({"op": "M4", "shape": [W, H, D], "angle": ang, "variant": var, "ms": round(ms, 4),
                              "alg_GB_s": round(2 * nvox / ms / 1e6, 1), "frac_of_8TBs": round(2 * nvox / ms / 1e6 / 8000, 4),
                              "equals_first_variant": same}), flush=True)
    L.set_tuning("rotate_tile", 0)


if __name__ == "__main__":
    main()
