"""process_voxel_grid chains: the bit-sliced chain (csrc/sliced.hip) against the byte chain (tune sliced = 1), device-resident.
python tools/slicedbench.py [--shapes 512x278x512,1024x1024x1024] [--intervals 5,45] [--rounds 3]
One JSON line per (shape, interval): ms per call (min of rounds), per rotation step, results equal."""
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
    ap.add_argument("--shapes", default="512x278x512,355x512x355,512x512x512,1024x1024x1024")
    ap.add_argument("--intervals", default="5,45")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    for sh in a.shapes.split(","):
        W, H, D = (int(v) for v in sh.split("x"))
        nvox = W * H * D
        d_mwh = dev.from_numpy((rng.random((W, H)) < 0.8).astype(np.uint8))
        d_occ = dev.DeviceBuffer(nvox); d_o = dev.DeviceBuffer(nvox); d_t = dev.DeviceBuffer(nvox)
        dev.synth_occ(0, W, H, D, 0, d_occ)
        for ai in (int(v) for v in a.intervals.split(",")):
            res = {0: [], 1: []}
            outs = {}
            for r in range(a.rounds):
                for mode in (0, 1):
                    pb3d._lib.set_tuning("sliced", mode)
                    res[mode].append(round(timeit(lambda: dev.process_grid(d_occ, W, H, D, d_mwh, ai, d_o, d_t), a.reps), 4))
                    if r == 0:
                        outs[mode] = d_o.download((W, H, D)) if nvox <= 1 << 28 else None
            pb3d._lib.set_tuning("sliced", 0)
            same = None if outs[0] is None else bool(np.array_equal(outs[0], outs[1]))
            nrot = 90 // ai
            print(json.dumps({"shape": [W, H, D], "interval": ai, "rotation_steps": nrot, "ms_sliced": res[0], "ms_bytes": res[1],
                              "per_step_us_sliced": round(1e3 * min(res[0]) / nrot, 1), "per_step_us_bytes": round(1e3 * min(res[1]) / nrot, 1),
                              "results_equal": same}), flush=True)
        for b in (d_mwh, d_occ, d_o, d_t):
            b.free()


if __name__ == "__main__":
    main()
