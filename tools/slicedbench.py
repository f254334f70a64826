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
    ap.add_argument("--tune", default="", help="development knobs, e.g. s32_order=1")
    a = ap.parse_args()
    for kv in [t for t in a.tune.split(',') if t]:
        pb3d._lib.set_tuning(kv.split('=')[0], int(kv.split('=')[1]))
    rng = np.random.default_rng(5)
    for sh in a.shapes.split(","):
        W, H, D = (int(v) for v in sh.split("x"))
        nvox = W * H * D
        d_mwh = dev.from_numpy((rng.random((W, H)) < 0.8).astype(np.uint8))
        d_occ = dev.DeviceBuffer(nvox); d_o = dev.DeviceBuffer(nvox); d_t = dev.DeviceBuffer(nvox)
        dev.synth_occ(0, W, H, D, 0, d_occ)
        for ai in (int(v) for v in a.intervals.split(",")):
            # modes: the bit-sliced chain (its last 90-degree step un-slices in its own stores), the same with the last step as a table
            # step + a separate un-slicing pass (round 3's form), the byte chain
            modes = {"sliced": (0, 0), "sliced_unfused_last": (0, 1), "bytes": (1, 0)}
            res = {m: [] for m in modes}
            outs = {}
            for r in range(a.rounds):
                for m, (sl, fl) in modes.items():
                    pb3d._lib.set_tuning("sliced", sl); pb3d._lib.set_tuning("s32_fuse_last", fl)
                    res[m].append(round(timeit(lambda: dev.process_grid(d_occ, W, H, D, d_mwh, ai, d_o, d_t), a.reps), 4))
                    if r == 0:
                        outs[m] = d_o.download((W, H, D))
            pb3d._lib.set_tuning("sliced", 0); pb3d._lib.set_tuning("s32_fuse_last", 0)
            same = bool(np.array_equal(outs["sliced"], outs["bytes"]) and np.array_equal(outs["sliced_unfused_last"], outs["bytes"]))
            outs = None
            nrot = 90 // ai
            print(json.dumps({"shape": [W, H, D], "interval": ai, "rotation_steps": nrot, "ms_sliced": res["sliced"], "ms_sliced_unfused_last": res["sliced_unfused_last"],
                              "ms_bytes": res["bytes"], "per_step_us_sliced": round(1e3 * min(res["sliced"]) / nrot, 1),
                              "per_step_us_bytes": round(1e3 * min(res["bytes"]) / nrot, 1), "results_equal": same}), flush=True)
        for b in (d_mwh, d_occ, d_o, d_t):
            b.free()


if __name__ == "__main__":
    main()
