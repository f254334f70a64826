"""Chains of DIFFERENT generic-angle steps (process_voxel_grid with a small angle interval: no table is ever reused): what the
asynchronous table prefetch buys.  tune misc4 = 0: prefetch on the auxiliary stream; 2: tables built in line.
python tools/chainbench.py [--shapes 512x278x512,512x512x512,1024x1024x1024] [--interval 10]"""
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
    ap.add_argument("--shapes", default="512x278x512,512x512x512,1024x1024x1024")
    ap.add_argument("--interval", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    for sh in a.shapes.split(","):
        W, H, D = (int(v) for v in sh.split("x"))
        nvox = W * H * D
        d_mwh = dev.from_numpy((rng.random((W, H)) < 0.8).astype(np.uint8))
        d_occ = dev.DeviceBuffer(nvox); d_o = dev.DeviceBuffer(nvox); d_t = dev.DeviceBuffer(nvox)
        dev.synth_occ(0, W, H, D, 0, d_occ)
        res = {"0": [], "2": []}
        outs = {}
        for r in range(a.rounds):
            for mode in ("0", "2"):
                pb3d._lib.set_tuning("misc4", int(mode))
                res[mode].append(round(timeit(lambda: dev.process_grid(d_occ, W, H, D, d_mwh, a.interval, d_o, d_t), 3), 4))
                if r == 0:
                    outs[mode] = d_o.download((W, H, D)) if nvox <= 1 << 28 else None
        pb3d._lib.set_tuning("misc4", 0)
        same = None if outs["0"] is None else bool(np.array_equal(outs["0"], outs["2"]))
        nsteps = 90 // a.interval
        print(json.dumps({"shape": [W, H, D], "interval": a.interval, "rotation_steps": nsteps, "ms_prefetch": res["0"], "ms_inline": res["2"],
                          "per_step_us_prefetch": round(1e3 * min(res["0"]) / nsteps, 1), "per_step_us_inline": round(1e3 * min(res["2"]) / nsteps, 1),
                          "results_equal": same}), flush=True)
        for b in (d_mwh, d_occ, d_o, d_t):
            b.free()


if __name__ == "__main__":
    main()
