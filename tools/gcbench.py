"""global_carve with angle steps other than 90 at 1024^3 (or --size): the bit-sliced chain from the mask to the colours against the byte
chain with fused first / last steps (tune sliced = 1), interleaved on one box.  One JSON line per angle step."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd"))
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


ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--intervals", default="45,30,10,5")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
S = a.size
d_bhw = dev.DeviceBuffer(S * S); d_rgb = dev.DeviceBuffer(S * S * 3); d_mwh = dev.DeviceBuffer(S * S)
dev.synth_mask16(S, d_binary_hw=d_bhw, d_rgb_hw3=d_rgb, d_binary_wh=d_mwh)
d_col = dev.DeviceBuffer(S ** 3 * 3)
for ai in (int(v) for v in a.intervals.split(",")):
    # sliced: the last 90-degree step writes the colours itself; sliced_unfused_last: table step + separate colour un-slicing (round 3's form)
    modes = {"sliced": (0, 0), "sliced_unfused_last": (0, 1), "bytes": (1, 0)}
    res = {m: [] for m in modes}
    for r in range(a.rounds):
        for m, (sl, fl) in modes.items():
            pb3d._lib.set_tuning("sliced", sl); pb3d._lib.set_tuning("s32_fuse_last", fl)
            res[m].append(round(timeit(lambda: dev.global_carve(d_bhw, d_rgb, S, S, ai, d_col), 3), 4))
    pb3d._lib.set_tuning("sliced", 0); pb3d._lib.set_tuning("s32_fuse_last", 0)
    print(json.dumps({"size": S, "interval": ai, "rotation_steps": 90 // ai, "ms_sliced": res["sliced"], "ms_sliced_unfused_last": res["sliced_unfused_last"],
                      "ms_bytes": res["bytes"]}), flush=True)
