import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, pb3d
from pb3d import device as dev
S = 1024; nvox = S ** 3
lib, L = pb3d._lib.load(), pb3d._lib
d_mwh = dev.DeviceBuffer(S * S); dev.synth_mask16(S, d_binary_wh=d_mwh)
d_occ = dev.DeviceBuffer(nvox); d_o1 = dev.DeviceBuffer(nvox)
dev.synth_occ(0, S, S, S, 0, d_occ)
L.check(lib.pb3d_dev_memset(L.ctx(), C.c_void_p(d_o1.ptr), 7, nvox))
M = np.empty(9); off = np.empty(3)
L.check(lib.pb3d_rotinv(45, L.p_dbl(M))); L.check(lib.pb3d_offset(L.p_dbl(M), (C.c_int64 * 3)(S, S, S), L.p_dbl(off)))
e0, e1 = dev.Event(), dev.Event()
for _ in range(3):
    e0.record(); dev.rotate_carve(d_occ, S, S, S, M, off, d_mwh, d_o1); e1.record(); dev.sync()
    print("ms", e1.elapsed_ms_since(e0))
h = d_o1.download((S, 64 * 1024))[:4].astype(np.uint64).sum()
print("DBG", os.environ.get("PB3D_DBG"), "sum of first 256K outputs", int(h))
