"""Device-resident timing of every op of the path (M1..M8 of SURVEY.md 8(d), plus the N1/N2 kernels of the notebook-1 chain) on
synthetic inputs.  Development/measurement tool: python tools/opbench.py [--size 1024] [--shape WxHxD] [--ops M1,M3,...,N2]; one JSON
line per op.  Pricing: `alg_B_per_voxel` is SURVEY 8(d)'s algorithmic figure for the sweeps that are EXECUTED (a folded 0-degree step
moves nothing and is not priced); ops whose intermediates are not bytes (the bit-sliced chains) also carry `moved_B_per_voxel`, the
bytes that cross the HBM by design, and their `frac_of_8TBs` is computed from THAT (never from bytes that are not moved)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
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
    ap.add_argument("--ops", default="M1,M2,M3,M4,M5,M6,M7,M8,A2,A6,A9,N2")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--tune", default="", help="development knobs, e.g. rot90_fill=6 (pb3d_set_tuning)")
    a = ap.parse_args()
    for kv in [t for t in a.tune.split(',') if t]:
        pb3d._lib.set_tuning(kv.split('=')[0], int(kv.split('=')[1]))
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
        for parts in (2, 4, 8):      # what one rank of an N-GPU run does (X-slab of S/N planes); ideal = ms(M1) / N
            ms = timeit(lambda: dev.carve_mask(d_in, S // parts, S, S, 3, d_mwh, d_out), 50)
            print(json.dumps({"op": "M1/slab", "planes": S // parts, "ms": round(ms, 4), "alg_GB_s": round(6 * nvox / parts / ms / 1e6, 1)}), flush=True)
        d_in.free(); d_out.free()
    if "A2" in ops or "A6" in ops:      # the two elementwise helpers of the path (not in SURVEY's M list): 4 B/voxel each
        d_c = dev.DeviceBuffer(nvox * 3); d_o = dev.DeviceBuffer(nvox); d_o2 = dev.DeviceBuffer(nvox)
        dev.synth_sem(0, S, S, S, 1, d_c)
        report("A2", "_occupancy(rgb grid)", timeit(lambda: dev.occupancy(d_c, nvox, d_o), 10), 4)
        dev.carve_mask(d_o, S, S, S, 1, d_mwh, d_o2)
        report("A6", "apply_colored_mask_to_voxel_grid", timeit(lambda: dev.color_apply(d_o2, S, S, S, d_rgb, d_c), 10), 4)
        d_c.free(); d_o.free(); d_o2.free()
    d_occ = dev.DeviceBuffer(nvox); d_o1 = dev.DeviceBuffer(nvox); d_tmp = dev.DeviceBuffer(nvox)
    dev.synth_occ(0, S, S, S, 0, d_occ)
    if "M2" in ops:
        report("M2", "carve_voxel_grid_with_masks(occ,binary)", timeit(lambda: dev.carve_mask(d_occ, S, S, S, 1, d_mwh, d_o1), 20), 2)
    if "M3" in ops:
        report("M3", "process_voxel_grid(occ,binary,90)", timeit(lambda: dev.process_grid(d_occ, S, S, S, d_mwh, 90, d_o1, d_tmp), a.reps), 2)
    if "M3" in ops:
        for ai in (45, 60, 30, 5):     # chained: len(range(0, 91, ai)) steps; the 0-degree carve is folded (it moves nothing): 90 // ai sweeps
            nrot = 90 // ai
            ms = timeit(lambda: dev.process_grid(d_occ, S, S, S, d_mwh, ai, d_o1, d_tmp), a.reps)
            # two and more rotation steps run bit-sliced (csrc/sliced.hip): 1 B in, 1/8 out; nrot x (1/8 + 1/8); 1/8 in, 1 B out
            moved = 2.0 * nrot if nrot < 2 else 2.25 + 0.25 * nrot
            r = {"op": "M3+", "name": f"process_voxel_grid(occ,binary,{ai}): {nrot} rotation sweeps (0 deg folded)", "size": S, "ms": round(ms, 4),
                 "ms_per_sweep": round(ms / nrot, 4), "alg_B_per_voxel": 2 * nrot, "alg_GB_s": round(2 * nrot * nvox / ms / 1e6, 1),
                 "moved_B_per_voxel": moved, "moved_GB_s": round(moved * nvox / ms / 1e6, 1), "frac_of_8TBs": round(moved * nvox / ms / 1e6 / PEAK, 4)}
            print(json.dumps(r), flush=True)
    if "M4" in ops:
        M = np.empty(9); off = np.empty(3)
        for ang in (45, 5):
            L.check(lib.pb3d_rotinv(ang, L.p_dbl(M))); L.check(lib.pb3d_offset(L.p_dbl(M), (C.c_int64 * 3)(S, S, S), L.p_dbl(off)))
            report("M4", f"one rotate+carve step, {ang} deg", timeit(lambda: dev.rotate_carve(d_occ, S, S, S, M, off, d_mwh, d_o1), a.reps), 2)
    d_occ.free(); d_o1.free(); d_tmp.free()
    d_col = None
    if any(o in ops for o in ("M5", "M6", "M7", "M8", "A9", "N2")):
        d_col = dev.DeviceBuffer(nvox * 3)
        ms = timeit(lambda: dev.global_carve(d_bhw, d_rgb, S, S, 90, d_col), a.reps)
        if "M5" in ops:
            report("M5", "global_carve(binary,rgb,90)", ms, 3)
            for ai in (45, 60):
                report("M5+", f"global_carve(binary,rgb,{ai})", timeit(lambda: dev.global_carve(d_bhw, d_rgb, S, S, ai, d_col), a.reps), 3)
            dev.global_carve(d_bhw, d_rgb, S, S, 90, d_col)
    if "M6" in ops:
        # six 90-degree part jobs of notebook 1 on the structured grid (labels 1..9 of the synthetic mask are the part colours)
        import synth_host
        lab, binary, rgb = synth_host.mask16(S)
        names = ["full_building", "chhatris", "plinth", "front_minarets", "small_minarets", "dome"]
        msub = np.zeros((len(names), S, S), np.uint8)
        for j, nm in enumerate(names):
            msub[j] = np.all(rgb == np.array(pb3d.PART_COLORS[nm], np.uint8), axis=-1).T
        mcarve = np.ascontiguousarray(msub.transpose(0, 2, 1))      # W == H: _mask_to_wh transposes again
        d_ms = dev.from_numpy(msub); d_mc = dev.from_numpy(mcarve); d_out = dev.DeviceBuffer(nvox * 3)
        ang = (C.c_int * 6)(*([90] * 6)); skip = (C.c_int * 6)(*[0 if msub[j].any() else 1 for j in range(6)])
        fn = lambda: L.check(lib.pb3d_part_carve_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, C.c_void_p(d_ms.ptr), C.c_void_p(d_mc.ptr),
                                                      ang, skip, 6, C.c_void_p(d_out.ptr)))
        ms = timeit(fn, max(2, a.reps // 2), warm=1)
        # the six jobs run as ONE fused sweep (k_part90): it owes a read of the colour grid as occupancy source (3 B), a read of
        # the kept rows as output source (<= 3 B) and the write (3 B) -- <= 9 B/voxel whatever the number of jobs.  (SURVEY's
        # 6 B/voxel PER JOB prices the reference's job-by-job execution, which this sweep does not perform.)
        # priced at what the sweep really moves (PMC, profiles/r02_opbench_pmc_traffic.json: 3.26 GB read + 3.22 GB written = 6.0 B/voxel;
        # an upper bound of 9 B/voxel flattered the figure in round 2)
        report("M6", "part_carve, six 90-degree jobs (one fused sweep)", ms, 6, {"jobs": 6, "ms_per_job": round(ms / 6, 4),
                                                                                 "reference_execution_B_per_voxel": 36})
        for b in (d_ms, d_mc, d_out):
            b.free()
    if "A9" in ops:
        # connected components of one part colour on the carved 1024^3 colour grid (left_right_guided_carve / recolor set-up):
        # label + per-component statistics; labels are int32 (4 B/voxel written), the grid is read once (3 B/voxel)
        col = np.array(pb3d.PART_COLORS["full_building"], np.uint8)
        d_lab = dev.DeviceBuffer(nvox * 4)
        ncomp = C.c_int64(0)
        lab_fn = lambda: L.check(lib.pb3d_label_color_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, L.p_u8(col), C.c_void_p(d_lab.ptr), C.byref(ncomp)))
        ms = timeit(lab_fn, 2, warm=1)
        report("A9", "connected components of one colour (label)", ms, 7, {"components": ncomp.value})
        nc = max(1, ncomp.value)
        bbox = (C.c_int64 * (6 * nc))(); cnt = (C.c_int64 * nc)(); csum = (C.c_int64 * (3 * nc))()
        st_fn = lambda: L.check(lib.pb3d_component_stats_dev(L.ctx(), C.c_void_p(d_lab.ptr), S, S, S, ncomp.value, bbox, cnt, csum))
        report("A9", "component statistics (bbox, count, coordinate sums)", timeit(st_fn, 2, warm=1), 4)
        d_lab.free()
    if "N2" in ops:
        # the N1 / N2 kernels of the notebook-1 chain on the carved colour grid: orientation (transpose + flip: 3 B read, 3 B written),
        # the four in-place extrusions (a column scan reads at most the 3 B/voxel it crosses and writes `depth` cells), labelling WITH the
        # component statistics (3 B read + 4 B of int32 labels written), recolouring (4 B of labels read, flagged voxels written)
        d_o = dev.DeviceBuffer(nvox * 3)
        report("N2", "orient (transpose(2,1,0,3) + flip)", timeit(lambda: L.check(lib.pb3d_orient_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, C.c_void_p(d_o.ptr))), a.reps), 6)
        d_v = dev.DeviceBuffer(S * S)
        L.check(lib.pb3d_dev_memset(L.ctx(), C.c_void_p(d_v.ptr), 1, S * S))
        fc = np.array(pb3d.PART_COLORS["windows"], np.uint8)
        for axis, nm in ((2, "z"), (0, "x")):
            for plus in (1, 0):
                fn = lambda: L.check(lib.pb3d_extrude_dev(L.ctx(), C.c_void_p(d_o.ptr), S, S, S, C.c_void_p(d_v.ptr), S, axis, plus, 10, L.p_u8(fc), C.c_void_p(d_o.ptr)))
                # the bytes an extrusion needs depend on the data (a column is scanned up to its first occupied voxel, then `depth` cells are
                # written): no algorithmic-byte figure, the time only
                ms = timeit(fn, a.reps)
                print(json.dumps({"op": "N2", "name": f"extrude_from_surface axis {axis} {'+' if plus else '-'} depth 10, in place (k_extrude_{nm})", "size": S,
                                  "ms": round(ms, 4), "alg_B_per_voxel": None, "note": "data-dependent traffic: scan to the first occupied voxel of every column under the mask"}), flush=True)
        from pb3d.voxel_carving_utils import _label_stats
        col = np.array(pb3d.PART_COLORS["full_building"], np.uint8)
        d_lab = dev.DeviceBuffer(nvox * 4)
        out = {}
        def lab_fn():
            out["n"] = _label_stats(d_col, (S, S, S), col, d_lab)[0]
        report("N2", "connected components of one colour + statistics (one pass, one round trip)", timeit(lab_fn, 2, warm=1), 7, {"components": out["n"]})
        def lab_fn2():
            out["n"] = _label_stats(d_col, (S, S, S), col, d_lab, members_only=True)[0]
        members = int(np.count_nonzero(np.all(dev.DeviceGrid(d_col, (S, S, S, 3)).numpy() == col, axis=-1))) if S <= 512 else None
        report("N2", "... labels written for the colour's voxels only (members_only: what the notebook-1 chain uses)", timeit(lab_fn2, 2, warm=1), 3,
               {"components": out["n"], "member_voxels": members, "note": "3 B/voxel read + 4 B per MEMBER voxel written (not priced)"})
        lab_fn()
        flags = np.ones(max(1, out["n"]), np.uint8)
        fn = lambda: L.check(lib.pb3d_recolor_components_dev(L.ctx(), C.c_void_p(d_lab.ptr), nvox, L.p_u8(flags), max(1, out["n"]), L.p_u8(fc), C.c_void_p(d_o.ptr)))
        report("N2", "recolor_backward_components: recolour pass over the label volume (k_recolor_flagged)", timeit(fn, 2, warm=1), 4)
        fn2 = lambda: L.check(lib.pb3d_recolor_last_labelled_dev(L.ctx(), C.c_void_p(d_lab.ptr), nvox, L.p_u8(flags), max(1, out["n"]), L.p_u8(fc), C.c_void_p(d_o.ptr), 3))
        ms = timeit(fn2, 2, warm=1)
        print(json.dumps({"op": "N2", "name": "recolor_backward_components: recolour pass over the labelling's membership bits (k_recolor_bits)", "size": S, "ms": round(ms, 4),
                          "alg_B_per_voxel": 0.125, "note": "1 bit/voxel read + 4 B label and 3 B colour per member voxel"}), flush=True)
        for b in (d_o, d_v, d_lab):
            b.free()
    if "M7" in ops or "M8" in ops:
        cols = np.ascontiguousarray(np.array(list(pb3d.PART_COLORS.values()), np.uint8))
        n = C.c_int64(0)
        cnt = lambda: L.check(lib.pb3d_points_count_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, 3, L.p_u8(cols), len(cols), 1, C.byref(n)))
        cnt()
        npts = n.value
        d_pts = dev.DeviceBuffer(max(1, npts) * 12); d_pc = dev.DeviceBuffer(max(1, npts) * 3)
        fill = lambda: L.check(lib.pb3d_points_fill_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, 3, L.p_u8(cols), len(cols), 1, npts,
                                                        C.c_void_p(d_pts.ptr), C.c_void_p(d_pc.ptr)))
        def both():
            cnt(); fill()
        ms = timeit(both, max(2, a.reps // 2), warm=1)
        fillfrac = npts / nvox
        if "M7" in ops:
            ms_fill = timeit(fill, max(3, a.reps), warm=1)          # the fill pass alone (the scanned offsets of the last count stay in the context)
            ms_cnt = timeit(cnt, max(3, a.reps), warm=1)            # the count pass alone, with its host round trip
            report("M7", "get_voxel_points_by_parts (10 parts): count + fill", ms, round(3 + 15 * fillfrac, 3),
                   {"points": npts, "fill": round(fillfrac, 4), "count_pass_ms": round(ms_cnt, 4), "fill_pass_ms": round(ms_fill, 4)})
            n1 = C.c_int64(0)
            one = lambda: L.check(lib.pb3d_points_extract_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, 3, L.p_u8(cols), len(cols), npts,
                                                              C.c_void_p(d_pts.ptr), C.c_void_p(d_pc.ptr), C.byref(n1)))
            ms1 = timeit(one, max(2, a.reps // 2), warm=1)
            assert n1.value == npts
            report("M7", "get_voxel_points_by_parts (10 parts): ONE pass (pb3d_points_extract_dev, decoupled look-back)", ms1,
                   round(3 + 15 * fillfrac, 3), {"points": npts, "fill": round(fillfrac, 4)})
        if "M8" in ops and npts:
            from pb3d.camera_geometry import look_at_rotation
            cam = np.array([S / 2, S / 2, -2.5 * S], np.float32); tgt = np.array([S / 2, S / 2, S / 2], np.float32)
            R = np.ascontiguousarray(look_at_rotation(cam, tgt), np.float64); cd = np.ascontiguousarray(cam, np.float64)
            Hi = Wi = S
            d_img = dev.DeviceBuffer(Hi * Wi * 3)
            prec = (C.c_int * 4)(0, 0, 0, 0)
            fn = lambda: L.check(lib.pb3d_project_dev(L.ctx(), C.c_void_p(d_pts.ptr), 0, C.c_void_p(d_pc.ptr), npts, L.p_dbl(R), L.p_dbl(cd),
                                                      float(1.2 * S), S / 2.0, S / 2.0, prec, Hi, Wi, C.c_void_p(d_img.ptr)))
            ms = timeit(fn, max(2, a.reps // 2), warm=1)
            # 12 B of coordinates per point are read by the point kernel; colours are only read for the <= H*W winners
            r = {"op": "M8", "name": "project_colored_voxels (f32 camera)", "size": S, "ms": round(ms, 4), "points": npts,
                 "Mpts_s": round(npts / ms / 1e3, 1), "alg_B_per_point": 12, "alg_GB_s": round(12 * npts / ms / 1e6, 1),
                 "frac_of_8TBs": round(12 * npts / ms / 1e6 / PEAK, 4)}
            print(json.dumps(r), flush=True)
            d_img.free()
        d_pts.free(); d_pc.free()
    if d_col is not None:
        d_col.free()
    return res


if __name__ == "__main__":
    main()
