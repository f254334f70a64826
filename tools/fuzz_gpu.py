"""Long randomised GPU-vs-oracle sweep (development tool, not part of the test suite): python tools/fuzz_gpu.py [seconds] [seed].
Covers carve, colour apply, occupancy, process_voxel_grid (both generic-angle tile kernels pinned in turn), part_carve,
global_carve, point extraction and projection on random shapes (odd / aligned / ragged), densities and dtypes."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "part-based-3d-reconstruction_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import pb3d  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    oracle.set_threads(min(16, len(os.sched_getaffinity(0))))
    pal = np.array(list(oracle.PART_COLORS.values()), np.uint8)
    names = list(oracle.PART_COLORS)
    t0 = time.time(); n = 0; counts = {}; last = t0
    def ok(tag, a, b, info):
        counts[tag] = counts.get(tag, 0) + 1
        if not np.array_equal(a, b):
            print("MISMATCH", tag, info, flush=True); sys.exit(1)
    while time.time() - t0 < budget:
        n += 1
        if time.time() - last > 60:
            last = time.time(); print(f"... {n} cases, {int(last - t0)} s", flush=True)
        big = n % 7 == 0 or os.environ.get("FUZZ_BIG") == "1"      # FUZZ_BIG=1: every case large enough for the tiled kernels
        lo = 120 if os.environ.get("FUZZ_BIG") == "1" else 1
        W = int(rng.integers(lo, 300 if big else 100)); H = int(rng.integers(max(1, lo // 3), 70 if big else 40)); D = int(rng.integers(lo, 300 if big else 100))
        r = n % 5
        if r == 0: D = W
        if r == 1: W = D = int(rng.choice([16, 32, 48, 64, 96, 128, 160, 256]))
        if r == 2: D = W + 2 * int(rng.integers(-10, 10)); D = max(1, D)
        if n % 11 == 3:      # rows that are not whole lines on H * D % 128 == 0 grids: the flat 90-degree kernels (k_rot90_flat, k_global_carve90f)
            H = int(rng.choice([16, 32, 64, 128])); W = D = int(rng.integers(129, 300)); D += 2 * int(rng.integers(0, 3))
        ai = int(rng.choice([90, 90, 60, 45, 30, 20, 10, 120, 7]))
        dens = rng.uniform(0.05, 0.95)
        g = (rng.random((W, H, D)) < dens).astype(np.uint8) if n % 6 else rng.integers(0, 256, (W, H, D), dtype=np.uint8)
        m = rng.random((H, W)) < rng.uniform(0.2, 1.0)
        info = (n, W, H, D, ai)
        want = oracle.process_voxel_grid(g, m, ai)
        for sliced in (0, 1):        # the bit-sliced path and the byte chain (arithmetic kernel + permutation kernels)
            pb3d._lib.set_tuning("sliced", sliced)
            ok("process", pb3d.process_voxel_grid(g, m, ai), want, info + (sliced,))
        pb3d._lib.set_tuning("sliced", 0)
        if n % 4 == 1 and W * H * D <= 300000:       # the same loop on a grid of another dtype (csrc/rotate_typed.hip), bytes and dtype compared
            dt = str(rng.choice(["bool", "int8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64", "complex64", "complex128"]))
            if dt == "bool": gt = rng.random((W, H, D)) < dens
            elif dt.startswith("complex"): gt = ((rng.random((W, H, D)) * 400 - 200) + 1j * (rng.random((W, H, D)) * 10 - 5)).astype(dt)
            elif dt.startswith("float"): gt = (rng.random((W, H, D)) * 400 - 200).astype(dt)
            elif dt.startswith("u"): gt = (rng.random((W, H, D)) * min(float(np.iinfo(dt).max), 2.0 ** 45)).astype(dt)
            else: gt = ((rng.random((W, H, D)) - 0.5) * min(float(np.iinfo(dt).max), 2.0 ** 45) * 2).astype(dt)
            gt[rng.random((W, H, D)) > dens] = 0
            wt = oracle.process_voxel_grid_typed(gt, m, ai); ot = pb3d.process_voxel_grid(gt, m, ai)
            ok("process_typed", ot.view(np.uint8), wt.view(np.uint8), info + (dt,))
            if ot.dtype != wt.dtype:
                print("MISMATCH process_typed dtype", info, dt, ot.dtype, wt.dtype, flush=True); sys.exit(1)
        col = pal[rng.integers(0, len(pal), (W, H, D))] * (rng.random((W, H, D, 1)) < dens).astype(np.uint8)
        ok("carve_rgb", pb3d.carve_voxel_grid_with_masks(col, m), oracle.carve_voxel_grid_with_masks(col, m), info)
        ok("carve_occ", pb3d.carve_voxel_grid_with_masks(g, m), oracle.carve_voxel_grid_with_masks(g, m), info)
        rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        c3 = rng.integers(0, 3, (W, H, D), dtype=np.uint8)
        ok("color_apply", pb3d.apply_colored_mask_to_voxel_grid(c3, rgb), oracle.apply_colored_mask_to_voxel_grid(c3, rgb), info)
        sem = pal[rng.integers(0, len(pal), (H // 3 + 1, W // 3 + 1))].repeat(3, 0).repeat(3, 1)[:H, :W]
        jobs = [([names[int(rng.integers(0, len(names)))], names[int(rng.integers(0, len(names)))]], int(rng.choice([90, 90, 45, 30]))) for _ in range(int(rng.integers(1, 5)))]
        ok("part_carve", pb3d.part_carve(col, sem, jobs), oracle.part_carve(col, sem, jobs), info + (jobs,))
        binary = (~np.all(sem == pal[9], axis=-1)).astype(np.uint8)
        ga = int(rng.choice([90, 90, 45]))
        ok("global_carve", pb3d.global_carve(binary, sem, ga), oracle.global_carve(binary, sem, ga), info + (ga,))
        sel = [names[i] for i in rng.choice(len(names), int(rng.integers(1, len(names))), replace=False)]
        gp, gc = pb3d.get_voxel_points_by_parts(col, oracle.PART_COLORS, sel)
        op, oc = oracle.get_voxel_points_by_parts(col, oracle.PART_COLORS, sel)
        ok("points", gp, op, info); ok("points_cols", gc, oc, info)
        st = int(rng.choice([1, 1, 2, 3]))
        a = pb3d.voxel_grid_to_points(g, stride=st) if False else None
        if len(gp):
            f64 = n % 3 == 0
            cam = np.array([W / 2 + rng.normal(), H / 2 + rng.normal(), -2.5 * max(W, D)], np.float64 if f64 else np.float32)
            tgt = np.array([W / 2, H / 2, D / 2], cam.dtype)
            args = (cam, tgt, float(1.2 * max(W, H)), W / 2.0, H / 2.0, int(H + 3), int(W + 5))
            with np.errstate(all="ignore"):
                ok("project", pb3d.project_colored_voxels(gp, gc, *args), oracle.project_colored_voxels(gp, gc, *args), info + (f64,))
        if os.environ.get("FUZZ_COMPONENTS", "1") == "1" and n % 3 == 0 and W * H * D <= 400000:
            import contextlib, io
            # blob-like colour grid: a few boxes per part colour -> several 3-D components per colour
            cg = np.zeros((W, H, D, 3), np.uint8)
            for _ in range(int(rng.integers(1, 9))):
                lo = [int(rng.integers(0, s)) for s in (W, H, D)]
                hi = [min(s, l + int(rng.integers(1, max(2, s // 2)))) for s, l in zip((W, H, D), lo)]
                cg[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = pal[int(rng.integers(0, 4))]
            cg[rng.random((W, H, D)) < 0.02] = 0
            tc = tuple(int(v) for v in pal[int(rng.integers(0, 4))])
            ang = int(rng.choice([60, 90, 45]))
            b1, b2 = io.StringIO(), io.StringIO()
            with contextlib.redirect_stdout(b1):
                a = pb3d.left_right_guided_carve(cg, sem, tc, angle=ang)
            with contextlib.redirect_stdout(b2):
                b = oracle.left_right_guided_carve(cg, sem, tc, angle=ang)
            ok("lrgc", a, b, info + (tc, ang))
            if b1.getvalue() != b2.getvalue():
                print("MISMATCH lrgc log", info, flush=True); sys.exit(1)
            k = int(rng.integers(0, 5)); sa = int(rng.integers(0, 3)); nc = tuple(int(v) for v in pal[5])
            ok("recolor", pb3d.recolor_backward_components(cg, tc, nc, k=k, sort_axis=sa), oracle.recolor_backward_components(cg, tc, nc, k=k, sort_axis=sa), info + (k, sa))
            for axis in (2, 0):
                m2 = rng.random((H, W)) < 0.7
                if axis == 0 and W != D:
                    continue                                   # the reference indexes the (H,W) mask with (y,z) on axis 0: needs W == D
                dirn = "+" if rng.random() < 0.5 else "-"; dep = int(rng.integers(0, 6))
                fc = None if rng.random() < 0.5 else tuple(int(v) for v in pal[6])
                ok("extrude", pb3d.extrude_from_surface(cg, m2, axis, direction=dirn, depth=dep, fill_color=fc),
                   oracle.extrude_from_surface(cg, m2, axis, direction=dirn, depth=dep, fill_color=fc), info + (axis, dirn, dep, fc))
            # z-buffer and visibility (float32 / float64 cameras)
            for f64 in (False, True):
                dt = np.float64 if f64 else np.float32
                camd = {"cam_pos": np.array([W / 2, H / 2, -2.0 * max(W, D)], dt), "target": np.array([W / 2, H / 2, D / 2], dt),
                        "f": float(1.1 * max(W, H)), "cx": W / 2.0, "cy": H / 2.0}
                Hh, Ww = H + 2, W + 3
                with np.errstate(all="ignore"):
                    zg = pb3d.compute_global_depth_buffer(cg, camd, Hh, Ww); zo = oracle.compute_global_depth_buffer(cg, camd, Hh, Ww)
                    ok("zbuf", zg, zo, info + (f64,))
                    pp, _ = oracle.get_voxel_points_by_parts(cg, {"t": tc}, ["t"])
                    if len(pp):
                        ok("visible", pb3d.project_part_visible(pp, camd, zo, Hh, Ww), oracle.project_part_visible(pp, camd, zo, Hh, Ww), info + (f64,))
            img_a = pal[rng.integers(0, 6, (H, W))]; img_b = pal[rng.integers(0, 6, (H, W))]
            pcs = {names[i]: oracle.PART_COLORS[names[i]] for i in range(6)}
            ia, ib = pb3d.compute_partwise_iou(img_a, img_b, pcs), oracle.compute_partwise_iou(img_a, img_b, pcs)
            if repr(ia) != repr(ib):
                print("MISMATCH iou", info, ia, ib, flush=True); sys.exit(1)
            counts["iou"] = counts.get("iou", 0) + 1
            # notebook-3 deformation of one part: coordinates + colours, projection IoU, deformed grid
            plabels = {names[i]: tuple(int(v) for v in pal[i]) for i in range(4)}
            part = names[int(rng.integers(0, 4))]
            dfm = {"scale_xz": float(rng.uniform(0.6, 1.5)), "scale_y": float(rng.uniform(0.6, 1.5)), "shift_xz": float(rng.uniform(-20, 20)),
                   "shift_y": float(rng.uniform(-20, 20))}
            ishape = (int(rng.integers(20, 200)), int(rng.integers(20, 200)))
            ca, cca = pb3d.deform_part(cg, plabels, part, dfm, ishape); cb, ccb = oracle.deform_part(cg, plabels, part, dfm, ishape)
            ok("deform_coords", ca, cb, info + (part, dfm, ishape)); ok("deform_colors", cca, ccb, info)
            imgd = pal[rng.integers(0, 4, ishape)]
            camp = {"cam_pos": np.array([D / 2, H / 2, -2.0 * max(W, D)], np.float32), "target": np.array([D / 2, H / 2, W / 2], np.float32),
                    "f": float(1.1 * max(ishape)), "cx": ishape[1] / 2.0, "cy": ishape[0] / 2.0}
            with np.errstate(all="ignore"):
                pa, ia2 = pb3d.evaluate_part_deform(cg, plabels, part, dfm, imgd, camp); pb_, ib2 = oracle.evaluate_part_deform(cg, plabels, part, dfm, imgd, camp)
            ok("deform_projection", pa, pb_, info)
            if ia2 != ib2:
                print("MISMATCH deform iou", info, ia2, ib2, flush=True); sys.exit(1)
            saved = {p_: {"deform": {"scale_xz": float(rng.uniform(0.7, 1.3)), "scale_y": float(rng.uniform(0.7, 1.3)), "shift_xz": float(rng.uniform(-10, 10)),
                                      "shift_y": float(rng.uniform(-10, 10))}} for p_ in list(plabels)[:int(rng.integers(1, 5))]}
            ok("deformed_grid", pb3d.build_deformed_grid(cg, plabels, saved, ishape), oracle.build_deformed_grid(cg, plabels, saved, ishape), info)
            if n % 6 == 0 and W == D:
                # the whole notebook-1 pipeline (global_carve output -> part jobs -> component-guided carve -> extrusions -> orientation
                # -> recolouring of the back minarets) on a random blocky part image; the printed log is part of the result
                PCN = pb3d.PART_COLORS_NP
                pnames = ["full_building", "chhatris", "plinth", "front_minarets", "small_minarets", "dome", "main_door", "windows"]
                palp = np.array([PCN[q] for q in pnames] + [(0, 0, 0)], np.uint8)
                semf = palp[rng.integers(0, len(palp), (H // 4 + 1, W // 4 + 1))].repeat(4, 0).repeat(4, 1)[:H, :W]
                ext = semf.copy(); ext[np.all(ext == PCN["main_door"], axis=-1)] = PCN["full_building"]
                binm = ext.any(-1).astype(np.uint8)
                gcar = oracle.global_carve(binm, ext, 90)
                jobsn = [([pnames[int(rng.integers(0, 6))]], int(rng.choice([90, 90, 45]))) for _ in range(int(rng.integers(1, 5)))]
                sym = {pnames[int(rng.integers(0, 6))]: int(rng.choice([60, 90])) for _ in range(int(rng.integers(0, 3)))}
                extr = {pnames[int(rng.integers(6, 8))]: int(rng.integers(1, 4)) for _ in range(int(rng.integers(0, 3)))}
                b1, b2 = io.StringIO(), io.StringIO()
                with contextlib.redirect_stdout(b1):
                    fa = pb3d.partwise_carve(gcar, ext, semf, PCN, jobsn, sym, extr)
                with contextlib.redirect_stdout(b2):
                    fb = oracle.partwise_carve(gcar, ext, semf, PCN, jobsn, sym, extr)
                ok("partwise_carve", fa, fb, info + (jobsn, sym, extr))
                if b1.getvalue() != b2.getvalue():
                    print("MISMATCH partwise log", info, flush=True); sys.exit(1)
            for st in (1, 2, 3):
                gp2, gc2, _ = pb3d.voxel_grid_to_points(cg, stride=st); op2, oc2, _ = oracle.voxel_grid_to_points(cg, stride=st)
                ok("grid_to_points", gp2, op2, info + (st,)); ok("grid_to_points_cols", gc2, oc2, info + (st,))
    print(json.dumps({"seed": seed, "cases": n, "seconds": round(time.time() - t0, 1), "checks": counts}), flush=True)


if __name__ == "__main__":
    main()
