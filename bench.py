#!/usr/bin/env python3
"""Benchmark of the semantic voxel-carving hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W           (N > 1: launched by torch.distributed.run,
                                                              one rank per GPU; RANK/LOCAL_RANK/... from env)

Workload = BASELINE.json configs[3]: synthetic 16-label mask, 1024^3 semantic RGB grid (M1 of
SURVEY.md 8(d): carve_voxel_grid_with_masks(sem1024, binary), 6 algorithmic bytes per voxel).
One "step" = one carve of the whole 1024^3 grid, inputs resident in HBM.  With N ranks the grid is
cut into N slabs along axis 0 (contiguous; no data-path collective), every rank carves its slab
(strong scaling), the timed region is K steps between barrier + device sync on both sides and the
MAX over ranks is taken.  The slab all-gather (RCCL over xGMI) that reassembles the carved volume is
timed separately in the same run and reported in "allgather" (see DESIGN.md for why it cannot be
inside `value`).  No PyTorch anywhere: ctypes -> libpb3d.so -> hand-written HIP kernels.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "part-based-3d-reconstruction_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
INFINITY_CACHE_BYTES = 256 << 20   # memory-side cache in front of the HBM (MI355X_MICROARCH.md)
ALG_BYTES_PER_VOXEL = 6        # M1: read 3 + write 3


def cpu_baseline(S, d_in, m_wh, planes):
    """The CPU oracle (oracle/pb3d_oracle.c, a port of the reference's np.where carve) timed on this
    box's host cores on a bounded sample: the first `planes` X-planes of the same synthetic grid."""
    from oracle import oracle as orc
    sample = d_in.download((planes, S, S, 3))
    mask = np.ascontiguousarray(m_wh[:planes])
    nvox = planes * S * S
    # the GPU box grants a 16-CPU share per GPU: never spin up more OpenMP threads than that
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    orc.set_threads(max(1, min(16, avail)))
    cores = orc.get_threads()
    best = None
    t_all = time.perf_counter()
    out = None
    for _ in range(5):
        t0 = time.perf_counter()
        out = None                       # one 3 GB result at a time
        out = orc.carve_voxel_grid_with_masks(sample, mask if planes != S else mask.T)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        if time.perf_counter() - t_all > 20:
            break
    # the expression the reference itself executes (utils/voxel_carving_utils.py:87), single-threaded NumPy
    t0 = time.perf_counter()
    ref_form = np.where(mask[:, :, None, None].astype(bool), sample, 0)
    t_np = time.perf_counter() - t0
    assert np.array_equal(ref_form, out)
    del ref_form
    return {"value": round(nvox / best / 1e6, 1), "unit": "Mvoxel/s", "cores": cores, "kind": "port",
            "sample": (f"the whole {S}^3 synthetic grid" if planes == S else f"first {planes} of {S} X-planes of the same synthetic grid") +
                      f" ({nvox / 1e6:.0f} Mvoxel, best of <=5 passes; the 1-core np.where form once)",
            "numpy_where_1core_Mvoxel_s": round(nvox / t_np / 1e6, 1)}, out


def source_digest(rel):
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(PKG, "csrc", rel), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_traffic(kernel_substr, nvox, source_file):
    """(HBM bytes per launch, where the figure comes from).  The bytes are PMC counters of a rocprofv3 run of this same
    command, summarised into profiles/pmc_traffic.json by tools/pmc_summary.py together with a digest of the kernel's source
    file; a record made from a different version of the kernel is NOT reported (traffic = null) -- counters cannot be read
    from inside the run, so the only honest alternatives are a fresh profile or no number."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, "no profiles/pmc_traffic.json"
    try:
        for rec in json.load(open(path)):
            if kernel_substr in rec.get("kernel", "") and rec.get("voxels_per_launch") == nvox:
                have, want = rec.get("source_sha256"), source_digest(source_file)
                if have is None or have != want:
                    return None, (f"stale: profiles/pmc_traffic.json ({rec.get('tag', '')}) was measured on another version of "
                                  f"csrc/{source_file}; re-run tools/pmc_summary.py")
                return rec.get("hbm_bytes_per_launch"), f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {rec.get('tag', '')} (csrc/{source_file} {have})"
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable profiles/pmc_traffic.json: {e}"
    return None, "no record for this kernel and launch size"


def compact_gather_phase(cp, dev, pdist, args, rank, world, S, planes, x0, d_in, d_mwh, d_full, total_vox):
    """The compact form of the reassembly: the slab is carved as 1-byte LABELS, ONE ncclAllGather moves
    a third of the bytes of the RGB form, and the volume is expanded to RGB locally only for consumers that need RGB.  The
    RGB -> label conversion of the input is set-up (a label-resident pipeline never holds RGB); the result is checked against
    the RGB gather that ran just before (d_full)."""
    pal = dev.synth_palette16()                                     # the 16 colours of the synthetic grid: labels 1..16, carved = 0
    slab_vox = planes * S * S
    d_lab_in = dev.DeviceBuffer(slab_vox); d_lab_full = dev.DeviceBuffer(slab_vox * world); d_rgb_full = dev.DeviceBuffer(slab_vox * world * 3)
    try:
        dev.rgb_to_label(d_in, slab_vox, pal, d_lab_in)

        def timed(fn, reps=5):
            for _ in range(2):
                fn()
            dev.sync(); cp.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            dev.sync()
            return cp.allreduce_max(time.perf_counter() - t0) / reps

        t_lab = timed(lambda: pdist.carve_labels_sharded(d_lab_in, S, S, S, d_mwh, d_lab_full))
        t_rgb = timed(lambda: pdist.carve_labels_sharded(d_lab_in, S, S, S, d_mwh, d_lab_full, pal, d_rgb_full))
        same = True
        for r in range(world):                                      # one plane out of every rank's slab: expanded labels == gathered RGB
            off = (r * planes + planes // 2) * S * S * 3
            same = same and bool(np.array_equal(d_rgb_full.download((S, S, 3), byte_offset=off), d_full.download((S, S, 3), byte_offset=off)))
        ok_all = cp.allreduce_max(0 if same else 1) == 0
        return {"form": "u8 label slabs, in place, one ncclAllGather; RGB expanded locally on every rank", "bytes_per_rank": slab_vox,
                "ms_carve_plus_label_allgather": round(t_lab * 1e3, 4), "value_incl_label_allgather_Mvoxel_s": round(total_vox / t_lab / 1e6, 1),
                "ms_with_local_rgb_expansion": round(t_rgb * 1e3, 4), "value_incl_allgather_and_expansion_Mvoxel_s": round(total_vox / t_rgb / 1e6, 1),
                "equals_rgb_allgather": ok_all}
    finally:
        for b in (d_lab_in, d_lab_full, d_rgb_full):
            b.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=1024, help="grid edge (1024 = the BASELINE metric's config)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-planes", type=int, default=0, help="X-planes of the grid the CPU baseline is timed on (0 = all of them)")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the rotation-chain figures reported next to the headline (N = 1 only)")
    ap.add_argument("--slab-planes", type=int, default=0, help="PROFILING MODE (one process): carve only the first P X-planes of the grid -- exactly "
                    "what rank 0 of an N = size / P GPU run does -- so that PMC traffic of the slab sizes of N = 2, 4, 8 can be recorded on one GPU "
                    "(tools/slab_pmc.sh -> profiles/pmc_traffic.json); the JSON line is marked and is not a bench result")
    ap.add_argument("--weak", action="store_true", help="weak scaling: every rank carves its own full S^3 grid (default: the ONE "
                    "S^3 grid of the BASELINE metric is split into X-slabs = strong scaling)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # PB3D_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank control flow on a 1-GPU box (all ranks on GPU 0, no RCCL)
    one_device = os.environ.get("PB3D_BENCH_ONE_DEVICE") == "1"
    os.environ["PB3D_DEVICE"] = "0" if one_device else str(local_rank)
    if one_device:
        args.no_allgather = True

    import pb3d
    if not os.path.exists(pb3d._lib.LIB_PATH):      # fresh checkout: compile the HIP extension in tree first (rank 0 only)
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build(quiet=True)
        else:
            for _ in range(600):
                if os.path.exists(pb3d._lib.LIB_PATH):
                    break
                time.sleep(1)
    from pb3d import device as dev
    from pb3d import dist as pdist
    from pb3d.rendezvous import ControlPlane

    cp = ControlPlane(rank, world)
    S = args.size
    pb3d._lib.ctx()
    info = dev.device_info()

    planes = S if args.weak else pdist.equal_slabs(S, world)
    x0, x1 = (0, S) if args.weak else (rank * planes, (rank + 1) * planes)
    if args.slab_planes:
        if world != 1 or not 0 < args.slab_planes <= S:
            sys.exit("--slab-planes is a one-process profiling mode with 0 < P <= size")
        planes, x0, x1 = args.slab_planes, 0, args.slab_planes
        args.no_extras = args.no_cpu_baseline = True
    if args.weak:
        args.seed += rank
        args.no_allgather = True        # independent grids: nothing to reassemble
    slab_vox = planes * S * S
    slab_bytes = slab_vox * 3
    d_in = dev.DeviceBuffer(slab_bytes)
    d_full = dev.DeviceBuffer(slab_bytes * (1 if args.weak else world))   # the reassembled volume; this rank's slab lives at its slot
    d_out = d_full.at(0 if args.weak else rank * slab_bytes)
    d_mwh = dev.DeviceBuffer(S * S)
    dev.synth_sem(x0, x1, S, S, args.seed, d_in)
    dev.synth_mask16(S, d_binary_wh=d_mwh)
    d_mslab = d_mwh.at(x0 * S)
    dev.sync()

    def step():
        dev.carve_mask(d_in, planes, S, S, 3, d_mslab, d_out)

    for _ in range(args.warmup):
        step()
    dev.sync()
    cp.barrier()
    e0, e1 = dev.Event(), dev.Event()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    dev.sync()
    t_local = time.perf_counter() - t0
    cp.barrier()
    t = cp.allreduce_max(t_local)
    kernel_ms = e1.elapsed_ms_since(e0) / args.steps     # HIP events on the stream the kernel runs on
    kernel_ms_max = cp.allreduce_max(kernel_ms)

    total_vox = planes * world * S * S
    value = total_vox * args.steps / t / 1e6
    achieved_alg = ALG_BYTES_PER_VOXEL * slab_vox / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_source = pmc_traffic("k_carve_tiles", slab_vox, "carve.hip")
    # what the kernel must move given the mask: dropped columns are never read (3 B x kept voxels in, 3 B x all voxels out)
    kept = int(np.count_nonzero(d_mwh.download((S, S))[x0:x1])) * S
    need = 3 * kept + 3 * slab_vox
    # The timed steps repeat over the SAME input and output buffers.  A per-rank working set within a few multiples of the 256 MB
    # Infinity Cache (the slabs of N = 4 and 8: 1.6 / 0.8 GB) keeps part of itself there from step to step, and algorithmic bytes
    # over time then exceed what the HBM can deliver (measured on one GPU: 7.8 / 8.4 TB/s for 256 / 128 planes,
    # profiles/r03_opbench_all_ops.jsonl M1/slab).  Such a line is marked, and its `achieved` / `frac` are priced with the bytes that
    # must cross the HBM interface at all (PMC traffic of that slab size when recorded, else the mask's keep count) -- never above
    # the algorithmic figure, which stays in `frac_algorithmic`.
    working_set = 2 * slab_bytes
    cache_assisted = working_set < 8 * INFINITY_CACHE_BYTES
    achieved = min(achieved_alg, (traffic or need) / (kernel_ms * 1e-3) / 1e9) if cache_assisted else achieved_alg
    out = {
        "metric": f"Mvoxel/s carved (semantic carve, {S}^3 grid)", "value": round(value, 1), "unit": "Mvoxel/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"configs[3]: synthetic 16-label mask, {S}^3 semantic RGB grid, op M1 "
                               f"carve_voxel_grid_with_masks(sem, binary); " +
                               (f"one such grid per GPU, {world} GPU(s)" if args.weak else f"X-slab partition over {world} GPU(s)"),
                   "grid": [S, S, S, 3], "slab_planes_per_gpu": planes, "seed": args.seed, "device": info["name"]},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "cache_assisted": cache_assisted, "frac_algorithmic": round(achieved_alg / HBM_PEAK_GBS, 4),
                     "working_set_bytes_per_rank": working_set,
                     "kernel": "k_carve_tiles", "kernel_ms": round(kernel_ms, 4), "kernel_ms_max_rank": round(kernel_ms_max, 4),
                     "algorithmic_bytes_per_launch": ALG_BYTES_PER_VOXEL * slab_vox,
                     # `achieved`/`frac` follow SURVEY 8(d): ALGORITHMIC bytes (6 B/voxel) over kernel time.  The bytes that
                     # actually cross the HBM interface are fewer (masked-out columns are never read): the same time priced
                     # with those bytes -- the measured PMC traffic when a current profile exists, else the mask's keep count
                     "hbm_bytes_needed_per_launch": need, "hbm_achieved_GB_s": round((traffic or need) / (kernel_ms * 1e-3) / 1e9, 1),
                     "hbm_frac": round((traffic or need) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "hbm_bytes_basis": "pmc" if traffic else "3 B x kept voxels + 3 B x all voxels (mask keep count)"},
    }

    if args.slab_planes:
        out["profiling_mode"] = f"--slab-planes {planes}: rank 0's slab of an N = {S // planes} GPU run, carved by one process (not a bench result)"

    if world > 1 and not args.no_allgather:
        # The reassembly all-gather is measured AFTER the headline figure is final.  RCCL start-up is the one step of this
        # script that depends on the node's fabric state.  If it blocks inside the library, a watchdog thread prints the JSON
        # line with the failure recorded (rank 0) and ends EVERY rank with exit code 3: a hang is never reported as success,
        # and nothing tries to go on inside a process whose GPU stream is stuck.  A rank that raises inside the phase tells the
        # others through the control plane (abort), so they leave together instead of waiting in a collective.
        import threading
        from pb3d.rendezvous import ControlPlaneAbort, ControlPlaneError
        done = threading.Event()

        def watchdog():
            if not done.wait(float(os.environ.get("PB3D_ALLGATHER_TIMEOUT", "150"))):
                if rank == 0:
                    out["allgather"] = {"error": "RCCL all-gather phase timed out (process ended with exit code 3); the carve figures above were final before it started"}
                    print(json.dumps(out), flush=True)
                os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        phase_error = None
        try:
            uid = None
            if rank == 0:
                try:
                    uid = pdist.new_unique_id().tobytes()
                except Exception as e:  # noqa: BLE001 - RCCL missing / unusable: tell the other ranks to skip the phase
                    uid = f"{type(e).__name__}: {e}"
            uid = cp.broadcast(uid)
            if not isinstance(uid, bytes):
                raise RuntimeError(f"rank 0 could not create the RCCL id ({uid})")
            t_init = time.perf_counter()
            pdist.comm_init(np.frombuffer(uid, np.uint8), rank, world)
            comm_init_ms = cp.allreduce_max((time.perf_counter() - t_init) * 1e3)
            seen_rank, seen_n = pdist.comm_info()                 # asked of the communicator, not echoed from our own arguments
            nranks_seen_min = -cp.allreduce_max(-seen_n)
            out["rccl"] = {"nranks_seen": seen_n if nranks_seen_min == seen_n else nranks_seen_min, "rank_seen_on_rank0": seen_rank,
                           "comm_init_ms": round(comm_init_ms, 1)}
            for _ in range(2):
                pdist.allgather(d_out, d_full, slab_bytes)
            dev.sync(); cp.barrier()
            g0, g1 = dev.Event(), dev.Event()
            reps = 5
            g0.record()
            for _ in range(reps):
                pdist.allgather(d_out, d_full, slab_bytes)
            g1.record(); dev.sync()
            ag_ms = cp.allreduce_max(g1.elapsed_ms_since(g0) / reps)
            # carve + reassembly on one stream, back to back
            cp.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                step(); pdist.allgather(d_out, d_full, slab_bytes)
            dev.sync()
            t_both = cp.allreduce_max(time.perf_counter() - t0) / reps
            # the neighbour's slab really arrived: compare one of its planes with this rank's own carve of that plane
            nb = (rank + 1) % world
            d_chk_in = dev.DeviceBuffer(S * S * 3); d_chk_out = dev.DeviceBuffer(S * S * 3)
            dev.synth_sem(nb * planes, nb * planes + 1, S, S, args.seed, d_chk_in)
            dev.carve_mask(d_chk_in, 1, S, S, 3, d_mwh.at(nb * planes * S), d_chk_out)
            dev.sync()
            same = bool(np.array_equal(d_chk_out.download((S, S, 3)), d_full.download((S, S, 3), byte_offset=nb * slab_bytes)))
            ok_all = cp.allreduce_max(0 if same else 1) == 0
            d_chk_in.free(); d_chk_out.free()
            out["allgather"] = {"form": "rgb slabs, in place, one ncclAllGather", "bytes_per_rank": slab_bytes,
                                "ms": round(ag_ms, 4), "busbw_GB_s": round(slab_bytes * (world - 1) / (ag_ms * 1e-3) / 1e9, 1),
                                "value_incl_allgather_Mvoxel_s": round(total_vox / t_both / 1e6, 1), "reassembled_volume_verified": ok_all}
            out["allgather_compact"] = compact_gather_phase(cp, dev, pdist, args, rank, world, S, planes, x0, d_in, d_mwh, d_full, total_vox)
            pdist.comm_destroy()
        except ControlPlaneAbort as e:
            phase_error = f"aborted by another rank: {e}"
        except Exception as e:  # noqa: BLE001 - the headline figure must still be reported
            phase_error = f"{type(e).__name__}: {e}"
            cp.abort(f"rank {rank}: {phase_error}")
        if phase_error is not None:
            # the ranks are no longer in step (and RCCL may be half-initialised): report, then end every rank non-zero
            if rank == 0:
                out["allgather"] = {"error": phase_error}
                print(json.dumps(out), flush=True)
            done.set()
            os._exit(3)
        try:
            cp.barrier()      # every rank leaves the phase together; still under the watchdog
        finally:
            done.set()

    if rank == 0 and world == 1 and not args.no_extras:
        # Not part of `value`: the other kernels of the path on the same 1024^3 grid, timed after the headline figure is final (HIP events,
        # 3 repetitions each, device-resident) -- the rotation chains of process_voxel_grid / global_carve (reference
        # utils/voxel_carving_utils.py:104-126, :269-298), which run bit-sliced between their steps.
        nvox = S * S * S
        d_occ = dev.DeviceBuffer(nvox); d_o1 = dev.DeviceBuffer(nvox); d_tmp = dev.DeviceBuffer(nvox)
        d_bhw = dev.DeviceBuffer(S * S); d_rgb = dev.DeviceBuffer(S * S * 3)
        dev.synth_occ(0, S, S, S, 0, d_occ)
        dev.synth_mask16(S, d_binary_hw=d_bhw, d_rgb_hw3=d_rgb)

        def ms_of(fn, reps=3):
            fn(); dev.sync()
            a, b = dev.Event(), dev.Event()
            a.record()
            for _ in range(reps):
                fn()
            b.record(); dev.sync()
            return round(b.elapsed_ms_since(a) / reps, 4)

        ex = {"note": "1024^3 occupancy / colour grids, device-resident, ms per call; sweeps = rotation steps executed (the 0-degree step is folded)"}
        for ai in (90, 45, 5):
            ex[f"process_voxel_grid_angle{ai}_ms"] = ms_of(lambda: dev.process_grid(d_occ, S, S, S, d_mwh, ai, d_o1, d_tmp))
        ex["process_voxel_grid_angle5_sweeps"] = 18
        ex["process_voxel_grid_angle5_moved_B_per_voxel"] = 2.25 + 0.25 * 18
        for ai in (90, 45):
            ex[f"global_carve_angle{ai}_ms"] = ms_of(lambda: dev.global_carve(d_bhw, d_rgb, S, S, ai, d_full))
        # ... and the other ops of SURVEY 8(d) that round 4 rebuilt, on the colour grid global_carve(., ., 90) leaves: part_carve with the six
        # 90-degree jobs of notebook 1 (reference :139-160; ONE launch of k_part90_plane), point extraction of all ten parts (utils/voxel_utils.py:7-21;
        # count + fill), the notebook's orientation (:384-385) and the labelling of one part colour with its statistics (:175; labels for the
        # colour's voxels only, what partwise_carve uses).  Never part of `value`; a failure here is recorded, the headline line still leaves.
        try:
            import ctypes as C
            L, lib = pb3d._lib, pb3d._lib.load()
            d_col = dev.DeviceBuffer(nvox * 3); d_pc = dev.DeviceBuffer(nvox * 3)
            dev.global_carve(d_bhw, d_rgb, S, S, 90, d_col)
            rgb_hw = d_rgb.download((S, S, 3))
            names = ["full_building", "chhatris", "plinth", "front_minarets", "small_minarets", "dome"]
            msub = np.zeros((len(names), S, S), np.uint8)
            for j, nm in enumerate(names):
                msub[j] = np.all(rgb_hw == np.array(pb3d.PART_COLORS[nm], np.uint8), axis=-1).T
            mcarve = np.ascontiguousarray(msub.transpose(0, 2, 1))          # W == H: upstream's _mask_to_wh transposes a square mask again
            d_ms = dev.from_numpy(msub); d_mc = dev.from_numpy(mcarve)
            ang = (C.c_int * 6)(*([90] * 6)); skip = (C.c_int * 6)(*[0 if msub[j].any() else 1 for j in range(6)])
            ex["part_carve_six_90deg_jobs_ms"] = ms_of(lambda: L.check(lib.pb3d_part_carve_dev(
                L.ctx(), C.c_void_p(d_col.ptr), S, S, S, C.c_void_p(d_ms.ptr), C.c_void_p(d_mc.ptr), ang, skip, 6, C.c_void_p(d_pc.ptr))))
            ex["orient_ms"] = ms_of(lambda: L.check(lib.pb3d_orient_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, C.c_void_p(d_pc.ptr))))
            cols = np.ascontiguousarray(np.array(list(pb3d.PART_COLORS.values()), np.uint8))
            npt = C.c_int64(0)
            cnt = lambda: L.check(lib.pb3d_points_count_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, 3, L.p_u8(cols), len(cols), 1, C.byref(npt)))
            cnt()
            d_pts = dev.DeviceBuffer(max(1, npt.value) * 12); d_pcl = dev.DeviceBuffer(max(1, npt.value) * 3)

            def extract():
                cnt()
                L.check(lib.pb3d_points_fill_dev(L.ctx(), C.c_void_p(d_col.ptr), S, S, S, 3, L.p_u8(cols), len(cols), 1, npt.value,
                                                 C.c_void_p(d_pts.ptr), C.c_void_p(d_pcl.ptr)))
            ex["points_by_parts_count_fill_ms"] = ms_of(extract)
            ex["points_by_parts_points"] = int(npt.value)
            from pb3d.voxel_carving_utils import _label_stats
            d_lab = dev.DeviceBuffer(nvox * 4)
            fb = np.array(pb3d.PART_COLORS["full_building"], np.uint8)
            ex["label_one_colour_members_only_with_stats_ms"] = ms_of(lambda: _label_stats(d_col, (S, S, S), fb, d_lab, members_only=True))
            for b in (d_col, d_pc, d_ms, d_mc, d_pts, d_pcl, d_lab):
                b.free()
        except Exception as e:  # noqa: BLE001 - extras never take the headline line down
            ex["extras_error"] = f"{type(e).__name__}: {e}"
        dev.carve_mask(d_in, planes, S, S, 3, d_mslab, d_out)      # d_full holds the carve again (the CPU check below reads it)
        dev.sync()
        out["extras"] = ex
        for b in (d_occ, d_o1, d_tmp, d_bhw, d_rgb):
            b.free()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        lab_planes = min(S, args.cpu_planes) if args.cpu_planes > 0 else S      # default: the WHOLE grid
        m_wh = d_mwh.download((S, S))
        base, cpu_out = cpu_baseline(S, d_in, m_wh, lab_planes)
        gpu_out = d_full.download((lab_planes, S, S, 3))
        assert np.array_equal(gpu_out, cpu_out), "GPU carve differs from the CPU oracle on the sample"
        base["gpu_matches_on_sample"] = True
        out["cpu_baseline"] = base
    cp.barrier()
    if rank == 0:
        print(json.dumps(out))
    cp.close()
    for b in (d_in, d_full, d_mwh):
        b.free()


if __name__ == "__main__":
    main()
