"""CPU-only, world_size 2 over gloo: the slab sharding of the carve path and its single all-gather
reassembly give byte-for-byte the single-process result.  Each rank carves its slab with the CPU
oracle standing in for the device kernel (the partition / reassembly logic is what is under test;
the per-slab kernels are checked against the same oracle in the GPU parity tests)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    try:
        for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist
        import synth_host
        from oracle import oracle as orc
        from pb3d.dist import slab_bounds
        orc.set_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        S = 32
        lab, binary, rgb = synth_host.mask16(S)
        # (a) M1: semantic carve of a sharded RGB grid, even partition -> one all_gather
        x0, x1 = slab_bounds(S, rank, world)
        slab_in = synth_host.sem_slab(x0, x1, S, S, seed=1)
        m_wh = np.ascontiguousarray(binary.T)
        slab_out = orc.carve_voxel_grid_with_masks(slab_in, m_wh[x0:x1])     # mask rows of this slab's planes
        parts = [torch.empty_like(torch.from_numpy(slab_out)) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(slab_out))
        full = np.concatenate([p.numpy() for p in parts], axis=0)
        # full grid is square in (W,H): the (H,W) test of _mask_to_wh wins, so hand it the (H,W) image
        want = orc.carve_voxel_grid_with_masks(synth_host.sem_slab(0, S, S, S, seed=1), binary)
        ok_a = np.array_equal(full, want)
        # (b) global_carve: output X-slab of a replicated-input job (the slab of the full result)
        g_full = orc.global_carve(binary, rgb, 90)
        mine = np.ascontiguousarray(g_full[x0:x1])
        parts = [torch.empty_like(torch.from_numpy(mine)) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(mine))
        ok_b = np.array_equal(np.concatenate([p.numpy() for p in parts], axis=0), g_full)
        # (c) uneven partition (S=31 planes over 2 ranks): bounds still tile the axis exactly
        b = [slab_bounds(31, r, world) for r in range(world)]
        ok_c = b[0][0] == 0 and b[-1][1] == 31 and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        # (e) points partition of the projection: per-rank key images ((global index + 1) << 24 | rgb), ONE all_reduce(MAX),
        #     local resolve == the unsharded projection (last writer wins)
        rng = np.random.default_rng(5)
        N = 4000
        pts = rng.integers(0, 12, (N, 3)).astype(np.float32); cols = rng.integers(0, 256, (N, 3), dtype=np.uint8)
        cam = np.array([6, 5, -30], np.float32); tgt = np.array([6, 6, 6], np.float32)
        Hh, Ww = 40, 48
        want = orc.project_colored_voxels(pts, cols, cam, tgt, 80.0, 24.0, 20.0, Hh, Ww)
        from pb3d.dist import point_shard_bounds, resolve_keys
        i0, i1 = point_shard_bounds(N, rank, world)
        keys = np.zeros((Hh, Ww), np.int64)
        for i in range(i0, i1):     # the shard's points one by one: which pixel each hits (oracle, single point)
            one = orc.project_colored_voxels(pts[i:i + 1], np.full((1, 3), 255, np.uint8), cam, tgt, 80.0, 24.0, 20.0, Hh, Ww)
            hit = np.argwhere(one.any(-1))
            if len(hit):
                v, u = hit[0]
                key = ((i + 1) << 24) | (int(cols[i, 2]) << 16) | (int(cols[i, 1]) << 8) | int(cols[i, 0])
                keys[v, u] = max(keys[v, u], key)
        tk = torch.from_numpy(keys)
        dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        ok_e = np.array_equal(resolve_keys(tk.numpy().astype(np.uint64)), want)
        # (f) Y-slab partition (planes are independent: rotation about Y, masks per (x,y)): a chained 45-degree
        #     process_voxel_grid, global_carve and part_carve run per rank on its planes with its mask rows, no exchange;
        #     one all_gather of the (equal) sub-volumes, concatenated along axis 1 == the unsharded result
        from pb3d.dist import y_slab_image, y_slab_grid, assemble_y_slabs

        def gathered(mine):
            parts = [torch.empty_like(torch.from_numpy(mine)) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(mine)))
            return assemble_y_slabs([p.numpy() for p in parts])
        occ = synth_host.occ_slab(0, S, S, S, seed=3)
        ok_f = np.array_equal(gathered(orc.process_voxel_grid(y_slab_grid(occ, rank, world), y_slab_image(binary, rank, world), 45)),
                              orc.process_voxel_grid(occ, binary, 45))
        ok_f = ok_f and np.array_equal(gathered(orc.global_carve(y_slab_image(binary, rank, world), y_slab_image(rgb, rank, world), 90)), g_full)
        # part_carve on a NON-square grid (W = 32, H = 24): with W == H the reference's _mask_to_wh transposes the already
        # transposed part mask once more (a quirk pb3d mirrors), which a slab with H_r != W cannot reproduce
        jobs = [(["full_building"], 90), (["chhatris", "plinth"], 45)]
        bin24, rgb24 = np.ascontiguousarray(binary[4:28]), np.ascontiguousarray(rgb[4:28])
        g24 = orc.global_carve(bin24, rgb24, 90)                                    # (32, 24, 32, 3)
        ok_f = ok_f and np.array_equal(gathered(orc.part_carve(y_slab_grid(g24, rank, world), y_slab_image(rgb24, rank, world), jobs)),
                                       orc.part_carve(g24, rgb24, jobs))
        # (g) compact reassembly: the slab is carved as 1-byte LABELS, ONE all_gather moves a third of the bytes, every
        #     rank expands the reassembled label volume locally: == the RGB gather of (a), byte for byte
        pal16 = synth_host.palette16()                                         # the 16 colours of the synthetic grid; label k <-> pal16[k-1]
        table = np.concatenate([np.zeros((1, 3), np.uint8), pal16])
        key = lambda a: a[..., 0].astype(np.int64) | (a[..., 1].astype(np.int64) << 8) | (a[..., 2].astype(np.int64) << 16)
        lut = {int(key(table[k])): k for k in range(len(table))}
        slab_lab = np.vectorize(lut.__getitem__, otypes=[np.uint8])(key(slab_in))
        lab_out = orc.carve_voxel_grid_with_masks(slab_lab, m_wh[x0:x1])        # the same carve op, C = 1
        parts = [torch.empty_like(torch.from_numpy(lab_out)) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(lab_out))
        ok_g = np.array_equal(table[np.concatenate([p.numpy() for p in parts], axis=0)], full)
        # (d) max-over-ranks timing reduction used by bench.py
        t = torch.tensor([0.25 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok_d = float(t) == 0.25 + world - 1
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, ok_a, ok_b, ok_c, ok_d and ok_e and ok_f and ok_g))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc(), str(e), None))


@pytest.mark.timeout(300)
def test_sharded_carve_world2_gloo():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1:] == (True, True, True, True), r
