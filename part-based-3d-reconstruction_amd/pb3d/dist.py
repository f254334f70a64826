"""Slab sharding of the voxel grid across the GPUs of one node and its reassembly.

The carve path is independent per (x,y) column, so the grid is cut into slabs along its
slowest axis (axis 0: X in the (W,H,D) working orientation -- the Z axis of the stored
(D,H,W,3) artefacts).  Slabs are contiguous in memory, every rank runs the same kernels on
its slab with no communication, and ONE RCCL all-gather reassembles the volume.  One
process per GPU; the unique id travels over whatever control plane the launcher provides
(bench.py uses a torch.distributed gloo group only for that and for barriers).
"""
import ctypes as C

import numpy as np

from . import _lib


def slab_bounds(n_planes, rank, nranks):
    """[x0, x1) of rank's slab: planes are dealt as evenly as possible, the first
    n_planes % nranks ranks get one extra plane."""
    if not (0 <= rank < nranks):
        raise ValueError(f"rank {rank} outside 0..{nranks - 1}")
    base, extra = divmod(int(n_planes), int(nranks))
    x0 = rank * base + min(rank, extra)
    return x0, x0 + base + (1 if rank < extra else 0)


def equal_slabs(n_planes, nranks):
    """Planes per rank when the partition is even (what a single all-gather needs)."""
    if n_planes % nranks:
        raise ValueError(f"{n_planes} planes do not split evenly over {nranks} ranks; an all-gather needs equal slabs")
    return n_planes // nranks


def new_unique_id():
    uid = np.zeros(128, np.uint8)
    _lib.check(_lib.load().pb3d_comm_unique_id(_lib.p_u8(uid)))
    return uid


def comm_init(unique_id, rank, nranks):
    uid = np.ascontiguousarray(unique_id, np.uint8)
    _lib.check(_lib.load().pb3d_comm_init(_lib.ctx(), _lib.p_u8(uid), int(rank), int(nranks)))


def allgather(d_send, d_recv, bytes_per_rank):
    """Enqueue the slab all-gather on the context stream (d_send may be the rank's slot of d_recv)."""
    from .device import _ptr
    _lib.check(_lib.load().pb3d_allgather_dev(_lib.ctx(), _ptr(d_send), _ptr(d_recv), int(bytes_per_rank)))


def comm_destroy():
    _lib.check(_lib.load().pb3d_comm_destroy(_lib.ctx()))


def carve_sharded_reference_layout(carve_slab, gather, W, rank, nranks):
    """Host-level description of the sharded carve used by the CPU (gloo) tests and bench.py:
    carve_slab(x0, x1) -> this rank's carved slab; gather(slab) -> full volume."""
    x0, x1 = slab_bounds(W, rank, nranks)
    return gather(carve_slab(x0, x1))
