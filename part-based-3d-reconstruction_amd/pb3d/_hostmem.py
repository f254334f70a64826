"""Host memory for the arrays the shim returns.

A download into a FRESH np.empty array is dominated by first-touch page faults, not by PCIe: measured on the MI355X hosts,
219 MB arrive in 4.1 ms when the destination pages exist and in 18.5 ms when they do not (glibc returns arrays above 32 MiB
to the OS on free, so every np.empty faults again; parallel pre-touching, MADV_HUGEPAGE and MAP_POPULATE did not help).  The
pool below keeps the memory of results the caller has RELEASED and hands it out again for later results.

    pb3d.set_result_pool(4096)        # megabytes the pool may hold; 0 (the default) = plain np.empty every time
    # or: PB3D_RESULT_POOL_MB=... in the environment

The pool is OPT-IN.  A pooled result is an ordinary writable ndarray whose ultimate .base is a `_Lease` object (flags.owndata
is False -- the one observable difference).  Ownership does not depend on interpreter internals: every result is created
through the lease's buffer export, so NumPy itself keeps the lease alive for as long as the result or anything derived from it
(slice, reshape, transpose, view ...) exists; when the last of them dies the lease is collected and its `weakref.finalize`
callback puts the backing pages on the free list.  Memory can therefore never be handed out while a live array still maps it.
"""
import os
import threading
import weakref

import numpy as np

_lock = threading.RLock()      # re-entrant: a finalizer (_give_back) may run on this thread inside empty() when the GC fires there
_cap_bytes = 0
_free = []          # released backing buffers (np.uint8 arrays), most recently released last
_leased_bytes = 0   # bytes of backing buffers currently lent to live results


class _Lease:
    """Exports one backing buffer; lives exactly as long as the arrays made from it."""
    __slots__ = ("__array_interface__", "__weakref__")

    def __init__(self, back):
        self.__array_interface__ = back.__array_interface__


def _give_back(back, generation):
    global _leased_bytes
    with _lock:
        _leased_bytes -= back.nbytes
        if generation == _generation and _cap_bytes and _free_bytes() + back.nbytes <= _cap_bytes:
            _free.append(back)


_generation = 0


def set_result_pool(max_megabytes):
    """Let the pool hold up to max_megabytes of released result memory (0 disables it and drops what it holds)."""
    global _cap_bytes, _generation
    with _lock:
        _cap_bytes = max(0, int(max_megabytes)) << 20
        _generation += 1                      # buffers lent under the old setting are not taken back
        _free.clear()


def _free_bytes():
    return sum(b.nbytes for b in _free)


def stats():
    """(bytes on the free list, bytes lent to live results) -- for tests and diagnostics."""
    with _lock:
        return _free_bytes(), _leased_bytes


def empty(shape, dtype=np.uint8):
    """np.empty(shape, dtype), served from released result memory when the pool is enabled and has a fitting buffer."""
    global _leased_bytes
    dtype = np.dtype(dtype)
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if _cap_bytes == 0 or nbytes < (1 << 20) or nbytes > _cap_bytes:
        return np.empty(shape, dtype)
    with _lock:
        best = -1
        for i, b in enumerate(_free):
            if nbytes <= b.nbytes <= 2 * nbytes + (1 << 20) and (best < 0 or b.nbytes < _free[best].nbytes):
                best = i
        if best >= 0:
            back = _free.pop(best)
        else:
            while _free and _free_bytes() + _leased_bytes + nbytes > _cap_bytes:
                _free.pop(0)                   # make room: drop the longest-released buffers
            if _leased_bytes + nbytes > _cap_bytes:
                return np.empty(shape, dtype)  # the caller holds everything the pool may own: plain allocation
            back = np.empty(nbytes, np.uint8)
        _leased_bytes += back.nbytes
        gen = _generation
    lease = _Lease(back)
    weakref.finalize(lease, _give_back, back, gen)       # holds `back` (not the lease) until the lease dies
    arr = np.asarray(lease)                              # .base is the lease; every view of arr keeps it alive
    del lease
    return arr[:nbytes].view(dtype).reshape(shape)


def empty_like(a):
    return empty(a.shape, a.dtype)


set_result_pool(int(os.environ["PB3D_RESULT_POOL_MB"]) if os.environ.get("PB3D_RESULT_POOL_MB", "").isdigit() else 0)
