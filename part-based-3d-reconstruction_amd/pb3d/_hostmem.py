"""Host memory for the arrays the shim returns.

A download into a FRESH np.empty array is dominated by first-touch page faults, not by PCIe: measured on the MI355X hosts,
219 MB arrive in 4.1 ms when the destination pages exist and in 18.5 ms when they do not (glibc returns arrays above 32 MiB
to the OS on free, so every np.empty faults again; parallel pre-touching, MADV_HUGEPAGE and MAP_POPULATE did not help).  The
pool below keeps the memory of results the caller has RELEASED and hands it out again for later results.

    pb3d.set_result_pool(4096)        # megabytes the pool may hold (default 1024); 0 = plain np.empty every time
    # or: PB3D_RESULT_POOL_MB=... in the environment

A pooled result is an ordinary writable ndarray whose .base is the pool's backing buffer (flags.owndata is False -- the one
observable difference).  A backing buffer is reused only when nothing else references it: NumPy makes every view (slice,
reshape, transpose ...) of a result point its .base at the backing buffer, so CPython's reference count of the buffer says
exactly whether the caller still holds the result or anything derived from it.
"""
import os
import sys

import numpy as np

_cap_bytes = 0
_bufs = []          # backing uint8 arrays, most recently used last


def set_result_pool(max_megabytes):
    """Let the pool hold up to max_megabytes of released result memory (0 disables it and drops what it holds)."""
    global _cap_bytes
    _cap_bytes = max(0, int(max_megabytes)) << 20
    if _cap_bytes == 0:
        _bufs.clear()
    else:
        _trim(0)


def _held():
    return sum(b.nbytes for b in _bufs)


def _trim(extra):
    while _bufs and _held() + extra > _cap_bytes:
        for i in range(len(_bufs)):
            if sys.getrefcount(_bufs[i]) == 2:        # the list + getrefcount's argument: nobody else holds it
                del _bufs[i]
                break
        else:
            return                                      # everything is in use by the caller


def empty(shape, dtype=np.uint8):
    """np.empty(shape, dtype), served from released result memory when the pool is enabled and has a fitting buffer."""
    dtype = np.dtype(dtype)
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if _cap_bytes == 0 or nbytes < (1 << 20) or nbytes > _cap_bytes:
        return np.empty(shape, dtype)
    best = -1
    for i in range(len(_bufs)):
        b = _bufs[i]
        if b.nbytes >= nbytes and b.nbytes <= 2 * nbytes + (1 << 20) and sys.getrefcount(b) == 3 and (best < 0 or b.nbytes < _bufs[best].nbytes):
            best = i                                    # references: the list, `b`, getrefcount's argument
        del b
    if best >= 0:
        back = _bufs.pop(best)
    else:
        _trim(nbytes)
        if _held() + nbytes > _cap_bytes:
            return np.empty(shape, dtype)               # the caller holds everything the pool owns: plain allocation
        back = np.empty(nbytes, np.uint8)
    _bufs.append(back)
    return back[:nbytes].view(dtype).reshape(shape)


def empty_like(a):
    return empty(a.shape, a.dtype)


set_result_pool(int(os.environ["PB3D_RESULT_POOL_MB"]) if os.environ.get("PB3D_RESULT_POOL_MB", "").isdigit() else 1024)
