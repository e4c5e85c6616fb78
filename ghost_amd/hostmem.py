"""Page-locked host arrays for results (include/ghostcwt.h: gcwt_host_alloc).

A result of the reference's size -- 856 MB of float64 for one channel x 1e6 samples x 107 scales
(transforms.py:185) -- costs more in first-touch page faults than in PCIe time when it lands in a fresh
``np.empty``.  ``empty(shape, dtype)`` hands out NumPy arrays whose memory is page-locked: the device's copy
engines write them directly, and a block goes back to a small pool when the last array that views it is
dropped, so that the next result of that size reuses memory that is already mapped.  ``limit_bytes`` bounds
what the pool keeps (and the size of a single pinned request); beyond it callers fall back to pageable arrays.
"""
import ctypes as C
import threading

import numpy as np

from ._lib import lib

__all__ = ["empty", "is_pinned", "limit_bytes", "trim"]

limit_bytes = 4 << 30          # kept by the pool at most, and the largest single pinned request
_GRANULE = 2 << 20
_lock = threading.Lock()
_free = []                     # [(nbytes, ptr)]
_kept = 0


class _Block:
    """One gcwt_host_alloc allocation; returned to the pool (or freed) when the last view of it dies."""
    __slots__ = ("ptr", "nbytes")

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def __del__(self):
        global _kept
        try:
            with _lock:
                if _kept + self.nbytes <= limit_bytes:
                    _free.append((self.nbytes, self.ptr))
                    _kept += self.nbytes
                    return
            lib.gcwt_host_free(C.c_void_p(self.ptr))
        except Exception:                      # interpreter shutdown: the process's memory goes with it
            pass


def _take(nbytes):
    global _kept
    need = -(-nbytes // _GRANULE) * _GRANULE
    with _lock:
        best = None
        for i, (n, _) in enumerate(_free):
            if need <= n <= need + need // 4 and (best is None or n < _free[best][0]):
                best = i
        if best is not None:
            n, ptr = _free.pop(best)
            _kept -= n
            return _Block(ptr, n)
    ptr = C.c_void_p()
    if lib.gcwt_host_alloc(C.byref(ptr), need) != 0 or not ptr.value:
        return None
    return _Block(ptr.value, need)


def empty(shape, dtype):
    """Page-locked ndarray, or None when the request is over ``limit_bytes`` or the allocation fails."""
    dtype = np.dtype(dtype)
    count = int(np.prod(shape, dtype=np.int64))
    nbytes = count * dtype.itemsize
    if nbytes == 0 or nbytes > limit_bytes:
        return None
    blk = _take(nbytes)
    if blk is None:
        return None
    buf = (C.c_char * nbytes).from_address(blk.ptr)
    buf._block = blk                            # the ctypes view keeps the block; NumPy keeps the view
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


def is_pinned(arr):
    """True for arrays (and views of arrays) made by ``empty``."""
    base = arr
    while isinstance(base, np.ndarray) and base.base is not None:
        base = base.base
    if isinstance(base, memoryview):
        base = base.obj
    return hasattr(base, "_block")


def trim():
    """Frees what the pool holds (blocks still viewed by live arrays are not touched)."""
    global _kept
    with _lock:
        blocks, _free[:] = list(_free), []
        _kept = 0
    for _, ptr in blocks:
        lib.gcwt_host_free(C.c_void_p(ptr))
