"""Page-locked host arrays for results (include/ghostcwt.h: gcwt_host_alloc).

A result of the reference's size -- 856 MB of float64 for one channel x 1e6 samples x 107 scales
(transforms.py:185) -- costs more in first-touch page faults than in PCIe time when it lands in a fresh
``np.empty``.  ``empty(shape, dtype)`` hands out NumPy arrays whose memory is page-locked: the device's copy
engines write them directly, and a block goes back to a small pool when the last array that views it is
dropped, so that the next result of that size reuses memory that is already mapped.  ``limit_bytes`` bounds
what the pool keeps (and the size of a single pinned request); beyond it callers fall back to pageable arrays.
"""
import collections
import ctypes as C
import os
import threading

import numpy as np

from ._lib import lib

__all__ = ["empty", "is_pinned", "limit_bytes", "trim"]

limit_bytes = 4 << 30          # kept by the pool at most, and the largest single pinned request
_GRANULE = 2 << 20
_lock = threading.Lock()
_free = []                     # [(nbytes, ptr)]
_kept = 0
# Blocks whose last view died, waiting to be sorted into the pool.  __del__ may run inside ANY allocation of the thread
# that holds _lock (the cyclic collector is triggered by allocations), so it must not take that lock: deque.append is
# atomic, and _take / trim drain the deque under the lock.
_returned = collections.deque()


class _Block:
    """One gcwt_host_alloc allocation; returned to the pool (or freed) when the last view of it dies."""
    __slots__ = ("ptr", "nbytes")

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def __del__(self):
        try:
            _returned.append((self.nbytes, self.ptr))      # lock-free: see _returned
        except Exception:                      # interpreter shutdown: the process's memory goes with it
            pass


def _drain():
    """Sorts the returned blocks into the pool up to ``limit_bytes``; the rest is handed back to free.  _lock held."""
    global _kept
    over = []
    while True:
        try:
            n, ptr = _returned.popleft()
        except IndexError:
            break
        if _kept + n <= limit_bytes:
            _free.append((n, ptr))
            _kept += n
        else:
            over.append(ptr)
    return over


def _take(nbytes):
    global _kept
    need = -(-nbytes // _GRANULE) * _GRANULE
    with _lock:
        over = _drain()
    for ptr in over:
        lib.gcwt_host_free(C.c_void_p(ptr))
    with _lock:
        best = None
        for i, (n, _) in enumerate(_free):
            if need <= n <= need + need // 4 and (best is None or n < _free[best][0]):
                best = i
        if best is not None:
            n, ptr = _free.pop(best)
            _kept -= n
            return _Block(ptr, n)
    # the pages are placed where the allocating thread runs: for the time of the allocation it stays on the host cores
    # of the current device's NUMA node (a copy across sockets runs at half the link's rate), best effort
    old = _beside_device()
    try:
        ptr = C.c_void_p()
        if lib.gcwt_host_alloc(C.byref(ptr), need) != 0 or not ptr.value:
            return None
    finally:
        if old is not None:
            try:
                os.sched_setaffinity(0, old)
            except OSError:
                pass
    return _Block(ptr.value, need)


_node_cpus = {}


def _beside_device():
    """Moves the calling thread onto the cores of the current device's NUMA node (within the process's affinity);
    returns the previous affinity to restore, or None when nothing was changed."""
    try:
        dev = C.c_int(-1)
        if lib.gcwt_current_device(C.byref(dev)) != 0:
            return None
        if dev.value not in _node_cpus:
            from .dist import parse_cpulist
            buf = C.create_string_buffer(64)
            cpus = None
            if lib.gcwt_device_pci_bus_id(dev.value, buf, 64) == 0:
                with open("/sys/bus/pci/devices/%s/numa_node" % buf.value.decode().lower()) as fh:
                    node = int(fh.read().strip())
                if node >= 0:
                    with open("/sys/devices/system/node/node%d/cpulist" % node) as fh:
                        cpus = parse_cpulist(fh.read())
            _node_cpus[dev.value] = cpus
        cpus = _node_cpus[dev.value]
        old = os.sched_getaffinity(0)
        if not cpus or not (cpus & old) or (cpus & old) == old:
            return None
        os.sched_setaffinity(0, cpus & old)
        return old
    except Exception:
        return None


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
        over = _drain()
        blocks, _free[:] = list(_free), []
        _kept = 0
    for ptr in over + [ptr for _, ptr in blocks]:
        lib.gcwt_host_free(C.c_void_p(ptr))
