"""Where the device sits and where this process may run: the device's NUMA node, the process's CPU affinity, the nodes'
CPU lists, and the nodes the pages of a page-locked block landed on (/proc/self/numa_maps) -- with the link rate of a
400 MB float32 result into that block."""
import sys, os, time, ctypes as C; sys.path.insert(0, '.')
import numpy as np
from ghost_amd._lib import lib
from ghost_amd import hostmem
from ghost_amd.engine import DeviceBuffer
dev = C.c_int(-1); lib.gcwt_current_device(C.byref(dev))
buf = C.create_string_buffer(64); lib.gcwt_device_pci_bus_id(dev.value, buf, 64)
bdf = buf.value.decode().lower()
node = open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip()
aff = sorted(os.sched_getaffinity(0))
print("device", dev.value, bdf, "numa node", node, "| affinity: %d cpus %s..%s" % (len(aff), aff[:3], aff[-3:]))
for n in sorted(os.listdir("/sys/devices/system/node")):
    if n.startswith("node"):
        print(" ", n, open("/sys/devices/system/node/%s/cpulist" % n).read().strip())
try:
    print("mems allowed:", [l.strip() for l in open("/proc/self/status") if l.startswith("Mems_allowed_list") or l.startswith("Cpus_allowed_list")])
except OSError:
    pass
a = hostmem.empty((100, 1000000), np.float32)
addr = a.ctypes.data
hit = [l for l in open("/proc/self/numa_maps") if int(l.split()[0], 16) <= addr < int(l.split()[0], 16) + a.nbytes + (2 << 20)]
print("pinned block:", [" ".join(w for w in l.split() if w.startswith("N") or w.startswith("bind") or w.startswith("default") or w.startswith("prefer")) for l in hit][-2:])
src = DeviceBuffer(a.nbytes)
for it in range(4):
    t0 = time.perf_counter()
    lib.gcwt_rows_to_host(src.ptr, 1000000, 100, 1000000, a.ctypes.data_as(C.c_void_p), 1000000, 16)
    dt = time.perf_counter() - t0
print("D2H 400 MB into it: %.1f ms = %.1f GB/s" % (1e3 * dt, 0.4 / dt))
# float64 through the staging ring (float32 over the link, widened by the worker pool), and the workers alone
b = hostmem.empty((100, 1000000), np.float64)
for it in range(3):
    t0 = time.perf_counter()
    lib.gcwt_rows_to_host(src.ptr, 1000000, 100, 1000000, b.ctypes.data_as(C.c_void_p), 1000000, 16 | 8)
    dt = time.perf_counter() - t0
print("D2H 400 MB widened into 800 MB: %.1f ms" % (1e3 * dt))
from concurrent.futures import ThreadPoolExecutor
a32 = np.asarray(a)
def job(k, n=8):
    r0, r1 = 100 * k // n, 100 * (k + 1) // n
    b[r0:r1] = a32[r0:r1]
with ThreadPoolExecutor(8) as ex:
    for it in range(3):
        t0 = time.perf_counter(); list(ex.map(job, range(8))); dt = time.perf_counter() - t0
print("numpy widening alone, 8 threads, pinned -> pinned: %.1f ms = %.1f GB/s written" % (1e3 * dt, 0.8 / dt))
print("threads of this process run on cpus:", sorted(os.sched_getaffinity(0))[:4], "... current cpu", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")
