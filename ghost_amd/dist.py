"""Multi-GPU control plane: one process per GPU on one node.

The data path has no collective -- channels are sharded in contiguous blocks and
each rank transforms its own block.  What ranks share is
  * the Morse filter bank, broadcast once from rank 0 with RCCL over xGMI, and
  * barriers / a max-reduce, used only to time runs.
Rendezvous (the 128-byte RCCL id) goes through a file in /tmp keyed by the
launcher's pid and MASTER_PORT, so nothing here needs torch.  If RCCL cannot be
initialised the same interface is served by files (and every rank builds its own
bank); ``Comm.backend`` says which one is live.
"""
import ctypes as C
import os
import tempfile
import time

__all__ = ["Comm", "shard_channels", "shard_time_blocks", "env_rank", "parse_cpulist", "pin_to_device_numa"]


def env_rank():
    """(rank, world_size, local_rank) from the torchrun-style environment."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


def shard_channels(n_channels, rank, world):
    """Contiguous block [start, stop) of rank's channels; sizes differ by at most 1."""
    base, extra = divmod(n_channels, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_time_blocks(segments, rank, world):
    """Few channels, long recording: rank's share of the plan's time blocks.

    ``segments`` is ``CwtPlan.segments()`` -- [(core_start, core_stop, fft_length)].  Blocks
    are dealt in contiguous runs of near-equal total length; every rank reads the whole
    recording from the host (each block carries its own halo, so no rank needs another's
    samples) and streams its runs with ``CwtPlan.execute_block``.  Returns the sample
    range (start, stop) covered by the rank, or (s, s) when it has nothing to do."""
    if not segments:
        return 0, 0
    first, last = segments[0][0], segments[-1][1]
    total = last - first
    lo_t = first + total * rank // world
    hi_t = first + total * (rank + 1) // world
    # cut at block boundaries: a block belongs to the rank its first sample falls to
    mine = [(a, b) for a, b, _ in segments if lo_t <= a < hi_t]
    if not mine:
        return lo_t, lo_t
    return mine[0][0], mine[-1][1]


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist) -> {0, 1, 2, 3, 8, 10, 11}."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_device_numa(device, sysfs="/sys"):
    """Keeps this process's threads on the host cores of the GPU's NUMA node (multi-rank runs:
    the rank's launch thread and its staging copies stay beside its device).  Best effort:
    returns (numa_node, n_cpus) or (None, 0) when sysfs does not say or the node's cores are
    outside the process's current affinity (a container's CPU share is never widened)."""
    try:
        from ._lib import lib, check
        buf = C.create_string_buffer(64)
        check(lib.gcwt_device_pci_bus_id(int(device), buf, 64))
        bus = buf.value.decode().lower()
        with open(os.path.join(sysfs, "bus/pci/devices", bus, "numa_node")) as fh:
            node = int(fh.read().strip())
        if node < 0:
            return None, 0
        with open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)) as fh:
            cpus = parse_cpulist(fh.read()) & set(os.sched_getaffinity(0))
        if not cpus:
            return None, 0
        # every thread of the process (the HIP runtime's helpers exist already by the time a rank knows its
        # device): sched_setaffinity(0, ...) alone would move the calling thread only
        try:
            tids = [int(t) for t in os.listdir("/proc/self/task")]
        except OSError:
            tids = [0]
        for tid in tids:
            try:
                os.sched_setaffinity(tid, cpus)
            except OSError:
                pass                                  # a thread that ended meanwhile
        return node, len(cpus)
    except Exception:
        return None, 0


def _launcher_start():
    """Start time (clock ticks since boot) of the parent process: with the pid it names one
    launcher instance, so a reused pid cannot pick up an older run's files."""
    try:
        with open("/proc/%d/stat" % os.getppid()) as fh:
            return fh.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


def _session_dir():
    key = "%s_%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.getppid(), _launcher_start(),
                           os.environ.get("TORCHELASTIC_RUN_ID", "none"))
    d = os.path.join(os.environ.get("GHOSTCWT_RDZV_DIR", tempfile.gettempdir()),
                     "ghostcwt_" + key)
    os.makedirs(d, exist_ok=True)
    return d


def _publish(path, data):
    tmp = path + ".tmp.%d" % os.getpid()
    with open(tmp, "wb") as fh:
        fh.write(data)
    os.replace(tmp, path)


def _wait_for(path, timeout):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError("rendezvous file %s did not appear" % path)
        time.sleep(0.0005)
    with open(path, "rb") as fh:
        return fh.read()


class Comm:
    """barrier(), allreduce_max(x), broadcast_bank(plan) over ``world`` ranks."""

    def __init__(self, rank=None, world=None, *, use_rccl=True, timeout=120.0, session=None,
                 device=None):
        if rank is None:
            rank, world, _ = env_rank()
        self.rank, self.world, self.timeout = rank, world, timeout
        self.device = device          # HIP ordinal of this rank (the current device is per thread)
        self.dir = session or (_session_dir() if world > 1 else None)
        self._seq = 0
        self._handle = None
        self._rccl_abandoned = False
        self.backend = "single" if world == 1 else "file"
        self.rccl_error = None
        if os.environ.get("GHOSTCWT_COMM", "") == "file":   # rehearsals on a single GPU
            use_rccl = False
        if world > 1 and use_rccl:
            # communicator set-up in a helper thread: if RCCL's bootstrap hangs (it has no
            # timeout of its own) the run carries on with the file backend instead
            import threading
            th = threading.Thread(target=self._try_rccl, daemon=True)
            th.start()
            th.join(float(os.environ.get("GHOSTCWT_RCCL_TIMEOUT", "90")))
            if th.is_alive():
                self.rccl_error = "RCCL communicator set-up timed out"
                self._handle = None
                self._rccl_abandoned = True
        if world > 1:
            # all ranks must agree on the backend
            ok = self._file_allreduce_max(1.0 if self._handle else 0.0, "agree_min", negate=True)
            if ok < 1.0 and self._handle:
                self._drop_rccl()
            self.backend = "rccl" if self._handle else "file"

    # -- RCCL ---------------------------------------------------------------
    def _try_rccl(self):
        try:
            from ._lib import lib, check, COMM_ID_BYTES
            if self.device is not None:
                check(lib.gcwt_set_device(int(self.device)))
            path = os.path.join(self.dir, "rccl_id")
            if self.rank == 0:
                buf = C.create_string_buffer(COMM_ID_BYTES)
                check(lib.gcwt_comm_unique_id(buf))
                _publish(path, buf.raw)
                ident = buf.raw
            else:
                ident = _wait_for(path, self.timeout)
            h = C.c_void_p()
            check(lib.gcwt_comm_create(C.byref(h), self.rank, self.world,
                                       C.create_string_buffer(ident, COMM_ID_BYTES)))
            if not self._rccl_abandoned:
                self._handle = h
            else:                         # finished after the run gave up on it: do not leak it
                lib.gcwt_comm_destroy(h)
        except Exception as e:            # RCCL missing / init failed: fall back to files
            self.rccl_error = "%s: %s" % (type(e).__name__, e)
            self._handle = None

    def _drop_rccl(self, abort=False):
        """destroy: every operation on the communicator has completed (set-up disagreement,
        close).  abort: a collective failed -- a peer may never arrive, so nothing may wait on
        what is in flight (ncclCommAbort, no stream sync, no hipFree)."""
        from ._lib import lib
        h, self._handle = self._handle, None
        (lib.gcwt_comm_abort if abort else lib.gcwt_comm_destroy)(h)

    # -- file backend ---------------------------------------------------------
    def _file_allreduce_max(self, value, tag, negate=False):
        self._seq += 1
        stem = os.path.join(self.dir, "%s_%06d" % (tag, self._seq))
        _publish("%s.%d" % (stem, self.rank), repr(float(value)).encode())
        timeout = getattr(self, "_fallback_timeout", None) or self.timeout
        self._fallback_timeout = None                # (after an RCCL enqueue failure: the first operation only)
        vals = [float(_wait_for("%s.%d" % (stem, r), timeout)) for r in range(self.world)]
        return min(vals) if negate else max(vals)

    # -- interface ------------------------------------------------------------
    def _rccl_failed(self, what, err):
        """An RCCL-backed call failed after set-up.  Three cases, told apart by the library's code:
        * GCWT_ERR_COMM -- the collective could not be ENQUEUED (bad communicator, library state): the same on
          every rank, so the operation is served from the file backend instead; the first fall-back operation
          runs with a short time-out, so that a rank whose peers did not fail exits instead of waiting;
        * GCWT_ERR_COMM_INCOMPLETE -- it was enqueued and did not complete: a peer is gone.  The communicator
          is aborted (never destroyed: that would wait for the peer) and the error goes to the caller, whose
          process ends non-zero so that the launcher stops the rest;
        * anything else (a local HIP error or a refused argument from the plan's upload) is this rank's own
          failure: re-raised as it is, after the communicator is aborted so that nothing waits on it."""
        from ._lib import ERR_COMM, ERR_COMM_INCOMPLETE
        self.rccl_error = "%s: %s" % (what, err)
        code = getattr(err, "code", None)
        try:
            self._drop_rccl(abort=True)
        except Exception:
            self._handle = None
        self.backend = "file"
        if code == ERR_COMM:
            self._fallback_timeout = min(self.timeout, 20.0)     # for the first file operation only
            return
        if code == ERR_COMM_INCOMPLETE:
            raise RuntimeError("RCCL %s did not complete (%s): a peer rank is gone" % (what, err))
        raise err

    def barrier(self):
        if self.world == 1:
            return
        if self._handle:
            from ._lib import lib, check
            try:
                check(lib.gcwt_comm_barrier(self._handle))
                return
            except Exception as e:
                self._rccl_failed("barrier", e)
        self._file_allreduce_max(0.0, "barrier")

    def allreduce_max(self, value):
        if self.world == 1:
            return float(value)
        if self._handle:
            from ._lib import lib, check
            v = C.c_double(float(value))
            try:
                check(lib.gcwt_comm_allreduce_max(self._handle, C.byref(v)))
                return v.value
            except Exception as e:
                self._rccl_failed("allreduce_max", e)
        return self._file_allreduce_max(value, "max")

    def allgather(self, value):
        """[value of rank 0, ..., value of rank world-1] on every rank (world max-reduces: the
        control plane has no other collective and needs none)."""
        if self.world == 1:
            return [float(value)]
        return [self.allreduce_max(float(value) if r == self.rank else float("-inf"))
                for r in range(self.world)]

    def broadcast_bank(self, plan, root=0):
        """RCCL broadcast of rank ``root``'s filter bank into every rank's plan.
        Returns how the bank got there: 'rccl_broadcast' or 'local_build'."""
        plan.upload()                     # every rank builds its bank with the HIP kernel
        if self._handle:
            from ._lib import lib, check
            try:
                check(lib.gcwt_comm_broadcast_bank(self._handle, plan._handle, root))
                return "rccl_broadcast"
            except Exception as e:        # every rank has built the same bank itself already
                self._rccl_failed("broadcast_bank", e)
        return "local_build"

    def close(self):
        if self._handle:
            self._drop_rccl()
        if self.world > 1 and self.backend != "closed":
            self.backend = "closed"
            # leave no rendezvous files behind: every rank signs off, rank 0 removes the
            # directory once all have (nobody reads from it after signing off)
            try:
                _publish(os.path.join(self.dir, "done.%d" % self.rank), b"1")
                if self.rank == 0:
                    import shutil
                    for r in range(self.world):
                        _wait_for(os.path.join(self.dir, "done.%d" % r), 10.0)
                    shutil.rmtree(self.dir, ignore_errors=True)
            except (OSError, TimeoutError):
                pass
