"""Several GPUs behind one ``transform()``: channels sharded over the devices of one node.

The reference's only parallelism is a ``ThreadPool`` over the scales of ONE channel
(ghost/wave/transforms.py:206-218); channels are independent (``:57-58``).  Here a multichannel
recording is cut into contiguous channel blocks (``ghost_amd.dist.shard_channels``: the split
BASELINE.json names for config 4, 1024 channels over 8 MI355X), one ``CwtPlan`` and one host
thread per device slot -- ctypes releases the GIL for the length of a ``gcwt_execute``, so the
slots upload, compute and drain side by side -- and no collective: every plan builds the same
filter bank from the same frequencies (205 KB; ``bench.py``'s one-process-per-GPU launch is the
path that broadcasts it over RCCL instead).  Results stay on their devices as a list of
``DeviceResult``; ``ShardedResult.to_host`` stitches whatever rectangle is asked for.

``devices=[0, 0]`` is legal: two slots on one GPU (how this is tested on one-GPU boxes) compute
exactly the rows one plan over all channels does -- a channel's numbers do not depend on which
other channels share its plan (``precision='auto'``: the detector's verdicts are per channel,
but a slot reroutes ALL its channels when half of them or more are flagged, so under heavy in-band
interference a channel's rows may be the exact paths' in one split and the fast path's in another;
both meet the gate).
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from .dist import shard_channels
from .engine import CwtPlan, DeviceResult


def _split(n_channels, n_slots):
    return [shard_channels(n_channels, r, n_slots) for r in range(n_slots)]


class ShardedResult:
    """``DeviceResult``'s surface over several of them: ``shape`` (C, S, N), ``to_host`` of any
    (scale, sample) range, ``free``; ``parts`` = [(first channel, stop channel, DeviceResult)]."""

    def __init__(self, parts, shape, is_complex, pool=None):
        self.parts, self.shape, self.is_complex = parts, tuple(int(v) for v in shape), bool(is_complex)
        self._pool = pool                   # the plan's slot threads: one drain per device, side by side

    @property
    def buffer(self):                       # (DeviceResult's test for "still there")
        return None if any(r.buffer is None for _, _, r in self.parts) else self

    @property
    def nbytes(self):
        return sum(r.nbytes for _, _, r in self.parts)

    def to_host(self, dtype=None, scales=None, start=0, stop=None):
        from . import hostmem
        pieces = [None] * len(self.parts)

        def fetch(i):
            pieces[i] = self.parts[i][2].to_host(dtype, scales, start, stop)

        done = False
        if self._pool is not None and len(self.parts) > 1:
            try:
                list(self._pool.map(fetch, range(len(self.parts))))
                done = True
            except RuntimeError:            # the plan (and its threads) is closed: the results outlive it
                self._pool = None
        if not done:
            for i in range(len(self.parts)):
                fetch(i)
        if len(pieces) == 1:
            return pieces[0]
        shape = (self.shape[0],) + pieces[0].shape[1:]
        out = hostmem.empty(shape, pieces[0].dtype)
        if out is None:
            out = np.empty(shape, dtype=pieces[0].dtype)
        for (c0, c1, _), piece in zip(self.parts, pieces):
            out[c0:c1] = piece
        return out

    def free(self):
        for _, _, r in self.parts:
            r.free()


class ShardedPlan:
    """``CwtPlan``'s surface (what ``ContinuousWaveletTransform`` uses of it) over one plan per device slot."""

    def __init__(self, n_samples, n_channels, fs, freqs_hz, devices, **kw):
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("'devices' must name at least one device")
        if len(devices) > n_channels:
            devices = devices[:n_channels]            # a slot without a channel has nothing to do
        self.devices = devices
        self.blocks = _split(int(n_channels), len(devices))
        self.n_samples, self.n_channels = int(n_samples), int(n_channels)
        self.plans = []
        try:
            for (c0, c1), dev in zip(self.blocks, devices):
                self.plans.append(CwtPlan(n_samples, c1 - c0, fs, freqs_hz, device=dev, **kw))
        except Exception:
            self.close()
            raise
        self.n_freqs = self.plans[0].n_freqs
        self.out_dtype = self.plans[0].out_dtype
        self.out_shape = (self.n_channels, self.n_freqs, self.n_samples)
        self.info = dict(self.plans[0].info)
        self.info["workspace_bytes"] = sum(p.info["workspace_bytes"] for p in self.plans)
        self.info["out_bytes"] = sum(p.info["out_bytes"] for p in self.plans)
        self._pool = ThreadPoolExecutor(max_workers=len(self.plans), thread_name_prefix="ghostcwt-dev")
        self._lock = threading.Lock()

    # -- what transform() calls ------------------------------------------------------------
    def set_profiling(self, enabled):
        for p in self.plans:
            p.set_profiling(enabled)

    def execute_resident(self, x, result=None):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(self.n_channels, self.n_samples)
        old = [None] * len(self.plans)
        if isinstance(result, ShardedResult) and len(result.parts) == len(self.plans) and \
                all((a, b) == (c0, c1) for (a, b, _), (c0, c1) in zip(result.parts, self.blocks)):
            old = [r for _, _, r in result.parts]
        elif result is not None:
            result.free()

        def run(i):
            c0, c1 = self.blocks[i]
            # the slot's thread works on the slot's device: the result buffer is allocated there
            if self.devices[i] >= 0:
                _lib.check(_lib.lib.gcwt_set_device(self.devices[i]))
            return self.plans[i].execute_resident(x[c0:c1], old[i])

        with self._lock:
            futures = [self._pool.submit(run, i) for i in range(len(self.plans))]
            parts, first_error = [], None
            for i, fu in enumerate(futures):
                try:
                    parts.append((self.blocks[i][0], self.blocks[i][1], fu.result()))
                except Exception as e:                 # every slot finishes before anything is raised
                    first_error = first_error or e
            if first_error is not None:
                for _, _, r in parts:
                    r.free()
                for r in old:
                    if r is not None:
                        r.free()
                raise first_error
        return ShardedResult(parts, self.out_shape, self.out_dtype == np.complex64, self._pool)

    def precision_report(self):
        reps = [p.precision_report() for p in self.plans]
        pred = np.max([r["predicted"] for r in reps], axis=0)
        n = [r["rerouted"] for r in reps]
        # (a slot that could not reroute reports a negative count: the worst news wins)
        return {"predicted": pred, "worst": float(max(r["worst"] for r in reps)),
                "rerouted": min(n) if min(n) < 0 else max(n), "watched": all(r["watched"] for r in reps),
                "per_device": reps}

    def timings(self):
        ts = [p.timings() for p in self.plans]
        out = {k: max(t[k] for t in ts) for k in ts[0]}        # slots run side by side: the slowest one's stage
        out["per_device"] = ts
        return out

    def close(self):
        for p in getattr(self, "plans", []):
            p.close()
        self.plans = []
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=True)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
