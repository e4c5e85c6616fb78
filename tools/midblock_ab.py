"""Kernels of 1 - 5 K taps under precision = 'exact': the full-band path on the recording's whole grid against time blocks
of 16 384 samples through the fused kernel (k_fullband4; option fullband4 = 0: the same blocks by the two-pass kernels).
Parity of all three against the oracle on a small shape, then times at 128 ch x 1e6."""
import sys, time, os; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer, set_option
from ghost_amd.synthetic import lfp_channel, lfp
from oracle import ghost_oracle as orc
fs = 1000.0
f = np.geomspace(12.0, 2.0, 12)
n = 150000
x = np.stack([lfp_channel(n, fs, 40 + c) for c in range(2)]).astype(np.float32)
ref = np.stack([orc.cwt_amplitude(x[c].astype(np.float64), fs, f, n_threads=8) for c in range(2)])
def rel(a, b): return float((np.abs(a - b).max(axis=-1) / np.abs(b).max(axis=-1)).max())
for tag, kw, opt in (("whole grid", {}, 1), ("2^14 blocks, fused", {"max_fft_log2": 14}, 1), ("2^14 blocks, two-pass", {"max_fft_log2": 14}, 0)):
    set_option("fullband4", opt)
    p = CwtPlan(n, 2, fs, f, precision="exact", **kw)
    got = p.execute(x)
    info = p.info
    print("%-24s err %.2e  segments %d fullband %d lengths %s" % (tag, rel(got, ref), info.get("n_segments", -1), info["n_fullband"], p.scale_info()["length"][[0, -1]]))
    if tag.endswith("fused"): fused = got
    if tag.endswith("two-pass"): print("   fused vs two-pass: max rel diff %.2e" % rel(fused, got))
    p.close()
C, N = int(os.environ.get("MB_C", "128")), 1000000
xs = lfp(4, N); xs = np.tile(xs, (C // 4 + 1, 1))[:C]
xb = DeviceBuffer(xs.nbytes); xb.upload(xs)
for tag, kw, opt in (("whole grid", {}, 1), ("2^14 blocks, fused", {"max_fft_log2": 14}, 1), ("2^14 blocks, two-pass", {"max_fft_log2": 14}, 0)):
    set_option("fullband4", opt)
    p = CwtPlan(N, C, fs, f, precision="exact", **kw)
    ob = DeviceBuffer(p.info["out_bytes"])
    p.set_profiling(True)
    ts = []
    for it in range(4):
        t0 = time.perf_counter(); p.execute_device(xb, ob); ts.append(time.perf_counter() - t0)
    tm = p.timings()
    print("%-24s %d scales: execute %.2f ms (fullband %.2f, fwd %.2f) = %.3f ms per scale" % (tag, f.size, 1e3 * min(ts[1:]), tm["fullband_ms"], tm["fwd_fft_ms"], tm["fullband_ms"] / f.size))
    ob.free(); p.close()
