"""End-to-end time of the public API on BASELINE config 2 (1 ch x 1e6 x ~100 scales):
what a user of ContinuousWaveletTransform.transform() sees, host arrays in and out."""
import sys, time, os; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.wave import ContinuousWaveletTransform
from ghost_amd.synthetic import lfp_channel
fs = 1000.0; N = int(os.environ.get("API_N", "1000000"))
x = lfp_channel(N, fs, 0).astype(np.float64)
t = np.arange(N) / fs
cwt = ContinuousWaveletTransform()
for it in range(4):
    t0 = time.perf_counter()
    cwt.transform(x, fs=fs, timestamps=t, freq_limits=[2, 200], voices_per_octave=16,
                  dtype=np.float32 if os.environ.get("API_F32") else None)
    dt = time.perf_counter() - t0
    print("run %d: %.3f s  scales %d  amplitude %s %s -> %.2f Msamples/s" %
          (it, dt, cwt.frequencies.size, cwt.amplitude.shape, cwt.amplitude.dtype, N / dt / 1e6), flush=True)
p = cwt._plan
p.set_profiling(True)
x32 = x.astype(np.float32)[None]
for it in range(3):
    t0 = time.perf_counter(); r = p.execute(x32); dt = time.perf_counter() - t0
    print("plan.execute host->host %.3f s (device stages %.2f ms)" % (dt, p.timings()["total_ms"]), flush=True)
t0 = time.perf_counter(); r64 = r.astype(np.float64); print("astype f64 %.3f s" % (time.perf_counter() - t0))
