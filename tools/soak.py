"""Randomised soak of the whole path against the oracle: many layouts (channels, lengths,
epochs with gaps, sampling rates, frequency ranges down to large decimations, forced time
blocks, output modes, block requests).  Prints the worst relative error per case; exits
non-zero on the first case over the 1e-5 gate.  SOAK_N cases (default 60), SOAK_SEED;
SOAK_GRAPH=1: small cases also run device-resident four times (graph replay) and must
reproduce the host result bit for bit; SOAK_EXACT=1: most cases with precision = 'exact'; SOAK_BIG=1 adds recordings of up to 2.5 M samples (FFT lengths up to 2^22); SOAK_DETAIL=3e-6 prints the
per-scale errors of every case above that."""
import os, sys, time; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan
from oracle import ghost_oracle as orc

TOL = 1e-5
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "7")))
n_cases = int(os.environ.get("SOAK_N", "60"))
worst = 0.0
t_start = time.time()
for case in range(n_cases):
    fs = float(rng.choice([200.0, 1000.0, 1250.0, 30000.0]))
    n_ch = int(rng.integers(1, 4))
    sizes = [17, 500, 4096, 4097, 10000, 33333, 70000, 150000]
    if os.environ.get("SOAK_BIG"):             # long recordings too: FFT lengths up to 2^22, time blocks
        sizes += [400000, 1100000, 2500000]
    n = int(rng.choice(sizes))
    long_mode = bool(os.environ.get("SOAK_LONG"))    # kernels of millions of taps: FFT lengths 2^23 / 2^24 (long mode)
    if long_mode:
        fs, n_ch, n = 30000.0, 1, int(rng.choice([5000000, 9000000, 14000000]))
    k = int(rng.integers(0, 7))
    cuts = np.sort(rng.choice(np.arange(1, n), size=min(n - 1, k), replace=False)) if n > 8 else np.array([], int)
    edges = [0, *cuts.tolist(), n]
    eb = [[a + int(rng.integers(0, 3)), b] for a, b in zip(edges[:-1], edges[1:]) if b - a > 3 and rng.random() < 0.85]
    if not eb:
        eb = [[0, n]]
    shortest = min(b - a for a, b in eb)
    lo = max(1e-4 * fs, 10.0 * fs / max(shortest, 20))
    hi = 0.47 * fs
    ns = int(rng.integers(1, 9))
    f = np.sort(np.exp(rng.uniform(np.log(lo), np.log(hi), ns)))[::-1] if lo < hi else np.array([0.4 * fs])
    if long_mode:                                     # one epoch, a scale near the reference's floor, the rest at R >= 8
        eb = [[0, n]]
        floor_hz = float(orc.rad_to_hz(orc.morse_freq_bounds(n)[0], fs))
        f = np.sort(np.concatenate([[floor_hz * rng.uniform(1.001, 1.3)], np.exp(rng.uniform(np.log(0.5), np.log(900.0), 2))]))[::-1]
    output = ["complex", "amplitude", "power"][int(rng.integers(0, 3))]
    kw = dict(epoch_bounds=eb, output=output)
    gamma, beta = 3.0, 20.0
    if rng.random() < 0.5 and not long_mode:       # other Morse wavelets: light tails (fast path) and heavy ones
        gamma = float(rng.choice([1.0, 2.0, 3.0, 4.0, 6.0]))
        beta = float(np.round(np.exp(rng.uniform(np.log(1.5), np.log(80.0))), 1))
        kw.update(gamma=gamma, beta=beta)
    x = rng.standard_normal((n_ch, n)) * rng.uniform(0.1, 50)
    # round 4: half of the cases are recordings with steep spectra, mains interference or drift (white noise is the
    # easy case for float32 transforms: ghost_amd/synthetic.py: SPECTRUM_CLASSES), kept inside the measured envelope
    # (profiles/r04_dynamic_range.md: below the bands D ~ 1000, inside a level's band D ~ 65; a narrow scale holds
    # 0.05 - 0.3 of a white recording's std, so the line is 1 - 3 x and the drift 10 - 40 x the std; SOAK_WILD=1: 10 x those)
    kind = "white"
    # wavelets other than the default one mostly get no low cut (their kernels answer above 2e-8 down to zero frequency):
    # float32's envelope, so their backgrounds are pink .. 1/f^1.5 instead of 1/f^2 .. 1/f^3
    tilt = 1.0 if (gamma, beta) == (3.0, 20.0) else 0.5
    wild = 10.0 if os.environ.get("SOAK_WILD") else 1.0
    if rng.random() < 0.5 and n >= 500:
        kind = str(rng.choice(["brown", "f3", "line", "drift"]))
        t = np.arange(n) / fs
        for c in range(n_ch):
            if kind in ("brown", "f3"):
                spec = np.fft.rfft(x[c])
                kk = np.maximum(np.arange(spec.size, dtype=np.float64), 1.0)
                x[c] = np.fft.irfft(spec / kk ** ((1.0 if kind == "brown" else 1.5) * tilt), n=n)
            elif kind == "line":
                x[c] += wild * rng.uniform(1, 3) * x[c].std() * np.sin(2 * np.pi * rng.uniform(0.01, 0.4) * fs * t + c)
            else:
                x[c] += wild * rng.uniform(10, 40) * x[c].std() * np.sin(2 * np.pi * rng.uniform(0.1, 3.0) * t / (n / fs) + c)
    x = (x + rng.uniform(-100, 100, (n_ch, 1)) * x.std()).astype(np.float32)
    if rng.random() < 0.35 and not long_mode:
        kw["max_fft_log2"] = int(rng.choice([12, 13, 14, 16] + ([20, 21, 22] if os.environ.get("SOAK_BIG") else [])))
    if long_mode and rng.random() < 0.5:
        kw["max_fft_log2"] = 24
    if os.environ.get("SOAK_EXACT") and not long_mode and rng.random() < 0.6:     # precision = exact: no decimated path
        kw["precision"] = "exact"
    try:
        p = CwtPlan(n, n_ch, fs, f, **kw)
    except Exception as e:
        print("case %2d: plan refused: %s" % (case, str(e)[:90]), flush=True)
        continue
    ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f, np.array(eb), gamma=gamma, beta=beta)
                    for c in range(n_ch)])
    if output == "amplitude":
        ref = np.abs(ref)
    elif output == "power":
        ref = np.abs(ref) ** 2
    got = p.execute(x)
    scale = np.abs(ref).max(axis=2, keepdims=True)
    scale[scale == 0] = 1.0
    err = float((np.abs(got - ref) / scale).max())
    a = int(rng.integers(0, n)); ln = int(rng.integers(1, n - a + 1))
    blk = p.execute_block(x, a, ln)
    same = np.array_equal(blk, got[:, :, a:a + ln])
    si = p.scale_info()
    if os.environ.get("SOAK_GRAPH") and got.size <= 1 << 24 and p.info["n_fullband"] == 0:
        # small plans replay a HIP graph from the third device-resident execute on: the same numbers, bit for bit
        from ghost_amd.engine import DeviceBuffer
        xb, ob = DeviceBuffer(x.nbytes), DeviceBuffer(p.info["out_bytes"])
        xb.upload(x)
        for _ in range(4):
            p.execute_device(xb, ob)
        dev = ob.download(got.shape, got.dtype)
        same = same and np.array_equal(dev, got) and p.debug_graph_state() == 1
        xb.free(); ob.free()
    tol = 2 * TOL if output == "power" else TOL
    print("case %2d: %-5s fs %7.0f ch %d n %6d ep %d g,b %g,%4g scales %d R<=%5d direct %d blocks %d full %d segs %3d %-9s err %.2e block %s" %
          (case, kind, fs, n_ch, n, len(eb), gamma, beta, f.size, si["decimation"].max(), int((si["method"] == 1).sum()),
           int((si["method"] == 3).sum()), int((si["method"] == 2).sum()), len(p.segments()), output, err, "ok" if same else "DIFFERS"), flush=True)
    if os.environ.get("SOAK_DETAIL") and err / (tol / TOL) > float(os.environ["SOAK_DETAIL"]):   # per scale: where the error sits
        e_s = (np.abs(got - ref) / scale).max(axis=(0, 2))
        print("   per scale: " + ", ".join("%.4g Hz L %d R %d m %d %.1e" % (f[i], si["length"][i], si["decimation"][i], si["method"][i], e_s[i])
                                           for i in range(f.size)), flush=True)
        print("   epochs", eb, "kw", {k: v for k, v in kw.items() if k != "epoch_bounds"}, flush=True)
        if kw.get("precision") in (None, "auto", "high"):
            rep = p.precision_report()
            print("   predicted " + ", ".join("%.1e" % v for v in rep["predicted"]) + "  rerouted %d" % rep["rerouted"], flush=True)
    worst = max(worst, err / (tol / TOL))
    if err > tol or not same:
        print("FAILED", dict(fs=fs, n=n, eb=eb, f=f.tolist(), kw=kw, block=(a, ln), gamma_beta=(gamma, beta)))
        sys.exit(1)
    p.close()
print("worst %.2e over %d cases in %.0f s" % (worst, n_cases, time.time() - t_start))
