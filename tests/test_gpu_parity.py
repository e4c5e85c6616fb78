"""Parity of the HIP engine, through the C ABI / class surface, against the
golden vectors made from the reference and against the oracle (GPU).

Gate (BASELINE.json): max_n |y - ref| / max_n |ref| <= 1e-5 per (channel, scale)
on the complex coefficients, and the same on the amplitude."""
import numpy as np
import pytest

from conftest import rel_err
from oracle import ghost_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _cwt(x, fs, **kw):
    from ghost_amd.wave import ContinuousWaveletTransform
    cwt = ContinuousWaveletTransform()
    cwt.transform(x, fs=fs, **kw)
    return cwt


def _plan(x, fs, freqs, **kw):
    from ghost_amd.engine import CwtPlan
    x = np.atleast_2d(np.asarray(x, dtype=np.float32))
    p = CwtPlan(x.shape[1], x.shape[0], fs, freqs, **kw)
    return p, p.execute(x)


def test_config1_public_api(golden):
    """BASELINE config 1: 1 ch x 16384 @ 1 kHz, 32 scales, via transform()."""
    g = golden("g1_config1.npz")
    cwt = _cwt(g["x"], 1000.0, freq_limits=[5, 200], voices_per_octave=6)
    np.testing.assert_allclose(cwt.frequencies, g["frequencies"], rtol=1e-14)
    amp = cwt.amplitude
    assert amp.shape == (32, 16384) and amp.dtype == np.float64
    assert rel_err(amp[:, g["cols"]], g["amplitude_cols"]).max() < TOL
    np.testing.assert_allclose(amp.max(axis=1), g["amplitude_rowmax"], rtol=1e-5)
    assert amp.sum() == pytest.approx(float(g["amplitude_sum"]), rel=1e-6)
    np.testing.assert_allclose(cwt.power, np.square(amp))
    np.testing.assert_allclose(cwt.time, np.arange(16384) / 1000.0)


def test_results_stay_on_the_device_until_asked_for(golden):
    """transform() leaves the rows on the device; `amplitude` / `power` / `coefficients` bring them over on first
    access (reference semantics: whole float64 host arrays, transforms.py:203-204, 496-527), `fetch()` any (scale,
    sample) range without the rest, `lazy=False` restores results-on-return.  Page-locked destinations (the pool of
    ghost_amd.hostmem) and pageable ones (pool limit 0) give the same bits, float32 and widened float64."""
    from ghost_amd import hostmem
    from ghost_amd.wave import ContinuousWaveletTransform
    g = golden("g1_config1.npz")
    x, fs = g["x"], 1000.0
    eager = _cwt(x, fs, freq_limits=[5, 200], voices_per_octave=6, lazy=False)
    assert eager._amplitude is not None and eager._pending is None
    ref = eager.amplitude
    assert rel_err(ref[:, g["cols"]], g["amplitude_cols"]).max() < TOL
    lazy = _cwt(x, fs, freq_limits=[5, 200], voices_per_octave=6)
    assert lazy._amplitude is None and lazy.device_result.shape == (1, 32, 16384)
    # a slice straight from the device: scales 3..9, an unaligned sample range; float64 and float32
    piece = lazy.fetch(scales=slice(3, 10), start=1001, stop=7778)
    assert piece.dtype == np.float64 and lazy._amplitude is None
    np.testing.assert_array_equal(piece, ref[3:10, 1001:7778])
    np.testing.assert_array_equal(lazy.fetch(slice(31, 32), 0, 5, dtype=np.float32), ref[31:32, :5].astype(np.float32))
    assert hostmem.is_pinned(piece)
    amp = lazy.amplitude
    assert amp.dtype == np.float64 and amp.shape == (32, 16384) and hostmem.is_pinned(amp)
    np.testing.assert_array_equal(amp, ref)
    np.testing.assert_array_equal(lazy.fetch(start=16000), ref[:, 16000:])       # after the whole came over, too
    assert lazy.amplitude is amp                                                   # brought over once
    # a second transform on the same object reuses the device buffer and replaces the attributes
    buf = lazy.device_result.buffer.ptr.value
    lazy.transform(x[::-1].copy(), fs=fs, freq_limits=[5, 200], voices_per_octave=6, dtype=np.float32)
    assert lazy.device_result.buffer.ptr.value == buf and lazy._amplitude is None
    assert lazy.amplitude.dtype == np.float32 and not np.array_equal(lazy.amplitude, amp.astype(np.float32))
    np.testing.assert_array_equal(amp, ref)                                        # the first result is the caller's
    # pageable destinations (nothing pinned handed out) give the same numbers; multichannel, power and complex
    old_limit, hostmem.limit_bytes = hostmem.limit_bytes, 0
    try:
        xs = np.stack([x, x[::-1]])
        for out, dt in (("power", np.float64), ("complex", np.float64), ("amplitude", np.float32)):
            a = ContinuousWaveletTransform(); a.transform(xs, fs=fs, freq_limits=[8, 200], voices_per_octave=4,
                                                            multichannel=True, output=out, dtype=dt)
            unpinned = a.coefficients if out == "complex" else getattr(a, out)
            part = a.fetch(slice(2, 5), 100, 9000)
            assert not hostmem.is_pinned(unpinned)
            hostmem.limit_bytes = old_limit
            b = ContinuousWaveletTransform(); b.transform(xs, fs=fs, freq_limits=[8, 200], voices_per_octave=4,
                                                            multichannel=True, output=out, dtype=dt)
            pinned = b.coefficients if out == "complex" else getattr(b, out)
            hostmem.limit_bytes = 0
            assert hostmem.is_pinned(pinned) and pinned.shape == (2, len(a.frequencies), 16384)
            np.testing.assert_array_equal(unpinned, pinned)
            np.testing.assert_array_equal(part, pinned[:, 2:5, 100:9000])
            assert unpinned.dtype == (np.complex128 if out == "complex" else dt)
    finally:
        hostmem.limit_bytes = old_limit
    lazy.release_device()
    assert lazy.device_result is None and lazy.amplitude is not None
    with pytest.raises(ValueError):
        lazy.fetch()
    hostmem.trim()


def test_two_tone_known_answers(golden):
    g = golden("g4b_two_tone.npz")
    fs, n = 1000.0, 4096
    t = np.arange(n) / fs
    x = np.sin(2 * np.pi * 50 * t) + 0.5 * np.sin(2 * np.pi * 12 * t)
    cwt = _cwt(x, fs, freq_limits=[10, 100], voices_per_octave=4)
    np.testing.assert_allclose(cwt.amplitude[:, 2048], g["amplitude_col2048"], rtol=0, atol=2e-6)
    assert cwt.amplitude.sum() == pytest.approx(float(g["amplitude_sum"]), rel=1e-6)
    p, c = _plan(x, fs, [50.0, 12.0], output="complex")
    idx = g["w_idx"]
    np.testing.assert_allclose(c[0, 0, idx], g["w50"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(c[0, 1, idx], g["w12"], rtol=0, atol=2e-6)


def test_small_complex_odd_and_even_lengths(golden):
    g = golden("g2_complex_small.npz")
    p, c = _plan(g["x"], float(g["fs"]), g["frequencies"], output="complex")
    assert rel_err(c[0], g["coeffs"]).max() < TOL
    p, a = _plan(g["x"], float(g["fs"]), g["frequencies"], output="amplitude")
    assert rel_err(a[0], np.abs(g["coeffs"])).max() < TOL
    p, pw = _plan(g["x"], float(g["fs"]), g["frequencies"], output="power")
    assert rel_err(pw[0], np.abs(g["coeffs"]) ** 2).max() < 2 * TOL


def test_two_epochs(golden):
    """Epoch gap: independent zero-padded segments, global mean (transforms.py:143, :202)."""
    g = golden("g5_two_epochs.npz")
    cwt = _cwt(g["x"], float(g["fs"]), timestamps=g["timestamps"], output="complex")
    np.testing.assert_allclose(cwt.frequencies, g["frequencies"], rtol=1e-14)
    c = cwt.coefficients
    assert rel_err(c[:, g["cols"]], g["complex_cols"]).max() < TOL
    np.testing.assert_allclose(cwt.amplitude.max(axis=1), g["amplitude_rowmax"], rtol=1e-5)


def test_near_nyquist_direct_path(golden):
    """Scales whose response reaches Nyquist use the literal kernel; 280 Hz,
    whose exact response is below 2e-7 of its peak from 0.96 pi on, fits the R = 2 band."""
    g = golden("g6_near_nyquist.npz")
    p, c = _plan(g["x"], float(g["fs"]), g["frequencies"], output="complex")
    assert p.scale_info()["method"].tolist() == [1, 1, 1, 1, 0]
    assert rel_err(c[0], g["coeffs"]).max() < TOL


def test_default_grid_mixes_direct_and_spectral(golden):
    g = golden("g1_config1.npz")
    x = g["x"]
    f = orc.frequency_grid(1000.0, x.size)
    ref = orc.cwt_complex(x.astype(np.float64), 1000.0, f[::5])
    p, c = _plan(x, 1000.0, f[::5], output="complex")
    m = p.scale_info()["method"]
    assert m.min() == 0 and m.max() == 1
    assert rel_err(c[0], ref).max() < TOL


def test_multichannel_equals_per_channel_reference(golden):
    g = golden("g8_multichannel.npz")
    from ghost_amd.wave import ContinuousWaveletTransform
    cwt = ContinuousWaveletTransform()
    cwt.transform(g["x"], fs=1000.0, freq_limits=[20, 250], voices_per_octave=4,
                  multichannel=True, dtype=np.float32)
    np.testing.assert_allclose(cwt.frequencies, g["frequencies"], rtol=1e-14)
    amp = cwt.amplitude
    assert amp.shape == g["amplitude"].shape and amp.dtype == np.float32
    for c in range(3):
        assert rel_err(amp[c], g["amplitude"][c]).max() < TOL


def test_config2_reduced(golden):
    """BASELINE config 2 scale set (100 log-spaced 200..2 Hz), N = 65536."""
    g = golden("g9_config2_reduced.npz")
    p, c = _plan(g["x"], float(g["fs"]), g["frequencies"], output="complex")
    err = rel_err(c[0][:, g["cols"]], g["complex_cols"].astype(np.complex128))
    assert err.max() < TOL, err
    np.testing.assert_allclose(np.abs(c[0]).max(axis=1), g["amplitude_rowmax"], rtol=2e-5)
    # amplitude / power: k_synth7 on the levels R = 2, 4, 8, the interpolating kernel k_synthi from R = 16 up
    # (the scales that reach R = 256 are folded into the R = 128 level)
    p, a = _plan(g["x"], float(g["fs"]), g["frequencies"], output="amplitude")
    assert sorted(set(p.scale_info()["decimation"])) == [2, 4, 8, 16, 32, 64, 128]
    ref = np.abs(g["complex_cols"].astype(np.complex128))
    assert rel_err(a[0][:, g["cols"]], ref).max() < TOL
    p, pw = _plan(g["x"], float(g["fs"]), g["frequencies"], output="power")
    assert rel_err(pw[0][:, g["cols"]], ref ** 2).max() < 2 * TOL


def test_edge_shapes_against_oracle():
    """Odd lengths, tiny epochs, length-1-mod-hop sizes, constant and zero input."""
    from ghost_amd.synthetic import lfp_channel
    fs = 1000.0
    for n, f in [(257, [100.0, 60.0]), (4097, [200.0, 33.3, 9.0]), (12345, [150.0, 7.5]),
                 (70, [200.0])]:
        x = lfp_channel(n, fs, channel=n % 7)
        ref = orc.cwt_complex(x.astype(np.float64), fs, f)
        p, c = _plan(x, fs, f, output="complex")
        assert rel_err(c[0], ref).max() < TOL, (n, f)
        p, a = _plan(x, fs, f, output="amplitude")
        assert rel_err(a[0], np.abs(ref)).max() < TOL, (n, f)
    p, a = _plan(np.full(3000, 3.25, np.float32), fs, [50.0, 10.0])
    assert np.abs(a).max() < 1e-5          # constant input: mean removal leaves nothing
    p, a = _plan(np.zeros(3000, np.float32), fs, [50.0, 10.0])
    assert np.all(a == 0)


def test_seeded_random_layouts_against_oracle():
    """Twelve seeded random problems: channel counts, lengths (down to a handful of
    samples), several epochs with gaps, sampling rates, mixed direct/spectral scales, every
    output mode, forced time blocks."""
    rng = np.random.default_rng(20261003)
    ran = 0
    for case in range(12):
        fs = float(rng.choice([250.0, 1000.0, 2000.0, 30000.0]))
        n_ch = int(rng.integers(1, 5))
        n = int(rng.choice([9, 64, 300, 1000, 2049, 7777, 20000, 33000]))
        x = rng.standard_normal((n_ch, n)).astype(np.float32)
        x += rng.uniform(-3, 3, (n_ch, 1)).astype(np.float32)
        # epochs: 1-3 pieces covering parts of [0, n)
        cuts = np.sort(rng.choice(np.arange(1, n), size=min(n - 1, int(rng.integers(0, 5))),
                                  replace=False)) if n > 4 else np.array([], int)
        edges = [0, *cuts.tolist(), n]
        eb = [[a, b] for a, b in zip(edges[:-1], edges[1:]) if rng.random() < 0.8 or b - a == n]
        if not eb:
            eb = [[0, n]]
        # frequencies between ~40 cycles per shortest epoch (or 0.002 fs) and 0.45 fs
        shortest = min(b - a for a, b in eb)
        lo = max(0.002 * fs, 12.0 * fs / max(shortest, 24))
        f = np.sort(np.exp(rng.uniform(np.log(lo), np.log(0.45 * fs), int(rng.integers(1, 7)))))[::-1] \
            if lo < 0.45 * fs else np.array([0.4 * fs])
        output = ["complex", "amplitude", "power"][case % 3]
        kw = dict(epoch_bounds=eb, output=output)
        if case % 4 == 3 and n >= 20000:
            kw["max_fft_log2"] = 13
        ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f, np.array(eb))
                        for c in range(n_ch)])
        if output == "amplitude":
            ref = np.abs(ref)
        elif output == "power":
            ref = np.abs(ref) ** 2
        try:
            p, got = _plan(x, fs, f, **kw)
        except Exception as e:           # a layout the planner refuses must say so, not crash
            assert "UNSUPPORTED" in repr(e) or getattr(e, "code", 0) == -2, (case, repr(e))
            continue
        scale = np.abs(ref).max(axis=2, keepdims=True)
        scale[scale == 0] = 1.0
        err = (np.abs(got - ref) / scale).max()
        assert err < (2 * TOL if output == "power" else TOL), (case, fs, n, eb, f, output, err)
        ran += 1
    assert ran >= 10, "only %d of 12 random layouts were accepted by the planner" % ran


def test_batched_epochs_against_oracle():
    """40 epochs of uneven length with unused gaps between some of them: launched 16 at a
    time as extra 'channels'.  Every epoch must equal its own 'same' convolution, and a
    block request that touches only part of a batch must equal the slice."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd.synthetic import lfp
    fs, n = 1000.0, 120000
    x = lfp(2, n, fs) + np.array([[0.4], [-1.1]], np.float32)
    eb = [[i * 3000 + (i % 3) * 11, i * 3000 + 2000 + 37 * (i % 5)] for i in range(40)]
    f = [320.0, 140.0, 61.0, 33.0]                 # one direct scale, three decimation levels
    ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f, np.array(eb)) for c in range(2)])
    p = CwtPlan(n, 2, fs, f, epoch_bounds=eb, output="complex")
    assert p.debug_batches() == [(0, 16), (16, 16), (32, 8)]
    got = p.execute(x)
    scale = np.abs(ref).max(axis=2, keepdims=True)
    assert (np.abs(got - ref) / scale).max() < TOL
    inside = np.zeros(n, bool)
    for a, b in eb:
        inside[a:b] = True
    assert np.all(got[:, :, ~inside] == 0)         # samples of no epoch (transforms.py:185)
    # range covering the tail of one batch and the head of the next, cutting epochs in half
    blk = p.execute_block(x, 40000, 23456)
    np.testing.assert_array_equal(blk, got[:, :, 40000:63456])
    pa = CwtPlan(n, 2, fs, f, epoch_bounds=eb, output="amplitude")
    assert (np.abs(pa.execute(x) - np.abs(ref)) / scale).max() < TOL


def test_integration_stub_from_the_docs_runs():
    """INTEGRATION.md shows the ctypes stub a ghost maintainer would add; run exactly that
    text against the built library."""
    import os
    import re
    from conftest import ROOT
    from ghost_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C, numpy as np.*?)```", text, re.S).group(1)
    code = code.replace('C.CDLL("libghostcwt.so")', "C.CDLL(%r)" % _lib.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    from ghost_amd.synthetic import lfp_channel
    fs, n = 1000.0, 8192
    x = lfp_channel(n, fs, 3)
    f = np.array([150.0, 40.0, 12.0])
    amp = ns["cwt_amplitude"](x, fs, f, [[0, n]])
    assert amp.dtype == np.float64 and amp.shape == (3, n)
    ref = np.abs(orc.cwt_complex(x.astype(np.float64), fs, f))
    assert rel_err(amp, ref).max() < TOL


def test_fallback_synthesis_kernel(option):
    """k_synth (16 columns, staged tile) serves, level by level, the layouts the production
    kernel does not take: block halos above 48 (a long kernel on a short epoch caps the
    decimation) and levels with more than 256 scales; halos of 33..48 run the production
    kernel's deep-halo rows; option synth16 = 1 forces the 16-column kernel everywhere."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd.synthetic import lfp
    fs = 1000.0
    x = lfp(2, 1500, fs)
    for f, lo, hi in (([100.0, 10.0], 33, 48), ([100.0, 6.0], 49, 112)):   # L = 1395 / 2325 on P = 4096: R capped at 16
        p = CwtPlan(1500, 2, fs, f, output="complex")
        assert lo <= p.scale_info()["halo"][1] <= hi and np.all(p.scale_info()["method"] == 0)
        ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f) for c in range(2)])
        assert rel_err(p.execute(x), ref).max() < TOL
        pa = CwtPlan(1500, 2, fs, f, output="amplitude")
        assert rel_err(pa.execute(x), np.abs(ref)).max() < TOL
    # 300 scales inside one octave -> one level with more than 256 scales
    x1 = lfp(1, 6000, fs)
    f2 = np.geomspace(68.0, 38.0, 300)
    p2 = CwtPlan(6000, 1, fs, f2, output="amplitude")
    assert len(set(p2.scale_info()["decimation"].tolist())) == 1
    ref2 = np.abs(orc.cwt_complex(x1[0].astype(np.float64), fs, f2))
    assert rel_err(p2.execute(x1)[0], ref2).max() < TOL
    # forced, on a multi-epoch (batched) layout, every output mode
    option("synth16", 1)
    xs = lfp(2, 20000, fs)
    eb = [[i * 2000, i * 2000 + 1900] for i in range(10)]
    f3 = [150.0, 60.0, 25.0]
    ref3 = np.stack([orc.cwt_complex(xs[c].astype(np.float64), fs, f3, np.array(eb)) for c in range(2)])
    for output, want in (("complex", ref3), ("amplitude", np.abs(ref3)), ("power", np.abs(ref3) ** 2)):
        p3 = CwtPlan(20000, 2, fs, f3, epoch_bounds=eb, output=output)
        sc = np.abs(want).max(axis=2, keepdims=True)
        assert (np.abs(p3.execute(xs) - want) / sc).max() < 2 * TOL, output


def test_sixteen_column_build_of_the_production_kernel(option):
    """Options synth_cols = 16 and slow_fft = 1 select the other instantiations
    (k_synth7<.,16>, radix-2 FFT passes): same gate."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd.synthetic import lfp
    fs, n = 1000.0, 40000
    x = lfp(2, n, fs)
    f = [200.0, 90.0, 30.0, 11.0, 4.0]
    ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f) for c in range(2)])
    option("synth_cols", 16)
    for output, want in (("complex", ref), ("amplitude", np.abs(ref))):
        p = CwtPlan(n, 2, fs, f, output=output)
        assert rel_err(p.execute(x), want).max() < TOL
    option("synth_cols", None)
    option("slow_fft", 1)
    p = CwtPlan(n, 2, fs, f, output="complex")
    assert rel_err(p.execute(x), ref).max() < TOL


def test_device_resident_output_with_row_pitch():
    """Device in, device out (what bench.py times), with rows padded to a pitch so that
    every row starts on a 128-byte boundary although N is odd; the pad is left alone."""
    from ghost_amd.engine import CwtPlan, DeviceBuffer
    from ghost_amd.synthetic import lfp
    fs, n, pitch = 1000.0, 10001, 10016
    x = lfp(2, n, fs)
    f = [180.0, 55.0, 14.0]
    p = CwtPlan(n, 2, fs, f, output="amplitude")
    want = p.execute(x)
    p.set_row_pitch(pitch)
    xb = DeviceBuffer(x.nbytes); xb.upload(x)
    ob = DeviceBuffer(2 * 3 * pitch * 4)
    ob.upload(np.full((2, 3, pitch), -7.0, np.float32))
    p.execute_device(xb, ob)
    got = ob.download((2, 3, pitch), np.float32)
    np.testing.assert_array_equal(got[:, :, :n], want)
    assert np.all(got[:, :, n:] == -7.0)
    p.set_row_pitch(0)                                   # dense rows again
    ob2 = DeviceBuffer(2 * 3 * n * 4)
    p.execute_device(xb, ob2)
    np.testing.assert_array_equal(ob2.download((2, 3, n), np.float32), want)
    with pytest.raises(Exception):
        p.set_row_pitch(100); p.execute_device(xb, ob)   # pitch shorter than the rows


def test_plans_give_their_memory_back():
    """Create / run / close plans of several layouts (host and device results, time blocks,
    the sigtools operators): the device's free memory returns to where it was."""
    from ghost_amd.engine import CwtPlan, DeviceBuffer, device_memory
    from ghost_amd.sigtools import analytic_signal_hip, fastconv_hip
    from ghost_amd.synthetic import lfp
    fs, n = 1000.0, 60000
    x = lfp(4, n, fs)
    warm = CwtPlan(n, 4, fs, [100.0, 10.0]); warm.execute(x); warm.close()
    analytic_signal_hip(x[0][:5001])
    free0, total = device_memory()
    assert total > 200e9                      # 288 GB part
    for it in range(6):
        kw = dict(output=["amplitude", "complex"][it % 2])
        if it % 3 == 2:
            kw["max_fft_log2"] = 13
        p = CwtPlan(n, 4, fs, np.geomspace(300.0, 3.0, 10 + it), **kw)
        p.execute(x, wide=bool(it & 1))
        xb = DeviceBuffer(x.nbytes); xb.upload(x)
        ob = DeviceBuffer(p.info["out_bytes"])
        p.execute_device(xb, ob)
        xb.free(); ob.free(); p.close()
        fastconv_hip(x[0], np.ones(100))
        analytic_signal_hip(x[1][:5001])
    free1, _ = device_memory()
    assert abs(free1 - free0) < 64 << 20, (free0, free1)


def test_gap_between_epochs_is_zero():
    from ghost_amd.synthetic import lfp_channel
    fs = 1000.0
    x = lfp_channel(9000, fs, 2)
    eb = [[0, 4000], [5000, 9000]]                      # samples 4000..4999 belong to no epoch
    p, c = _plan(x, fs, [80.0, 20.0], epoch_bounds=eb, output="complex")
    assert np.all(c[0][:, 4000:5000] == 0)
    ref = orc.cwt_complex(x.astype(np.float64), fs, [80.0, 20.0], np.array(eb))
    # the oracle removes the mean of the whole array, like the reference
    assert rel_err(np.delete(c[0], np.s_[4000:5000], axis=1),
                   np.delete(ref, np.s_[4000:5000], axis=1)).max() < TOL


def test_full_size_properties():
    """BASELINE config 2 at full size (1 x 1e6 x 100 scales): properties that do not
    need the oracle at that size, plus an oracle spot-check on a few scales."""
    from ghost_amd.synthetic import lfp
    fs, n = 1000.0, 1000000
    f = np.geomspace(200.0, 2.0, 100)
    x = lfp(2, n, fs)
    p, c = _plan(x, fs, f, output="complex")
    # linearity: W(a x0 + b x1) = a W(x0) + b W(x1)
    mix = (0.7 * x[0] - 1.3 * x[1]).astype(np.float32)
    pm, cm = _plan(mix, fs, f, output="complex")
    lin = 0.7 * c[0] - 1.3 * c[1]
    assert rel_err(cm[0], lin).max() < 3e-6
    # amplitude / power modes agree with |complex|
    pa, a = _plan(x[:1], fs, f, output="amplitude")
    assert rel_err(a[0], np.abs(c[0])).max() < 1e-6
    # analytic filters: a pure tone at a scale's peak gives amplitude ~1 there, away from edges
    t = np.arange(n) / fs
    tone = np.sin(2 * np.pi * f[40] * t).astype(np.float32)
    pt, at = _plan(tone, fs, f[40:41])
    assert abs(at[0, 0, n // 2] - 1.0) < 1e-4
    # oracle spot check (literal path) on three scales
    sel = [0, 57, 99]
    ref = orc.cwt_complex(x[0].astype(np.float64), fs, f[sel])
    assert rel_err(c[0][sel], ref).max() < TOL


def test_rccl_single_rank_communicator():
    """RCCL entry points on one GPU (world size 1): id, init, max-reduce, bank broadcast."""
    import ctypes as C
    from ghost_amd._lib import lib, check, COMM_ID_BYTES
    from ghost_amd.engine import CwtPlan
    ident = C.create_string_buffer(COMM_ID_BYTES)
    check(lib.gcwt_comm_unique_id(ident))
    comm = C.c_void_p()
    check(lib.gcwt_comm_create(C.byref(comm), 0, 1, ident))
    v = C.c_double(3.5)
    check(lib.gcwt_comm_allreduce_max(comm, C.byref(v)))
    assert v.value == 3.5
    check(lib.gcwt_comm_barrier(comm))
    plan = CwtPlan(8192, 1, 1000.0, [100.0, 20.0])
    before = plan.filter_bank()
    check(lib.gcwt_comm_broadcast_bank(comm, plan._handle, 0))
    np.testing.assert_array_equal(plan.filter_bank(), before)
    lib.gcwt_comm_destroy(comm)


def test_time_blocks_are_seamless(golden):
    """Long epochs are processed in overlapping time blocks (transforms.py:529-597 sketches
    the same both-edges-discarded scheme); the seams must not show."""
    from ghost_amd.synthetic import lfp_channel
    fs, n = 1000.0, 40000
    x = lfp_channel(n, fs, 4)
    f = [200.0, 77.0, 20.0]
    ref = orc.cwt_complex(x.astype(np.float64), fs, f)
    p, c = _plan(x, fs, f, output="complex", max_fft_log2=13)
    assert len(p.segments()) > 4
    assert rel_err(c[0], ref).max() < TOL
    p, a = _plan(x, fs, f, output="amplitude", max_fft_log2=13)
    assert rel_err(a[0], np.abs(ref)).max() < TOL
    # two epochs, the second one cut into blocks; plus a near-Nyquist (direct) scale
    eb = [[0, 5000], [5000, 40000]]
    f2 = [350.0, 120.0, 25.0]
    ref2 = orc.cwt_complex(x.astype(np.float64), fs, f2, np.array(eb))
    p, c2 = _plan(x, fs, f2, output="complex", epoch_bounds=eb, max_fft_log2=13)
    assert p.scale_info()["method"].tolist() == [1, 0, 0]
    assert rel_err(c2[0], ref2).max() < TOL


def test_execute_block_streams_the_same_numbers():
    """gcwt_execute_block: any sample range, from the whole recording, equals the slice of
    the full transform (global mean included)."""
    from ghost_amd.synthetic import lfp
    fs, n = 1000.0, 30000
    x = lfp(2, n, fs) + 0.75                     # non-zero mean: must be the recording's
    f = [300.0, 150.0, 40.0, 12.0]
    p, full = _plan(x, fs, f, output="amplitude", max_fft_log2=13)
    segs = p.segments()
    for (a, b, _) in (segs[0], segs[2], segs[-1]):
        blk = p.execute_block(x, a, b - a)
        np.testing.assert_array_equal(blk, full[:, :, a:b])
    blk = p.execute_block(x, 7777, 9001, reuse_means=True)     # unaligned, spans several blocks
    np.testing.assert_array_equal(blk, full[:, :, 7777:7777 + 9001])
    p2, cfull = _plan(x, fs, f, output="complex")
    np.testing.assert_array_equal(p2.execute_block(x, 100, 5000), cfull[:, :, 100:5100])


def test_large_dc_offset_is_removed_in_double():
    """Raw ADC-like data: offset 20x the signal.  The global mean (transforms.py:142-143) is
    taken out in fp64 before any fp32 arithmetic, so the gate still holds."""
    from ghost_amd.synthetic import lfp_channel
    fs, n = 1000.0, 30000
    x = (300.0 * lfp_channel(n, fs, 2) + 6000.0).astype(np.float32)
    f = [220.0, 90.0, 31.0, 7.5]
    ref = orc.cwt_complex(x.astype(np.float64), fs, f)
    p, c = _plan(x, fs, f, output="complex")
    assert rel_err(c[0], ref).max() < TOL


def test_wide_host_output_is_the_float32_result_widened():
    """GCWT_OUT_F64: the reference's float64/complex128 result, widened while the copy from
    the device is in flight (rows staged through pinned buffers, scattered by threads)."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd.synthetic import lfp
    fs, n = 1000.0, 70001                       # odd length: padded device rows, dense host rows
    x = lfp(3, n, fs)
    f = np.geomspace(200.0, 5.0, 40)            # 120 rows x 280 KB: several staging chunks
    for output, narrow_dt, wide_dt in (("amplitude", np.float32, np.float64),
                                       ("complex", np.complex64, np.complex128)):
        p = CwtPlan(n, 3, fs, f, output=output)
        a = p.execute(x)
        b = p.execute(x, wide=True)
        assert a.dtype == narrow_dt and b.dtype == wide_dt and a.shape == b.shape
        np.testing.assert_array_equal(b, a.astype(wide_dt))
        blk = p.execute_block(x, 1234, 40000, wide=True)
        np.testing.assert_array_equal(blk, b[:, :, 1234:41234])
        p.close()
    # through the public API: float64 amplitude by default, float32 on request
    from ghost_amd.wave import ContinuousWaveletTransform
    cwt = ContinuousWaveletTransform()
    t = np.arange(n) / fs
    cwt.transform(x[0].astype(np.float64), fs=fs, timestamps=t, freq_limits=[5, 200])
    a64 = cwt.amplitude
    cwt.transform(x[0].astype(np.float64), fs=fs, timestamps=t, freq_limits=[5, 200], dtype=np.float32)
    assert a64.dtype == np.float64 and cwt.amplitude.dtype == np.float32
    np.testing.assert_array_equal(a64, cwt.amplitude.astype(np.float64))


def test_time_block_shards_reassemble_the_transform():
    """SURVEY.md 8e, few channels x long recording: ranks take runs of time blocks, read
    the whole recording and stream their part; glued together = the full transform."""
    from ghost_amd.dist import shard_time_blocks
    from ghost_amd.synthetic import lfp
    fs, n = 1000.0, 50000
    x = lfp(1, n, fs) - 0.3
    f = [250.0, 60.0, 9.0]
    p, full = _plan(x, fs, f, output="amplitude", max_fft_log2=13)
    world = 3
    parts = []
    for r in range(world):
        a, b = shard_time_blocks(p.segments(), r, world)
        assert b > a
        parts.append(p.execute_block(x, a, b - a))
    np.testing.assert_array_equal(np.concatenate(parts, axis=2), full)


def test_low_frequencies_at_high_sampling_rate():
    """BASELINE config 5 regime (30 kHz, 1-500 Hz): decimation far beyond 256."""
    from ghost_amd.synthetic import lfp_channel
    fs = 30000.0
    n = 300000
    x = lfp_channel(n, fs, 8)
    f = [500.0, 80.0, 12.0, 6.0]
    p, c = _plan(x, fs, f, output="complex")
    # 6 Hz could run at R = 2048; alone there, it is walked by the R = 1024 workgroups instead
    assert p.scale_info()["decimation"].tolist() == [32, 128, 1024, 1024]
    ref = orc.cwt_complex(x.astype(np.float64), fs, f)
    assert rel_err(c[0], ref).max() < TOL
    # 1 Hz at 30 kHz: 418 430-tap kernel, decimation 8192, in time blocks of 2^21
    n = 1500000
    x = lfp_channel(n, fs, 9)
    p, a = _plan(x, fs, [1.0, 30.0], output="amplitude")
    assert p.scale_info()["length"][0] == 418430 and p.scale_info()["decimation"][0] == 8192
    ref = np.abs(orc.cwt_complex(x.astype(np.float64), fs, [1.0, 30.0]))
    assert rel_err(a[0], ref).max() < TOL


GB_PAIRS = [(3, 8), (3, 4), (3, 2), (2, 8), (4, 30), (1, 5)]


@pytest.mark.parametrize("gamma,beta", GB_PAIRS)
def test_other_morse_parameters_inner_loop(golden, gamma, beta):
    """G11, driver D2: complex coefficients of the reference run with
    Morse(gamma=, beta=) (ghost/wave/morse.py:14-51, morseutils.py:93-151).  Light-tailed
    wavelets stay on the decimated path; heavy-tailed ones (3,4), (3,2), (2,8), (1,5) go
    through the time-domain and full-band paths -- same gate either way."""
    g = golden("g11_gamma_beta.npz")
    tag = "g%d_b%d" % (gamma, beta)
    p, c = _plan(g["x"], float(g["fs"]), g["frequencies"], output="complex", gamma=gamma, beta=beta)
    si = p.scale_info()
    np.testing.assert_array_equal(si["length"], g["lengths_" + tag])
    err = rel_err(c[0][:, g["cols"]], g["complex_cols_" + tag])
    print(tag, "methods", si["method"].tolist(), "err", err)
    assert err.max() < TOL
    p2, a = _plan(g["x"], float(g["fs"]), g["frequencies"], output="amplitude", gamma=gamma, beta=beta)
    assert rel_err(a[0][:, g["cols"]], np.abs(g["complex_cols_" + tag])).max() < TOL
    np.testing.assert_allclose(a[0].max(axis=1), g["rowmax_" + tag], rtol=1e-5)


@pytest.mark.parametrize("gamma,beta", GB_PAIRS)
def test_other_morse_parameters_public_api(golden, gamma, beta):
    """G11, driver D1: ContinuousWaveletTransform(wavelet=Morse(gamma=, beta=))
    (ghost/wave/transforms.py:42-46)."""
    from ghost_amd.wave import ContinuousWaveletTransform, Morse
    g = golden("g11_gamma_beta.npz")
    tag = "g%d_b%d" % (gamma, beta)
    cwt = ContinuousWaveletTransform(wavelet=Morse(gamma=gamma, beta=beta))
    cwt.transform(g["x"], fs=float(g["fs"]), freq_limits=[8, 300], voices_per_octave=4)
    np.testing.assert_allclose(cwt.frequencies, g["d1_frequencies_" + tag], rtol=1e-14)
    assert rel_err(cwt.amplitude[:, g["cols"]], g["d1_amplitude_cols_" + tag]).max() < TOL
    np.testing.assert_allclose(cwt.amplitude.max(axis=1), g["d1_rowmax_" + tag], rtol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_public_api_grid(golden, tag):
    """G13, driver D1: the public call over sampling rates 200 Hz .. 30 kHz, odd lengths, 4 .. 48
    voices per octave, limits the reference clamps (transforms.py:412-434) and timestamp gaps
    that cut the recording into epochs (preprocessing.py:78-114)."""
    g = golden("g13_api_grid.npz")
    cwt = _cwt(g["x_" + tag], float(g["fs_" + tag]), timestamps=g["t_" + tag],
               freq_limits=g["limits_" + tag].tolist(), voices_per_octave=int(g["voices_" + tag]))
    np.testing.assert_allclose(cwt.frequencies, g["frequencies_" + tag], rtol=1e-13)
    assert cwt.amplitude.dtype == np.float64 and cwt.amplitude.shape == (g["frequencies_" + tag].size, g["x_" + tag].size)
    assert rel_err(cwt.amplitude[:, g["cols_" + tag]], g["amplitude_cols_" + tag]).max() < TOL
    np.testing.assert_allclose(cwt.amplitude.max(axis=1), g["rowmax_" + tag], rtol=2e-5)


def test_fullband_path_long_kernels_epochs_and_blocks(option):
    """The full-band path on its own terms: kernels of thousands of taps (small beta, low
    frequencies), two epochs, several channels, forced time blocks, execute_block."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd import _lib
    fs, n = 1000.0, 50000
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((2, n)) + 0.3 * np.cumsum(rng.standard_normal((2, n)), axis=1) * 0.05).astype(np.float32)
    f = np.array([120.0, 31.0, 7.0, 2.5, 1.2])
    eb = np.array([[100, 30000], [30011, 50000]])
    ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f, eb, gamma=3, beta=3) for c in range(2)])
    option("blockconv", 0)               # (kernels up to 2560 taps go by blocks otherwise: the tests below)
    p = CwtPlan(n, 2, fs, f, gamma=3, beta=3, epoch_bounds=eb, output="complex")
    m = p.scale_info()["method"]
    assert (m == _lib.SCALE_FULLBAND).sum() >= 3 and (m == _lib.SCALE_SPECTRAL).sum() == 0
    got = p.execute(x)
    assert rel_err(got, ref).max() < TOL
    assert np.all(got[:, :, :100] == 0) and np.all(got[:, :, 30000:30011] == 0)
    blk = p.execute_block(x, 29000, 3000)
    np.testing.assert_array_equal(blk, got[:, :, 29000:32000])
    # time blocks of 2^14 samples: seams exact
    p2 = CwtPlan(n, 2, fs, f[:4], gamma=3, beta=3, epoch_bounds=eb, output="amplitude", max_fft_log2=14)
    assert len(p2.segments()) > 4
    got2 = p2.execute(x)
    assert rel_err(got2, np.abs(ref[:, :4])).max() < TOL


def test_blockconv_path_lengths_epochs_modes_and_ranges(option):
    """The block convolution (kernels.hip: k_bc_scales; fwd64.hip: k_bc_forward): kernels of 60 .. 2400 taps in
    several groups, three epochs (one shorter than a block, one starting mid-block), the three output modes and
    execute_block, against the oracle's fastconv with the literal kernels (convolution.py:68-87); and the same
    plan on the older paths (time domain / one FFT per segment) agrees."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd import _lib
    fs, n = 1000.0, 60000
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((3, n)) + 0.2 * np.cumsum(rng.standard_normal((3, n)), axis=1) * 0.05 + 40.0).astype(np.float32)
    f = np.array([300.0, 140.0, 61.0, 33.0, 17.0, 9.0, 4.1, 2.7])
    eb = np.array([[70, 21000], [21013, 22500], [30000, 59990]])
    ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f, eb, gamma=3, beta=3) for c in range(3)])
    p = CwtPlan(n, 3, fs, f, gamma=3, beta=3, epoch_bounds=eb, output="complex")
    si = p.scale_info()
    assert (si["method"] == _lib.SCALE_BLOCKCONV).sum() >= 6 and si["length"].max() > 2000
    assert p.info["n_blockconv"] == (si["method"] == _lib.SCALE_BLOCKCONV).sum()
    got = p.execute(x)
    assert rel_err(got, ref).max() < TOL
    assert np.all(got[:, :, :70] == 0) and np.all(got[:, :, 21000:21013] == 0) and np.all(got[:, :, 22500:30000] == 0)
    blk = p.execute_block(x, 20000, 11000)
    np.testing.assert_array_equal(blk, got[:, :, 20000:31000])
    for output, want in (("amplitude", np.abs(ref)), ("power", np.abs(ref) ** 2)):
        q = CwtPlan(n, 3, fs, f, gamma=3, beta=3, epoch_bounds=eb, output=output)
        assert rel_err(q.execute(x), want).max() < (2 * TOL if output == "power" else TOL), output
    option("blockconv", 0)
    old = CwtPlan(n, 3, fs, f, gamma=3, beta=3, epoch_bounds=eb, output="complex")
    assert old.info["n_blockconv"] == 0
    assert rel_err(old.execute(x), got).max() < 3e-6


def test_small_device_resident_executes_replay_a_graph(option):
    """Config 1's shape is launch-bound (fifteen kernels of microseconds): the second device-resident execute with the
    same arguments is captured into a HIP graph (api.cpp: execute_range) and later ones replay it.  The replay must
    read the buffers as they are at the time -- new samples through the same pointers -- serve block requests and two
    epochs, and give what the eager path gives, bit for bit."""
    from ghost_amd.engine import CwtPlan, DeviceBuffer
    from ghost_amd.synthetic import lfp
    fs, n, C = 1000.0, 16384, 2
    f = 200.0 / 2.0 ** (np.arange(32) / 6.0)
    eb = np.array([[0, 9000], [9003, n]])
    xs = [lfp(C, n, fs, seed=s) for s in (1, 2, 3)]
    eager = []
    option("graphs", 0)
    p0 = CwtPlan(n, C, fs, f, epoch_bounds=eb, output="amplitude")
    for x in xs:
        eager.append(p0.execute(x))
    blk0 = p0.execute_block(xs[2], 5000, 7000)
    option("graphs", None)
    p = CwtPlan(n, C, fs, f, epoch_bounds=eb, output="amplitude")
    xb, ob = DeviceBuffer(4 * C * n), DeviceBuffer(p.info["out_bytes"])
    for rep in range(2):
        for x, want in zip(xs, eager):                      # execute 1 eager, 2 captured, 3 .. 6 replayed
            xb.upload(x)
            p.execute_device(xb, ob)
            np.testing.assert_array_equal(ob.download((C, 32, n), np.float32), want)
    assert p.debug_graph_state() == 1 and p0.debug_graph_state() == 0
    ref = np.stack([orc.cwt_amplitude(xs[2][c].astype(np.float64), fs, f, eb) for c in range(C)])
    assert rel_err(ob.download((C, 32, n), np.float32), ref).max() < TOL
    bb = DeviceBuffer(4 * C * 32 * 7000)
    for _ in range(4):                                        # another key: eager, captured, replayed
        p.execute_block_device(xb, bb, 5000, 7000)
        np.testing.assert_array_equal(bb.download((C, 32, 7000), np.float32), blk0)
    p.execute_device(xb, ob)                                  # back to the first key
    np.testing.assert_array_equal(ob.download((C, 32, n), np.float32), eager[2])
    xb.free(); ob.free(); bb.free()


def test_fullband_sets_of_four_with_a_remainder(option):
    """The full-band row pass takes up to four scales per pass over the spectrum (kernels.h: FullbandSet): seven scales
    -- a set of four and one of three -- and a single one, two epochs, against the oracle."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd import _lib
    from ghost_amd.synthetic import lfp
    option("blockconv", 0)
    fs, n = 1000.0, 70000
    x = lfp(2, n, fs, seed=44)
    eb = np.array([[3, 41000], [41007, n]])
    for f in (np.array([9.5, 8.0, 7.0, 5.5, 4.4, 3.1, 2.2]), np.array([5.0])):
        ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f, eb, gamma=3, beta=2) for c in range(2)])
        p = CwtPlan(n, 2, fs, f, gamma=3, beta=2, epoch_bounds=eb, output="complex")
        assert (p.scale_info()["method"] == _lib.SCALE_FULLBAND).all()
        assert rel_err(p.execute(x), ref).max() < TOL


def test_blockconv_against_the_reference_itself(golden):
    """G15: the reference's own numbers (not the oracle's) for Morse(3, 2), (1, 5), (3, 5) with kernels of 27 .. 2250
    taps over two epochs -- complex coefficients of its inner loop (transforms.py:187-204) and the public call's
    amplitude -- against the engine, whose scales here go through the time domain (<= 48 taps) and the block
    convolution."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd import _lib
    from ghost_amd.wave import ContinuousWaveletTransform, Morse
    g = golden("g15_blockconv.npz")
    fs, f, cols, eb = float(g["fs"]), g["frequencies"], g["cols"], g["epochs"]
    x = g["x"]
    for gamma, beta in g["pairs"]:
        tag = "%g_%g" % (gamma, beta)
        p = CwtPlan(x.size, 1, fs, f, gamma=float(gamma), beta=float(beta), epoch_bounds=eb, output="complex")
        si = p.scale_info()
        np.testing.assert_array_equal(si["length"], g["lengths_" + tag])
        assert (si["method"] == _lib.SCALE_BLOCKCONV).sum() >= 4 and (si["method"] == _lib.SCALE_SPECTRAL).sum() == 0
        got = p.execute(x[None])[0]
        err = np.abs(got[:, cols] - g["complex_cols_" + tag]).max(axis=1) / g["rowmax_" + tag]
        assert err.max() < TOL, (tag, err)
    cwt = ContinuousWaveletTransform(wavelet=Morse(gamma=3.0, beta=2.0))
    cwt.transform(x[:17000].astype(np.float64), fs=fs, freq_limits=[4, 120], voices_per_octave=4)
    np.testing.assert_allclose(cwt.frequencies, g["api_frequencies"], rtol=1e-13)
    c17 = cols[cols < 17000]
    assert rel_err(cwt.amplitude[:, c17], g["api_amplitude_cols"]).max() < TOL


def test_blockconv_blocks_pair_the_same_way_for_any_range(option):
    """The forward transform of the block convolution carries two real blocks at a time; which two must not depend
    on the range a call asks for, or execute_block differs from execute in a last bit now and then (soak seed 831,
    case 185: three epochs, power, precision='exact', block request 13952 + 34449).  Blocks are paired even-aligned in
    recording time (kernels.h: BcBlocks)."""
    from ghost_amd.engine import CwtPlan
    fs, n = 1000.0, 70000
    rng = np.random.default_rng(831)
    spec = np.fft.rfft(rng.standard_normal((3, n)), axis=1)
    x = (np.fft.irfft(spec / np.maximum(np.arange(spec.shape[1]), 1.0) ** 1.5, n=n, axis=1) * 40 + 7.0).astype(np.float32)
    eb = np.array([[2, 9458], [9460, 50145], [50146, 50545]])
    f = np.array([342.3177436147193, 336.88327324365497, 186.76392799751358, 148.86165586583928, 135.53799461991628,
                  123.2482109509881])
    option("direct_max_len", 48)
    for precision in ("exact", "high"):
        kw = dict(gamma=3.0, beta=20.0) if precision == "exact" else dict(gamma=3.0, beta=3.0)
        ff = f if precision == "exact" else np.array([120.0, 47.0, 19.0, 8.0, 4.0])
        p = CwtPlan(n, 3, fs, ff, epoch_bounds=eb, output="power", precision=precision, **kw)
        assert p.info["n_blockconv"] >= 4
        got = p.execute(x)
        for a, ln in [(13952, 34449), (1, 9457), (9459, 3), (3329, 3328), (6655, 50000), (50146, 399), (0, n)]:
            np.testing.assert_array_equal(p.execute_block(x, a, ln), got[:, :, a:a + ln])


def test_blockconv_many_epochs(option):
    """Forty epochs (more than one launch's sixteen), from 9 samples to several blocks long, gaps between them,
    both precisions: every epoch is convolved on its own, zero outside (transforms.py:185, convolution.py:68-87)."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd import _lib
    option("direct_max_len", 48)         # (a recording this ragged would keep its short kernels in the time domain)
    fs, n = 1000.0, 120000
    rng = np.random.default_rng(17)
    x = (rng.standard_normal((2, n)) * 3 + 0.1 * np.cumsum(rng.standard_normal((2, n)), axis=1)).astype(np.float32)
    cuts = np.sort(rng.choice(np.arange(10, n - 10), size=39, replace=False))
    edges = [0, *cuts.tolist(), n]
    eb = np.array([[a + int(rng.integers(0, 4)), b - int(rng.integers(0, 3))] for a, b in zip(edges[:-1], edges[1:]) if b - a > 12])
    eb[3, 1] = eb[3, 0] + 9
    f = np.array([150.0, 42.0, 11.0, 3.3])
    ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f, eb, gamma=2, beta=3) for c in range(2)])
    scale = np.abs(ref).max(axis=2, keepdims=True)
    for precision in ("high", "fast"):
        p = CwtPlan(n, 2, fs, f, gamma=2, beta=3, epoch_bounds=eb, output="complex", precision=precision)
        assert (p.scale_info()["method"] == _lib.SCALE_BLOCKCONV).sum() >= 3
        got = p.execute(x)
        assert (np.abs(got - ref) / scale).max() < TOL, precision
        inside = np.zeros(n, bool)
        for a, b in eb:
            inside[a:b] = True
        assert np.all(got[:, :, ~inside] == 0)
        a, b = int(eb[20, 0]) - 5, int(eb[24, 1]) + 7
        np.testing.assert_array_equal(p.execute_block(x, a, b - a), got[:, :, a:b])


@pytest.mark.parametrize("n, p1", [(1000000, 256), (2000003, 512), (4000000, 1024)])
def test_fullband_fused_passes(n, p1, option):
    """FFT lengths 2^20 .. 2^22: the full-band path's product rides the inverse row pass and its column pass
    crops and stores (kernels.hip: k_fullband_rows, k_fullband_cols256 / colsq).  Two epochs in one batch, the
    three output modes, against the oracle's fastconv of the same scales (convolution.py:68-87)."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd import _lib
    from ghost_amd.synthetic import lfp
    option("blockconv", 0)
    fs = 1000.0
    x = lfp(2, n, fs, seed=n % 97)
    f = np.array([9.0, 2.5])
    cut = n // 3 + 5
    eb = np.array([[0, cut], [cut + 40, n]]) if p1 == 256 else None
    ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f, eb, gamma=3, beta=2) for c in range(2)])
    for output in ("complex", "amplitude", "power"):
        p = CwtPlan(n, 2, fs, f, gamma=3, beta=2, epoch_bounds=eb, output=output)
        assert (p.scale_info()["method"] == _lib.SCALE_FULLBAND).all()
        assert max(seg[2] for seg in p.segments()) == p1 * 4096
        got = p.execute(x)
        want = ref if output == "complex" else np.abs(ref) if output == "amplitude" else np.abs(ref) ** 2
        assert rel_err(got, want).max() < (2 * TOL if output == "power" else TOL), output
        if eb is not None:
            assert np.all(got[:, :, cut:cut + 40] == 0)
        del p, got


def test_headline_workload_is_checked():
    """BASELINE config 3 exactly as bench.py times it -- 128 ch x 1e6 samples x 100 scales,
    amplitude, input and the 51 GB result resident in HBM: rows of channels {0, 7, 127} x
    scales {0, 57, 99} against the oracle (transforms.py:187-204), and the tiled channels
    c / c + 8 (same input) bit for bit."""
    from ghost_amd.engine import CwtPlan, DeviceBuffer
    from ghost_amd.synthetic import lfp
    fs, C, N, S = 1000.0, 128, 1000000, 100
    f = np.geomspace(200.0, 2.0, S)
    plan = CwtPlan(N, C, fs, f, output="amplitude")
    base = lfp(8, N, fs, seed=1234)
    xbuf = DeviceBuffer(4 * C * N)
    for c in range(C):
        xbuf.upload(base[c % 8], offset_bytes=4 * c * N)
    obuf = DeviceBuffer(plan.info["out_bytes"])
    plan.execute_device(xbuf, obuf)
    scales = [0, 57, 99]
    for c in (0, 7, 127):
        ref = orc.cwt_amplitude(base[c % 8].astype(np.float64), fs, f[scales])
        for i, s in enumerate(scales):
            row = obuf.download((N,), np.float32, offset_bytes=4 * (c * S + s) * N)
            assert rel_err(row, ref[i]) < TOL, (c, s)
            twin = c + 8 if c + 8 < C else c - 8
            other = obuf.download((N,), np.float32, offset_bytes=4 * (twin * S + s) * N)
            assert np.array_equal(row, other), (c, twin, s)
    obuf.free()
    xbuf.free()


def test_host_results_of_any_row_length(option):
    """Host output streams through pinned staging tiles; rows longer than a tile (recordings
    beyond 8 M samples with the real 32 MB tiles) are cut by columns.  Forced here with
    small tiles: float64 (the reference's dtype, transform()'s default) and float32,
    amplitude and complex, must equal the untiled result."""
    from ghost_amd.synthetic import lfp
    fs, n = 1000.0, 30011
    x = lfp(2, n, fs)
    f = [120.0, 30.0, 9.0]
    for output in ("amplitude", "complex"):
        p, ref = _plan(x, fs, f, output=output)
        ref64 = p.execute(x, wide=True)
        np.testing.assert_array_equal(ref64, ref.astype(ref64.dtype))
        option("stage_floats", 4099)
        np.testing.assert_array_equal(p.execute(x), ref)
        np.testing.assert_array_equal(p.execute(x, wide=True), ref64)
        np.testing.assert_array_equal(p.execute_block(x, 1234, 20000, wide=True), ref64[:, :, 1234:21234])
        option("stage_floats", None)
    cwt = _cwt(x[0], fs, freq_limits=[10, 100], voices_per_octave=4)
    a = cwt.amplitude.copy()
    option("stage_floats", 1000)
    cwt2 = _cwt(x[0], fs, freq_limits=[10, 100], voices_per_octave=4)
    assert cwt2.amplitude.dtype == np.float64
    np.testing.assert_array_equal(cwt2.amplitude, a)


class _FakeASA:
    """Duck-typed nelpy.RegularlySampledAnalogSignalArray (nelpy is not installable here):
    the attributes ghost/formats/preprocessing.py:78-114 and postprocessing.py:55-59 read."""

    def __init__(self, data_rowsig, fs, lengths, gap=5.0):
        self._data_rowsig = np.asarray(data_rowsig)
        self._data_colsig = self._data_rowsig.T
        self.data = self._data_rowsig
        self.n_signals = self._data_rowsig.shape[0]
        self.fs = fs
        self.lengths = np.asarray(lengths)
        t = np.arange(self._data_rowsig.shape[1]) / fs
        for e in np.cumsum(lengths)[:-1]:
            t[e:] += gap
        self.abscissa_vals = t
        self.support = "support-of-the-input"


def test_nelpy_round_trip_end_to_end(monkeypatch):
    """a two-epoch, three-signal ASA in -> transform() on the GPU -> result
    wrapped back into an AnalogSignalArray.  Input side: ghost/formats/preprocessing.py:78-114
    (epochs from the ASA's lengths -- cumulative); output side:
    ghost/formats/postprocessing.py:41-65 against a stub nelpy module."""
    import sys
    import types
    from ghost_amd.wave import ContinuousWaveletTransform
    from ghost_amd.formats import output_numpy_or_asa
    from ghost_amd.synthetic import lfp
    fs = 1000.0
    lengths = [5200, 3300]
    x = lfp(3, sum(lengths), fs, seed=77)
    asa = _FakeASA(x, fs, lengths)
    cwt = ContinuousWaveletTransform()
    cwt.transform(asa, multichannel=True, freq_limits=[12, 300], voices_per_octave=4)
    f = cwt.frequencies
    assert cwt.fs == fs and cwt.amplitude.shape == (3, f.size, sum(lengths))
    eb = np.array([[0, 5200], [5200, 8500]])
    for c in range(3):          # the reference handles one signal per call (transforms.py:57-58)
        ref = orc.cwt_amplitude(x[c].astype(np.float64), fs, f, eb)
        assert rel_err(cwt.amplitude[c], ref).max() < TOL, c
    # one-signal ASA through the reference's own signature (n_signals = 1, no flag)
    one = _FakeASA(x[1:2], fs, lengths)
    cwt1 = ContinuousWaveletTransform()
    cwt1.transform(one, freq_limits=[12, 300], voices_per_octave=4)
    np.testing.assert_array_equal(cwt1.amplitude, cwt.amplitude[1])
    np.testing.assert_array_equal(cwt1.time, one.abscissa_vals)

    # output adapter with a stub nelpy (postprocessing.py:55-59 builds the ASA around the input)
    made = {}

    class StubASA:
        def __init__(self, data, *, abscissa_vals, fs, support, labels=None):
            made.update(data=data, abscissa_vals=abscissa_vals, fs=fs, support=support, labels=labels)

    stub = types.ModuleType("nelpy")
    stub.AnalogSignalArray = StubASA
    stub.RegularlySampledAnalogSignalArray = _FakeASA
    monkeypatch.setitem(sys.modules, "nelpy", stub)
    spectrogram = cwt1.amplitude.T                        # (n_samples, n_freqs): one signal per scale
    out = output_numpy_or_asa(one, spectrogram, output_type="asa", labels=["%.1f Hz" % v for v in f])
    assert isinstance(out, StubASA)
    assert made["data"].shape == (f.size, sum(lengths))   # ASAs are (n_signals, n_samples)
    np.testing.assert_array_equal(made["data"], cwt1.amplitude)
    np.testing.assert_array_equal(made["abscissa_vals"], one.abscissa_vals)
    assert made["fs"] == fs and made["support"] == "support-of-the-input" and len(made["labels"]) == f.size
    assert output_numpy_or_asa(one, spectrogram) is spectrogram
    with pytest.raises(TypeError):
        output_numpy_or_asa(np.zeros(3), spectrogram, output_type="asa")


def test_synthesis_kernels_agree(option):
    """The three ways a level can be synthesised -- the interpolating kernel (k_synthi: amplitude
    and power at R >= 16), k_synth7 (option interp = 0 sends every level there) and the 16-column
    fallback (option synth16 = 1) -- give the same rows, both epochs of a recording whose gap and
    ends fall on no multiple of 4 samples (the interpolating kernel stores 16 bytes at a time and
    switches to single samples where a window edge runs through them).  In the measure build
    (GHOSTCWT_LIB=libghostcwt_measure.so) k_synth8 is compared as well."""
    from ghost_amd._lib import lib
    from ghost_amd.synthetic import lfp
    fs = 1000.0
    x = lfp(2, 40001, fs)
    f = np.geomspace(180.0, 3.0, 37)
    eb = [[0, 15001], [15006, 40001]]
    for output in ("amplitude", "power"):
        for name in ("interp", "synth16", "synth_cols"):
            option(name, None)
        p, prod = _plan(x, fs, f, output=output, epoch_bounds=eb)
        assert p.info["n_interp"] > 0 and any(d is not None for d in p.debug_interp()["levels"])
        scale = np.abs(prod).max(axis=-1, keepdims=True)
        option("interp", 0)
        p7, ref7 = _plan(x, fs, f, output=output, epoch_bounds=eb)
        assert p7.info["n_interp"] == 0
        assert (np.abs(prod - ref7) / scale).max() < 2e-6, output
        option("synth16", 1)
        _, ref16 = _plan(x, fs, f, output=output, epoch_bounds=eb)
        assert (np.abs(prod - ref16) / scale).max() < 2e-6, output
        option("synth16", None)
        # the pipelined interpolating kernel (synthp.hip: producer / consumer waves, double-buffered z; option
        # synthp = 1; measured a tie, so the measure build alone holds it) makes the same rows; blocks per workgroup
        # 1, 2 and 4 and both task shares
        option("interp", None)
        for lgnb, help_ in ((None, None), (0, 0), (1, 40), (2, None)) if lib.gcwt_debug_measure_build() else ():
            option("synthp", 1)
            option("synthp_lgnb", lgnb)
            option("synthp_help", help_)
            pp, gotp = _plan(x, fs, f, output=output, epoch_bounds=eb)
            assert pp.info["n_interp"] > 0
            assert (np.abs(gotp - ref7) / scale).max() < 2e-6, (output, lgnb, help_)
            assert not gotp[:, :, 15001:15006].any()
            # any sample range from the whole recording: window edges on no multiple of 4
            blk = pp.execute_block(x, 7777, 9001)
            np.testing.assert_array_equal(blk, gotp[:, :, 7777:7777 + 9001])
        if lib.gcwt_debug_measure_build():
            for name in ("synthp", "synthp_lgnb", "synthp_help"):
                option(name, None)
            for cols in (32, 16):
                option("synth_kernel", 8)
                option("synth_cols", cols)
                _, got = _plan(x, fs, f, output=output, epoch_bounds=eb)
                assert (np.abs(got - ref7) / scale).max() < 2e-6, (output, cols)
            option("synth_kernel", None)
            option("synth_cols", None)
        # zeros outside the epochs survive (transforms.py:185)
        assert not prod[:, :, 15001:15006].any()
    option("interp", None)
    # complex coefficients are never interpolated (the demodulation would have to be undone)
    pc, _ = _plan(x, fs, f, output="complex", epoch_bounds=eb)
    assert pc.info["n_interp"] == 0


def test_config5_regime_streamed_multichannel():
    """BASELINE config 5 at a size the oracle can follow: 3 channels @ 30 kHz, all 200 scales
    1-500 Hz (418 430-tap kernel at 1 Hz), the recording cut into overlapping time blocks that
    are streamed one by one into a device buffer (execute_block_device, as bench.py
    --config 5 does) by two 'ranks' that share the blocks (shard_time_blocks): the assembled
    result against the oracle on every 12th scale and the 1 Hz one, and against the
    whole-array call bit for bit."""
    from ghost_amd.dist import shard_time_blocks
    from ghost_amd.engine import CwtPlan, DeviceBuffer
    from ghost_amd.synthetic import lfp
    fs, C, n, S = 30000.0, 3, 2600000, 200
    f = np.geomspace(500.0, 1.0, S)
    x = lfp(C, n, fs, seed=505)
    plan = CwtPlan(n, C, fs, f, output="amplitude", max_fft_log2=21)
    segs = plan.segments()
    assert len(segs) >= 2 and all(s[2] == 1 << 21 for s in segs)
    assert plan.info["n_spectral"] == S and plan.scale_info()["decimation"].max() >= 8192
    xb = DeviceBuffer(x.nbytes)
    xb.upload(x)
    core = max(b - a for a, b, _ in segs)
    ring = DeviceBuffer(4 * C * S * core)
    out = np.empty((C, S, n), dtype=np.float32)
    first = True
    for rank in range(2):
        lo, hi = shard_time_blocks(segs, rank, 2)
        for a, b, _ in segs:
            if lo <= a < hi:
                plan.execute_block_device(xb, ring, a, b - a, reuse_means=not first)
                first = False
                out[:, :, a:b] = ring.download((C, S, b - a), np.float32)
    picks = sorted(set(range(0, S, 12)) | {S - 1})
    for c in range(C):
        ref = orc.cwt_amplitude(x[c].astype(np.float64), fs, f[picks], n_threads=8)
        assert rel_err(out[c][picks], ref).max() < TOL, c
    whole = plan.execute(x)
    np.testing.assert_array_equal(whole, out)


def test_config5_at_the_shape_the_bench_runs():
    """BASELINE config 5 at full size for one (channel group, time block), exactly as
    `bench.py --config 5` executes it: 24 of a GPU's 48 channels x 18e6 samples @ 30 kHz x 200
    scales 500..1 Hz, the middle time block (both edges are seams between blocks) into an 80 GB
    ring buffer.  Rows channels {0, 23} x scales {0, 100, 199 = 1 Hz, a 418 430-tap kernel} against
    the oracle over that block's window (transforms.py:529-597 streaming design, :187-204 numbers)."""
    import ctypes
    import importlib.util
    from ghost_amd.engine import CwtPlan, DeviceBuffer
    from ghost_amd.synthetic import lfp
    fs, group, n, S = 30000.0, 24, 18000000, 200
    f = np.geomspace(500.0, 1.0, S)
    plan = CwtPlan(n, group, fs, f, output="amplitude")
    segs = plan.segments()
    assert len(segs) >= 4 and plan.info["n_spectral"] == S
    base = lfp(2, n, fs, seed=1234)
    xb = DeviceBuffer(4 * group * n)
    for c in range(group):
        xb.upload(base[c % 2], offset_bytes=4 * c * n)
    a, b, _ = segs[len(segs) // 2]
    ring = DeviceBuffer(4 * group * S * (b - a))
    plan.execute_block_device(xb, ring, a, b - a)
    om = orc.hz_to_rad(f, fs)
    lengths = orc.morse_lengths(om)
    assert lengths[-1] == 418430
    worst = 0.0
    for c in (0, group - 1):
        xc = base[c % 2].astype(np.float64)
        xc -= xc.mean()                                              # transforms.py:142-143: global mean
        for sc in (0, S // 2, S - 1):
            L = int(lengths[sc])
            psi, _ = orc.morse_kernel(L, om[sc])
            w0, w1 = max(0, a - L), min(n, b + L)                    # every output needs the input within (L-1)/2
            ref = np.abs(orc.overlap_add_convolve(xc[w0:w1], psi)[a - w0:b - w0])
            row = ring.download((b - a,), np.float32, offset_bytes=4 * (c * S + sc) * (b - a))
            worst = max(worst, float(np.abs(row - ref).max() / ref.max()))
    print("config 5, block %d..%d: worst %.3g" % (a, b, worst))
    assert worst < TOL
    ring.free()
    xb.free()


def test_first_pass_windows_beside_strong_tones(option):
    """The synthesis skips, per scale, the first-pass inputs whose bins lie above the scale's
    band (gain below band_eps of the peak there: k_scale_windows).  A recording with tones 20 x
    the background just above the bands of three scales -- inside what is skipped -- must still
    meet the gate, and (measure build, which alone takes the accuracy-changing option prune_inputs)
    must agree with the build that computes every input."""
    from ghost_amd._lib import lib
    from ghost_amd.synthetic import lfp_channel
    fs, n = 1000.0, 60000
    f = np.geomspace(100.0, 5.0, 24)
    x = lfp_channel(n, fs, 3).astype(np.float64)
    t = np.arange(n) / fs
    amp = 20.0 * x.std()
    for k in (3, 10, 17):
        x += amp * np.sin(2 * np.pi * 1.9 * f[k] * t + k)
    x = x.astype(np.float32)
    ref = orc.cwt_complex(x.astype(np.float64), fs, f)
    p, c = _plan(x, fs, f, output="complex")
    assert rel_err(c[0], ref).max() < TOL
    p, a = _plan(x, fs, f, output="amplitude")
    assert rel_err(a[0], np.abs(ref)).max() < TOL
    if not lib.gcwt_debug_measure_build():
        return
    option("prune_inputs", 0)
    p, c_all = _plan(x, fs, f, output="complex")
    option("prune_inputs", None)
    assert rel_err(c_all[0], ref).max() < TOL
    diff = np.abs(c_all[0] - c[0]).max(axis=1) / np.abs(ref).max(axis=1)
    assert 0 < diff.max() < 2e-6, diff          # inputs are skipped, and what they carried is below the tolerance


def test_repeated_executes_are_bit_identical():
    """The level passes of a batch run on three streams that fork after the forward FFT and join
    before the synthesis (api.cpp: run_pipeline); a missing dependency would show up as
    run-to-run differences.  Seven decimation levels, two epochs, 8 repeats."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd.synthetic import lfp
    fs, C, n = 1000.0, 3, 120000
    f = np.geomspace(200.0, 2.0, 60)
    x = lfp(C, n, fs, seed=78)
    p = CwtPlan(n, C, fs, f, output="amplitude", epoch_bounds=[[0, 70000], [70300, n]])
    assert len(set(p.scale_info()["decimation"])) >= 6
    first = p.execute(x)
    for _ in range(8):
        np.testing.assert_array_equal(p.execute(x), first)


def test_fft_length_2_22_blocks():
    """bench.py --config 5's own FFT length: time blocks of 2^22 points (a 1024-point column
    pass: k_fft_colsq_real2, the column's four interleaved real subsequences two per FFT256), two channels with
    different offsets, scales from 500 Hz down to 1.5 Hz at 30 kHz, against the oracle; and the
    2^21-point plan of the same recording must give the same numbers to within rounding."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd.synthetic import lfp
    fs, C, n = 30000.0, 2, 4300000
    f = np.array([500.0, 170.0, 41.0, 9.5, 1.5])
    x = lfp(C, n, fs, seed=2222) + np.array([[40.0], [-3.0]], dtype=np.float32)
    plan = CwtPlan(n, C, fs, f, output="amplitude", max_fft_log2=22)
    segs = plan.segments()
    assert any(s[2] == 1 << 22 for s in segs), segs
    got = plan.execute(x)
    for c in range(C):
        ref = orc.cwt_amplitude(x[c].astype(np.float64), fs, f, n_threads=8)
        assert rel_err(got[c], ref).max() < TOL, c
    other = CwtPlan(n, C, fs, f, output="amplitude", max_fft_log2=21).execute(x)
    scale = got.max(axis=-1, keepdims=True)
    assert (np.abs(other - got) / scale).max() < 2e-6


def test_other_family_members_on_the_device(golden):
    """higher-order wavelets and the 'energy' normalisation on the device --
    G12, made by the reference's ``morsewave(..., n_wavelets=, normalization=)``
    (ghost/wave/morseutils.py:22-91) and ``fastconv_scipy``.  The bank / kernels are built
    from whatever spectrum samples the family member has; band and support are measured, so
    'energy' members (a spectrum that does not vanish at zero frequency) take the exact
    time-domain and full-band paths."""
    g = golden("g12_family.npz")
    fs, x, cols, f = float(g["fs"]), g["x"], g["cols"], g["frequencies"]
    for gamma, beta, energy, n_w in g["cases"]:
        norm = "energy" if energy else "bandpass"
        tag = "g%d_b%d_%s" % (gamma, beta, norm)
        for k in range(int(n_w)):
            p, c = _plan(x, fs, f, output="complex", gamma=gamma, beta=beta, normalization=norm, order=k)
            err = rel_err(c[0][:, cols], g["complex_cols_" + tag][k])
            print(tag, "order", k, "methods", p.scale_info()["method"].tolist(), "err", err.max())
            assert err.max() < TOL, (tag, k)


def test_large_offset_small_signal():
    """A channel whose mean is 1 000 x its spread (a DC-coupled amplifier): the reference
    removes the mean in float64 (transforms.py:142-143); here it is subtracted in fp64 before
    the samples become fp32 FFT input, so the signal keeps its low bits -- epoch edges, where
    a residual offset would show as a step, included."""
    from ghost_amd.synthetic import lfp
    fs, n = 1000.0, 10000
    rng = np.random.default_rng(131)
    x = (0.1 * rng.standard_normal((2, n)) + np.array([[97.3], [-81.9]])).astype(np.float32)
    f = np.array([390.0, 110.6, 67.3, 11.6, 5.5, 2.8])
    eb = [[0, 6100], [6103, n]]
    for gamma, beta in ((3, 20), (4, 7.4)):
        ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f, np.array(eb), gamma=gamma, beta=beta)
                        for c in range(2)])
        p, got = _plan(x, fs, f, output="complex", epoch_bounds=eb, gamma=gamma, beta=beta)
        assert rel_err(got, ref).max() < 3e-6, (gamma, beta)


def test_bench_runs_two_ranks_end_to_end(tmp_path):
    """`python bench.py --gpus 2` run bare: the launcher starts two fresh rank processes, each
    plans, uploads, times and (rank 0) checks its block against the oracle; one JSON line comes
    back.  The box has one GPU, so the ranks are allowed to share it for this rehearsal and
    the control plane is the file backend (RCCL refuses two ranks on one device); on a
    multi-GPU node the same command puts one rank on each GPU with RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(GHOSTCWT_ALLOW_SHARED_GPU="1", GHOSTCWT_COMM="file", GHOSTCWT_RDZV_DIR=str(tmp_path))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--channels", "8", "--samples", "150000", "--no-cpu-baseline",
                          "--no-ceilings"], env=env, capture_output=True, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["checked"] is True
    assert line["config"]["channels_total"] == 16 and line["config"]["comm"] == "file"
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    # without the rehearsal switch two ranks on one GPU are refused, loudly
    env.pop("GHOSTCWT_ALLOW_SHARED_GPU")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
                          "--warmup", "0", "--channels", "8", "--samples", "150000", "--no-cpu-baseline",
                          "--no-ceilings"], env=env, capture_output=True, timeout=600)
    assert res.returncode != 0 and b"ranks never share a GPU" in res.stderr


def test_bench_shards_time_blocks_over_two_ranks(tmp_path):
    """`python bench.py --gpus 2 --config 5 --shard time`: few channels, one long recording -- every rank holds
    all the channels and takes a contiguous run of the recording's time blocks (BASELINE.json north_star's split for
    config 5; no exchange: each block carries its halo).  Two ranks on this box's one GPU (file control plane), rank
    0 checks one of its blocks against the oracle; strong scaling is what the line says."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(GHOSTCWT_ALLOW_SHARED_GPU="1", GHOSTCWT_COMM="file", GHOSTCWT_RDZV_DIR=str(tmp_path))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "5", "--shard", "time",
                          "--channels", "2", "--group", "2", "--samples", "6000000", "--max-fft-log2", "21",
                          "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-ceilings", "--sustain", "0"],
                         env=env, capture_output=True, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["checked"] is True, line
    assert line["config"]["channels_total"] == 2 and "time-block-sharded x2" in line["config"]["parallelism"]
    # the whole job is the recording once, whatever the number of ranks
    assert line["value"] == pytest.approx(2 * 6000000 / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)


def test_split_levels_option(option):
    """Option split_levels = 1 (two block grids per decimation, x_R shared; measured slower on
    the headline workload, kept as an option) gives the same numbers to rounding."""
    from ghost_amd.synthetic import lfp
    fs = 1000.0
    x = lfp(2, 30000, fs)
    f = np.geomspace(190.0, 3.0, 41)
    p0, ref = _plan(x, fs, f, output="complex")
    option("split_levels", 1)
    p1, got = _plan(x, fs, f, output="complex")
    assert p1.info["n_levels"] > p0.info["n_levels"]
    assert set(p1.scale_info()["halo"].tolist()) >= {16} and p1.scale_info()["halo"].max() > 16
    scale = np.abs(ref).max(axis=-1, keepdims=True)
    assert (np.abs(got - ref) / scale).max() < 2e-6
    oracle = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, f) for c in range(2)])
    assert rel_err(got, oracle).max() < TOL


def test_time_domain_kernel_every_alignment_and_edge(option):
    """The time-domain kernel stores a scale's taps behind (7 - (L-1)//2) mod 8 zeros and works
    in groups of 8 taps, two groups per trip (csrc/kernels.hip: k_direct): lengths covering
    every residue and both parities of the group count, kernels up to the longest it takes,
    all output modes, tiles that end inside the range, and block requests whose first column
    is not a multiple of four samples (the store path without 16-byte alignment)."""
    from ghost_amd.engine import CwtPlan
    option("direct_max_len", 256)                  # (by default kernels beyond 48 taps go by blocks)
    fs, n = 1000.0, 6200                           # three tiles of 2048 outputs + a ragged tail
    rng = np.random.default_rng(77)
    x = (rng.standard_normal((3, n)) + np.array([[0.7], [-2.0], [0.0]])).astype(np.float32)
    # Morse(3, 20): one frequency per kernel length 33 .. 48 (highest frequency of each length)
    grid = np.linspace(0.30 * fs, 0.46 * fs, 4001)
    lens = orc.morse_lengths(orc.hz_to_rad(grid, fs))
    f_short = np.array([grid[lens == L].max() for L in range(33, 49) if np.any(lens == L)])[::-1]
    assert len(f_short) >= 14
    for gamma, beta, freqs in [(3.0, 20.0, f_short), (2.0, 2.0, np.array([330.0, 97.0, 60.0, 40.0, 26.0, 15.0]))]:
        eb = np.array([[0, 2501], [2503, n]])
        ref = np.stack([orc.cwt_complex(x[c].astype(np.float64), fs, freqs, eb, gamma=gamma, beta=beta)
                        for c in range(3)])
        scale = np.abs(ref).max(axis=2, keepdims=True)
        for output in ("complex", "amplitude", "power"):
            p = CwtPlan(n, 3, fs, freqs, gamma=gamma, beta=beta, epoch_bounds=eb, output=output)
            si = p.scale_info()
            direct = si["method"] == 1
            assert direct.sum() >= 3, (beta, si["method"])
            if beta == 20.0:
                assert len({int((L - 1) // 2) % 8 for L in si["length"][direct]}) == 8
            else:
                assert si["length"][direct].max() > 230 and si["length"][direct].min() < 16
            want = ref if output == "complex" else np.abs(ref) if output == "amplitude" else np.abs(ref) ** 2
            sc = scale if output != "power" else scale ** 2
            tol = 2 * TOL if output == "power" else TOL
            got = p.execute(x)
            assert (np.abs(got - want) / sc)[:, direct].max() < tol, (beta, output)
            assert np.all(got[:, :, 2501:2503] == 0)
            for start, length in [(1, 2047), (2049, 2050), (2502, 3698), (3, 1)]:
                blk = p.execute_block(x, start, length)
                np.testing.assert_array_equal(blk, got[:, :, start:start + length])


def test_devices_list_shards_the_channels_over_device_slots(golden):
    """``transform(multichannel=True, devices=[...])``: contiguous channel blocks, one plan and one host thread per
    entry (ghost_amd/multi.py; SURVEY 8(e): channels are independent, transforms.py:57-58; config 4's split).  An entry
    may repeat, so a one-GPU box runs two and three slots on its one device: the rows must be the bits one plan over
    all the channels makes, whole (``amplitude``) and piecewise (``fetch`` across a shard boundary), and the goldens
    of the reference hold for them."""
    from ghost_amd.wave import ContinuousWaveletTransform
    from ghost_amd.multi import ShardedResult
    from ghost_amd.synthetic import lfp
    g = golden("g8_multichannel.npz")
    kw = dict(fs=1000.0, freq_limits=[20, 250], voices_per_octave=4, multichannel=True, dtype=np.float32)
    one = ContinuousWaveletTransform(); one.transform(g["x"], **kw)
    two = ContinuousWaveletTransform(); two.transform(g["x"], devices=[0, 0], **kw)
    assert isinstance(two.device_result, ShardedResult) and [p[:2] for p in two.device_result.parts] == [(0, 2), (2, 3)]
    np.testing.assert_allclose(two.frequencies, g["frequencies"], rtol=1e-14)
    assert two.amplitude.shape == g["amplitude"].shape
    np.testing.assert_array_equal(two.amplitude, one.amplitude)
    for c in range(3):
        assert rel_err(two.amplitude[c], g["amplitude"][c]).max() < TOL
    with pytest.raises(ValueError):
        two.transform(g["x"], devices=[0, 0], device=0, **kw)
    with pytest.raises(ValueError):
        two.transform(g["x"], devices=[], **kw)
    # 128 channels (the headline's count) in two and three slots, float64 power, a second call on the same object
    x = np.tile(lfp(8, 100000), (16, 1)) * np.linspace(0.5, 2.0, 128, dtype=np.float32)[:, None]
    kw = dict(fs=1000.0, freq_limits=[2, 200], voices_per_octave=4, multichannel=True, output="power")
    one = ContinuousWaveletTransform(); one.transform(x, **kw)
    ref = one.power
    for devs, blocks in (([0, 0], [(0, 64), (64, 128)]), ([0, 0, 0], [(0, 43), (43, 86), (86, 128)])):
        many = ContinuousWaveletTransform()
        for rep in range(2):
            many.transform(x if rep == 0 else x[::-1].copy(), devices=devs, **kw)
        many.transform(x, devices=devs, **kw)
        assert [p[:2] for p in many.device_result.parts] == blocks
        assert many.device_result.shape == (128, len(many.frequencies), 100000)
        piece = many.fetch(scales=slice(3, 9), start=777, stop=5003)           # across every shard boundary
        np.testing.assert_array_equal(piece, ref[:, 3:9, 777:5003])
        assert many.power.dtype == np.float64
        np.testing.assert_array_equal(many.power, ref)
        rep = many.precision_report
        assert rep["rerouted"] == 0 and len(rep["per_device"]) == len(devs)
        many.release_device()
        assert many.device_result is None
