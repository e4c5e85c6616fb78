"""Host logic and the C-ABI surface (CPU only, no compute calls)."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from ghost_amd import _lib
from ghost_amd._lib import GhostCwtError
from ghost_amd.engine import CwtPlan
from ghost_amd.utils import get_contiguous_segments, is_sorted
from ghost_amd.wave import ContinuousWaveletTransform, Morse, Wavelet
from ghost_amd.wave import morseutils


def _grid_only(cwt, *args, **kwargs):
    """Run transform() far enough to build the grid; the device call itself may fail
    on a CPU-only box (there is no CPU fallback, by design)."""
    try:
        cwt.transform(*args, **kwargs)
    except GhostCwtError as e:
        assert e.code == _lib.ERR_NO_DEVICE
    return cwt.frequencies


def test_abi_exports_every_declared_symbol():
    declared = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        declared |= set(re.findall(r"\b(gcwt_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 30
    so = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(so, s)]
    assert not missing, missing
    assert _lib.lib.gcwt_abi_version() == 5


def test_morse_scalars_and_lengths(golden):
    g = golden("g4_scalars.npz")
    assert morseutils.morsefreq(3, 20) == pytest.approx(float(g["morsefreq"]), rel=1e-15)
    assert morseutils.morsehigh(3, 20) == pytest.approx(float(g["morsehigh"]), rel=1e-15)
    assert morseutils.morsehigh(2, 8) == pytest.approx(float(g["morsehigh_g2_b8"]), rel=1e-15)
    m = Morse()
    for n, key in [(16384, "bounds_16384"), (1000000, "bounds_1e6"), (4096, "bounds_4096")]:
        np.testing.assert_allclose(m.compute_freq_bounds(n), g[key], rtol=1e-15)
    m.fs = 1000.0
    np.testing.assert_array_equal(
        m.compute_lengths(g["len_freqs_hz"] / 500.0 * np.pi), g["lengths_1khz"])
    assert m.time_bandwidth == 60
    assert isinstance(m, Wavelet) and repr(m) == "Morse"
    c = m.copy()
    c.gamma = 2
    assert m.gamma == 3


def test_morse_call_matches_reference_kernels(golden):
    g = golden("g3_kernels.npz")
    for L in (36, 40, 70, 279, 1163, 1395):
        m = Morse(fs=1000.0)
        m.norm_radian_freq = float(g["omega_%d" % L])
        psi, psif = m(L)
        np.testing.assert_allclose(psif, g["psif_%d" % L], rtol=1e-12, atol=1e-300)
        assert np.abs(psi - g["psi_%d" % L]).max() < 1e-13 * np.abs(g["psi_%d" % L]).max()
    with pytest.raises(ValueError):
        Morse()(0)
    with pytest.raises(ValueError):
        Morse()(10, normalization="nope")
    for bad in (dict(fs=0), dict(gamma=-1), dict(beta=0), dict(freq=-3)):
        with pytest.raises(ValueError):
            Morse(**bad)


def test_frequency_grid_matches_reference(golden):
    g1 = golden("g1_config1.npz")
    x = g1["x"].astype(np.float64)
    cwt = ContinuousWaveletTransform()
    f = _grid_only(cwt, x, fs=1000.0, freq_limits=[5, 200], voices_per_octave=6)
    np.testing.assert_allclose(f, g1["frequencies"], rtol=1e-14)
    assert np.all(np.diff(f) < 0)                      # descending
    g4 = golden("g4_scalars.npz")
    f = _grid_only(ContinuousWaveletTransform(), x, fs=1000.0)
    np.testing.assert_allclose(f, g4["default_grid_16384"], rtol=1e-14)
    gb = golden("g4b_two_tone.npz")
    f = _grid_only(ContinuousWaveletTransform(), np.zeros(4096), fs=1000.0,
                   freq_limits=[10, 100], voices_per_octave=4)
    np.testing.assert_allclose(f, gb["frequencies"], rtol=1e-14)   # floor clamped to 17.03 Hz
    # explicit list: sorted ascending, clamped to the valid range (intent of transforms.py:151-158)
    f = _grid_only(ContinuousWaveletTransform(), np.zeros(4096), fs=1000.0,
                   freqs=[300.0, 50.0, 1.0, 20.0, 450.0])
    np.testing.assert_array_equal(f, [20.0, 50.0, 300.0])


def test_two_epoch_grid_uses_shortest_epoch(golden):
    g = golden("g5_two_epochs.npz")
    cwt = ContinuousWaveletTransform()
    f = _grid_only(cwt, g["x"].astype(np.float64), fs=1000.0, timestamps=g["timestamps"])
    np.testing.assert_allclose(f, g["frequencies"], rtol=1e-14)
    assert f.size == 45


def test_error_surface():
    cwt = ContinuousWaveletTransform()
    x = np.zeros(5000)
    with pytest.raises(ValueError):
        cwt.transform(x, fs=1000, voices_per_octave=5)
    with pytest.raises(ValueError):
        cwt.transform(x, fs=1000, freqs=[10, 20], freq_limits=[10, 20])
    with pytest.raises(ValueError):
        cwt.transform(x, fs=1000, parallel=3)
    with pytest.raises(ValueError):
        cwt.transform(x, fs=1000, verbose="yes")
    with pytest.raises(ValueError):
        cwt.transform(x, fs=-1)
    with pytest.raises(TypeError, match="missing 1 required keyword argument: 'fs'"):
        cwt.transform(x)
    with pytest.raises(TypeError):
        cwt.transform([1.0, 2.0, 3.0], fs=1000)
    with pytest.raises(ValueError):
        cwt.transform(np.zeros((2, 5000)), fs=1000)
    with pytest.raises(ValueError):
        cwt.transform(np.zeros((5000, 2)), fs=1000)
    with pytest.raises(ValueError):
        cwt.transform(x, fs=1000, timestamps=np.zeros(4999))
    for attr in ("frequencies", "amplitude", "power", "time"):
        with pytest.raises(ValueError):
            setattr(cwt, attr, 1)
    with pytest.raises(TypeError):
        cwt.fs = 1000.0
        cwt.wavelet = "morse"
    assert repr(cwt) == "ContinuousWaveletTransform"
    # shapes with one non-singleton dimension are fine up to the device call
    for shape in ((1, 5000), (5000, 1)):
        f = _grid_only(ContinuousWaveletTransform(), np.zeros(shape), fs=1000.0)
        assert f.size == 49


def test_contiguous_segments():
    fs = 1000.0
    t = np.arange(10000) / fs
    t[6000:] += 10.0
    np.testing.assert_array_equal(
        get_contiguous_segments(t, step=1 / fs, index=True), [[0, 6000], [6000, 10000]])
    np.testing.assert_array_equal(
        get_contiguous_segments(t, step=1 / fs, index=True, inclusive=True),
        [[0, 5999], [6000, 9999]])
    vals = get_contiguous_segments(t, step=1 / fs)
    np.testing.assert_allclose(vals, [[0.0, 6.0], [16.0, 20.0]])
    assert get_contiguous_segments(np.arange(5.0), step=1.0, index=True).tolist() == [[0, 5]]
    assert is_sorted([1, 2, 2, 3]) and not is_sorted([2, 1])
    with pytest.raises(TypeError):
        is_sorted("abc")


class FakeASA:
    """Duck-typed nelpy.RegularlySampledAnalogSignalArray (nelpy is not installed)."""

    def __init__(self, data_rowsig, fs, lengths):
        self._data_rowsig = np.asarray(data_rowsig)
        self._data_colsig = self._data_rowsig.T
        self.n_signals = self._data_rowsig.shape[0]
        self.fs = fs
        self.lengths = np.asarray(lengths)
        n = self._data_rowsig.shape[1]
        t = np.arange(n) / fs
        edge = np.cumsum(lengths)[:-1]
        for e in edge:
            t[e:] += 5.0
        self.abscissa_vals = t


def test_asa_adapter():
    from ghost_amd.formats import standardize_asa, is_asa_like
    seen = {}

    @standardize_asa(x="data", fs="fs", n_signals=1, abscissa_vals="timestamps")
    def fn(data, *, fs=None, timestamps=None, epoch_bounds=None):
        seen.update(data=data, fs=fs, t=timestamps, eb=epoch_bounds)

    asa = FakeASA(np.arange(10.0)[None, :], 100.0, [6, 4])
    assert is_asa_like(asa) and not is_asa_like(np.zeros(3))
    fn(asa)
    assert seen["data"].shape == (10, 1) and seen["fs"] == 100.0
    np.testing.assert_array_equal(seen["eb"], [[0, 6], [6, 10]])   # cumulative, unlike the reference
    fn(data=asa, fs=5.0)                                            # object's fs wins
    assert seen["fs"] == 100.0
    with pytest.raises(ValueError):
        fn(FakeASA(np.zeros((2, 10)), 100.0, [10]))
    fn(np.arange(8.0), fs=2.0)
    np.testing.assert_allclose(seen["t"], np.arange(8) / 2.0)      # default timestamps work
    np.testing.assert_array_equal(seen["eb"], [[0, 8]])
    with pytest.raises(TypeError):
        standardize_asa(x=3)
    with pytest.raises(ValueError):
        standardize_asa(x="data", n_signals=0)
    # transform() accepts the ASA and takes fs / epochs from it
    cwt = ContinuousWaveletTransform()
    big = FakeASA(np.random.default_rng(0).standard_normal((1, 10000)), 1000.0, [6000, 4000])
    f = _grid_only(cwt, big)
    assert cwt.fs == 1000.0 and f.size == 45


def test_planner_through_the_abi():
    f = np.geomspace(200, 2, 100)
    p = CwtPlan(1000000, 128, 1000.0, f)
    info = p.info
    assert info["n_spectral"] == 100 and info["n_direct"] == 0
    assert info["fft_length"] == 1 << 20 and info["block"] == 256
    assert info["out_bytes"] == 128 * 100 * 1000000 * 4
    si = p.scale_info()
    assert si["length"][0] == 70 and si["length"][-1] == 6974
    # the three scales that could run at R = 256 are walked by the R = 128 workgroups (planner.cpp)
    assert si["decimation"][0] == 2 and si["decimation"][-1] == 128
    assert np.all(np.diff(si["decimation"]) >= 0)
    assert np.all(si["hop"] >= 32) and np.all(si["hop"] + 2 * si["halo"] == 256)
    # the measured band of every spectral scale fits its decimated band
    assert np.all(si["theta_hi"] * si["decimation"] <= 2 * np.pi)
    assert np.all(si["theta_hi"] > 1.5 * 2 * np.pi * f / 1000.0)
    # scales whose response reaches Nyquist go to the direct path
    p2 = CwtPlan(4096, 1, 1000.0, [391.0, 300.0, 280.0, 270.0, 200.0])
    assert p2.scale_info()["method"].tolist() == [1, 1, 0, 0, 0]
    try:
        p2.upload()                          # succeeds on a GPU box
    except GhostCwtError as e:               # fails loudly without one: there is no CPU path
        assert e.code == _lib.ERR_NO_DEVICE and "no CPU path" in str(e)


def test_operator_entry_points_validate_then_need_a_device():
    """gcwt_fastconv / gcwt_dft / gcwt_analytic_signal: argument errors come back as
    GCWT_ERR_INVALID / UNSUPPORTED before any device is touched; a valid request on a box
    without a GPU fails with GCWT_ERR_NO_DEVICE -- never a CPU result."""
    from ghost_amd import sigtools
    lib = _lib.lib
    buf = (ctypes.c_float * 64)()
    assert lib.gcwt_dft(None, 8, 0, 0, buf, -1) == _lib.ERR_INVALID
    assert lib.gcwt_dft(buf, 0, 0, 0, buf, -1) == _lib.ERR_INVALID
    assert lib.gcwt_dft(buf, (1 << 21) + 1, 0, 0, buf, -1) == _lib.ERR_UNSUPPORTED
    assert lib.gcwt_analytic_signal(buf, 0, 0, buf, -1) == _lib.ERR_INVALID
    assert b"empty" in lib.gcwt_last_error()
    assert lib.gcwt_analytic_signal(buf, 16, 8, buf, -1) == _lib.ERR_INVALID
    assert b"fft_length" in lib.gcwt_last_error()
    assert lib.gcwt_analytic_signal(buf, 16, (1 << 21) + 1, buf, -1) == _lib.ERR_UNSUPPORTED
    assert lib.gcwt_fastconv(buf, 8, buf, 4, 0, 7, buf, -1) == _lib.ERR_INVALID
    x = np.arange(16.0)
    for call in (lambda: sigtools.analytic_signal_hip(x), lambda: sigtools.chirpz_dft_hip(x),
                 lambda: sigtools.fastconv_hip(x, np.ones(3))):
        try:
            res = call()
            assert res.shape[0] == 16        # GPU box
        except GhostCwtError as e:
            assert e.code == _lib.ERR_NO_DEVICE and "no CPU path" in str(e)
    # the Python wrappers raise the reference's exceptions before calling down
    with pytest.raises(ValueError):
        sigtools.analytic_signal_hip(x, fft_length=8)
    with pytest.raises(ValueError):
        sigtools.chirpz_dft_hip(np.zeros((2, 2)))


def test_planner_rejects_bad_requests():
    inf, nan = float("inf"), float("nan")
    for kw in (dict(n_samples=0), dict(fs=-1.0), dict(freqs=[-5.0]), dict(gamma=0.0),
               dict(bounds=[[0, 20000]]), dict(bounds=[[5, 5]]),
               # not a frequency: inf and 1e300 used to plan a one-tap "kernel", nan a length of INT64_MIN
               dict(freqs=[inf]), dict(freqs=[10.0, nan]), dict(freqs=[1e300]), dict(freqs=[500.1]),
               dict(fs=inf), dict(fs=nan), dict(gamma=inf), dict(gamma=nan),
               # epochs that share samples: both would write them
               dict(bounds=[[0, 6000], [5999, 10000]]), dict(bounds=[[4000, 10000], [0, 4001]]),
               dict(bounds=[[0, 100], [0, 100]])):
        args = dict(n_samples=10000, fs=1000.0, freqs=[10.0], gamma=3.0, bounds=None)
        args.update(kw)
        with pytest.raises(GhostCwtError) as e:
            CwtPlan(args["n_samples"], 1, args["fs"], args["freqs"], gamma=args["gamma"],
                    epoch_bounds=args["bounds"])
        assert e.value.code == _lib.ERR_INVALID
    # touching epochs in any order, and the Nyquist frequency itself, are requests
    CwtPlan(10000, 1, 1000.0, [10.0], epoch_bounds=[[6000, 10000], [0, 6000]])
    CwtPlan(10000, 1, 1000.0, [500.0])
    # a 3.2 M-tap kernel: refused until round 4, FFTs of 2^23 points (long mode) since
    assert CwtPlan(1 << 20, 1, 30000.0, [0.13]).segments()[0][2] == 1 << 23
    with pytest.raises(GhostCwtError) as e:
        CwtPlan(1 << 20, 1, 30000.0, [0.02])            # 21 M taps: no room for time blocks even at 2^24
    assert e.value.code == _lib.ERR_UNSUPPORTED


def test_decimated_model_matches_oracle(golden):
    """The algorithm the kernels implement, in float64 NumPy and with the planner's own
    decisions, against the goldens."""
    from decimated_model import cwt_decimated
    from conftest import rel_err
    g = golden("g2_complex_small.npz")
    c = cwt_decimated(g["x"], float(g["fs"]), g["frequencies"])
    assert rel_err(c, g["coeffs"]).max() < 5e-7      # block edges cut 4.5e-6 of the kernel's energy
    g = golden("g5_two_epochs.npz")
    c = cwt_decimated(g["x"], float(g["fs"]), g["frequencies"][::6], g["epoch_bounds"])
    assert rel_err(c[:, g["cols"]], g["complex_cols"][::6]).max() < 5e-7


def test_reference_import_name_is_an_alias():
    """`from ghost.wave import ContinuousWaveletTransform, Morse` -- the reference's import line
    (ghost/wave/__init__.py:3-5) -- works from the repository root and yields ghost_amd's own
    objects."""
    import ghost_amd
    import ghost_amd.wave.transforms
    from ghost.wave import ContinuousWaveletTransform, Morse
    import ghost.wave.morseutils as mu
    import ghost.sigtools.convolution as conv
    from ghost.formats import standardize_asa
    import ghost
    assert ContinuousWaveletTransform is ghost_amd.wave.transforms.ContinuousWaveletTransform
    assert Morse is ghost_amd.wave.Morse and mu is ghost_amd.wave.morseutils
    assert standardize_asa is ghost_amd.formats.standardize_asa
    assert ghost.__version__ == ghost_amd.__version__
    # the reference's operator names (ghost/sigtools/convolution.py:3-4, analytic.py:3, fourier.py:9) exist and
    # are bound to the GPU operators, the GPU names stay importable beside them
    from ghost.sigtools import (fastconv_scipy, fastconv_fftw, fastconv_freq_scipy, fastconv_freq_fftw,   # noqa: F401
                                analytic_signal_fftw, chirpz_dft, fastconv_hip)
    import inspect
    assert conv.fastconv_hip is ghost_amd.sigtools.convolution.fastconv_hip is fastconv_hip
    assert list(inspect.signature(fastconv_fftw).parameters) == ["signal", "kernel", "mode", "fft_length", "n_threads"]
    assert list(inspect.signature(fastconv_freq_fftw).parameters) == ["signal_td", "kernel_fd", "kernel_len", "mode", "n_threads"]
    assert list(inspect.signature(analytic_signal_fftw).parameters) == ["signal", "fft_length", "n_threads"]
    with pytest.raises(ValueError):
        fastconv_scipy(np.zeros((2, 8)), np.ones(3))        # "Signal must be 1D" before anything touches a device


def test_interpolated_levels_model_matches_oracle(golden):
    """Amplitude rows of the levels the planner hands to the interpolating synthesis
    (csrc/synthi.hip): q phases of the block transform, demodulated to each scale's band centre,
    then the planner's own 8-tap interpolators -- in float64 NumPy against the goldens.  Also
    what the planner promises about the design: a bound below 2e-7, demodulation bins that are
    multiples of q, interpolators that reproduce a constant."""
    from decimated_model import amplitude_interpolated
    from conftest import rel_err
    from ghost_amd.engine import CwtPlan
    g = golden("g1_config1.npz")
    x, f = g["x"], g["frequencies"]
    plan = CwtPlan(x.size, 1, 1000.0, f, output="amplitude")
    di = plan.debug_interp()
    designed = [(lv, d) for lv, d in zip(plan.debug_levels(), di["levels"]) if d is not None]
    assert designed and all(lv["decimation"] >= 16 for lv, _ in designed)
    assert plan.info["n_interp"] == sum((plan.scale_info()["decimation"] == lv["decimation"]).sum() for lv, _ in designed)
    for lv, d in designed:
        assert d["q"] * d["factor"] == lv["decimation"] and d["q"] in (2, 4, 8, 16)
        assert 0 < d["err_bound"] <= 2e-7 and 0 < d["alpha"] < 0.9
        assert d["coef"].shape == (2, d["factor"], 8)
        np.testing.assert_allclose(d["coef"].sum(axis=2), 1.0, atol=2e-6)      # DC passes unchanged
    demod = di["demod"][np.isin(plan.scale_info()["decimation"], [lv["decimation"] for lv, _ in designed])]
    assert (demod % 2 == 0).all() and (demod > 0).all() and (demod < 256).all()
    a = amplitude_interpolated(x, 1000.0, f, plan=plan)
    assert rel_err(a[:, g["cols"]], g["amplitude_cols"]).max() < 5e-7
    # complex output is never interpolated; option interp = 0 switches the design off (A/B runs)
    assert CwtPlan(x.size, 1, 1000.0, f, output="complex").info["n_interp"] == 0
    g = golden("g5_two_epochs.npz")
    a = amplitude_interpolated(g["x"], float(g["fs"]), g["frequencies"][::4], g["epoch_bounds"])
    assert rel_err(a[:, g["cols"]], np.abs(g["complex_cols"][::4])).max() < 5e-7


GB_PAIRS = [(3, 8), (3, 4), (3, 2), (2, 8), (4, 30), (1, 5)]


def test_exact_gain_is_the_response_of_the_reference_kernel():
    """csrc/morse_exact.h (host evaluation through the debug hook) and its NumPy twin
    against the DTFT of the literal kernel (oracle), for light and heavy tails, odd and
    even L, on the bank's grid and on a full-band grid."""
    from decimated_model import exact_gain, exact_response
    from oracle import ghost_oracle as orc
    fs = 1000.0
    for gamma, beta in [(3, 20)] + GB_PAIRS:
        f = np.array([390.0, 140.0, 40.0, 11.0])
        plan = CwtPlan(20000, 1, fs, f, gamma=gamma, beta=beta)
        si = plan.scale_info()
        om = orc.hz_to_rad(f, fs)
        np.testing.assert_array_equal(si["length"], orc.morse_lengths(om, gamma, beta))
        for i in range(f.size):
            L = int(si["length"][i])
            for b in (256 * 8, 4096):
                a = np.arange(0, b, 7)
                theta = 2 * np.pi * a / b
                ref = orc.kernel_response(theta, om[i], L, gamma, beta)
                d = (L - 1) / 2 - (L - 1) // 2
                ref_gain = (ref * np.exp(1j * theta * d))
                assert np.abs(ref_gain.imag).max() < 1e-12          # G is real
                twin = exact_gain(theta, om[i], L, gamma, beta)
                host = plan.debug_exact_gain(i, a, b)
                assert np.abs(twin - ref_gain.real).max() < 1e-12, (gamma, beta, f[i])
                assert np.abs(host - ref_gain.real).max() < 1e-12, (gamma, beta, f[i])
                assert np.abs(exact_response(theta, om[i], L, gamma, beta) - ref).max() < 1e-12


def test_other_family_members_on_the_host(golden):
    """Higher orders and 'energy' normalisation (morseutils.py:119-124, :181-196): the
    planner's spectrum samples, evaluated by the library's host code, give the response of
    the reference's kernel; the planner-driven float64 model reproduces G12."""
    from decimated_model import cwt_decimated, exact_gain
    from conftest import rel_err
    from oracle import ghost_oracle as orc
    g = golden("g12_family.npz")
    fs, x, cols, f = float(g["fs"]), g["x"], g["cols"], g["frequencies"]
    om = orc.hz_to_rad(f, fs)
    for gamma, beta, energy, n_w in g["cases"]:
        norm = "energy" if energy else "bandpass"
        tag = "g%d_b%d_%s" % (gamma, beta, norm)
        for k in range(int(n_w)):
            plan = CwtPlan(x.size, 1, fs, f, gamma=gamma, beta=beta, normalization=norm, order=k)
            si = plan.scale_info()
            for i in (1, 3):
                L = int(si["length"][i])
                a, b = np.arange(0, 4096, 11), 4096
                theta = 2 * np.pi * a / b
                ref = orc.kernel_response(theta, om[i], L, gamma, beta, norm, k)
                d = (L - 1) / 2 - (L - 1) // 2
                ref_gain = (ref * np.exp(1j * theta * d)).real
                scale = np.abs(ref_gain).max()
                assert np.abs(plan.debug_exact_gain(i, a, b) - ref_gain).max() < 1e-12 * scale, (tag, k, i)
                assert np.abs(exact_gain(theta, om[i], L, gamma, beta, norm, k) - ref_gain).max() < 1e-12 * scale
            if k == int(n_w) - 1:
                c = cwt_decimated(x, fs, f, gamma=gamma, beta=beta, plan=plan, normalization=norm, order=k)
                assert rel_err(c[:, cols], g["complex_cols_" + tag][k]).max() < 2e-6, (tag, k)
    with pytest.raises(ValueError):
        CwtPlan(1000, 1, fs, [10.0], normalization="peak")
    with pytest.raises(ValueError):
        CwtPlan(1000, 1, fs, [10.0], order=99)


def test_planner_measures_each_wavelet():
    """What the planner decides from the measured support of the kernel's response:
    the default wavelet keeps the fast path with hop >= 206 (212, and 206 for the level
    that takes in the three scales of the thin top decimation).  Heavy-tailed wavelets, whose
    L-tap truncation leaves side lobes above band_eps below zero frequency: where those die out
    within a few omega ((3,4), (2,8)) the scale keeps a decimated level whose band is shifted
    below zero to hold them -- at a lower decimation and a longer block halo than the default
    wavelet's -- and only the short kernels at the top go to the time domain; where they reach
    Nyquist ((3,2), (1,5)) the scale leaves the decimated path entirely."""
    fs, f = 1000.0, np.geomspace(200.0, 2.0, 100)
    p = CwtPlan(1000000, 128, fs, f)
    si, info = p.scale_info(), p.info
    assert info["n_spectral"] == 100 and info["n_direct"] == 0 and info["n_fullband"] == 0
    assert si["hop"].min() >= 206 and si["halo"].max() <= 32
    assert np.all(si["theta_hi"] * si["decimation"] <= 2 * np.pi * (1 + 1e-12))
    assert 0.80 < (si["support"] / (si["length"] / 2)).max() < 0.86
    # default grid: the top scales reach Nyquist and are short -> time domain
    g = CwtPlan(16384, 1, fs, np.geomspace(391.0, 4.4, 66)).scale_info()
    assert list(g["method"][:4]) == [1, 1, 1, 1] and np.all(g["method"][6:] == 0)
    f2 = np.geomspace(300.0, 3.0, 40)
    for gamma, beta in GB_PAIRS:
        q = CwtPlan(65536, 1, fs, f2, gamma=gamma, beta=beta)
        m, ln = q.scale_info()["method"], q.scale_info()["length"]
        si2, lv = q.scale_info(), q.debug_levels()
        if (gamma, beta) in ((3, 8), (4, 30)):
            assert (m == _lib.SCALE_SPECTRAL).sum() >= 36, (gamma, beta)
            assert not si2["theta_neg"][m == _lib.SCALE_SPECTRAL].any() and all(l["band_shift"] == 0 for l in lv)
        elif (gamma, beta) in ((3, 4), (2, 8)):
            spec = m == _lib.SCALE_SPECTRAL
            assert spec.sum() >= 23 and ln[~spec].max() <= 256
            assert np.all(m[~spec] == np.where(ln[~spec] <= 48, _lib.SCALE_DIRECT, _lib.SCALE_BLOCKCONV))
            assert np.all(si2["theta_neg"][spec] > 0) and all(l["band_shift"] > 0 for l in lv)
            # the band fits the level: [-theta_neg, theta_hi] inside [-shift, 256 - shift) bins
            for l in lv:
                mine = spec & (si2["decimation"] == l["decimation"])
                delta = 2 * np.pi / (256 * l["decimation"])
                assert np.all(si2["theta_neg"][mine] <= l["band_shift"] * delta * (1 + 1e-12))
                assert np.all(si2["theta_hi"][mine] <= (256 - l["band_shift"]) * delta * (1 + 1e-12))
                assert l["band_shift"] % max(1, l["decimation"] // 16) == 0
            # levels of one decimation share x_R, hence the shift
            by_r = {}
            for l in lv:
                assert by_r.setdefault(l["decimation"], l["band_shift"]) == l["band_shift"]
        else:
            assert not (m == _lib.SCALE_SPECTRAL).any(), (gamma, beta)
            # by kernel length: time domain, overlap-save blocks, one FFT of the whole segment (planner.h)
            assert np.all(m[ln <= 48] == _lib.SCALE_DIRECT)
            assert np.all(m[(ln > 48) & (ln <= 2560)] == _lib.SCALE_BLOCKCONV) and (ln > 256).any()
            assert np.all(m[ln > 2560] == _lib.SCALE_FULLBAND)
    # a looser tolerance is the caller's to ask for
    q = CwtPlan(65536, 1, fs, f2, gamma=2, beta=8, band_eps=1e-5)
    assert (q.scale_info()["method"] == _lib.SCALE_SPECTRAL).sum() >= 30


def test_level_designs_side_by_side_equal_one_after_the_other(option):
    """The interpolation designs of a plan's levels run on up to eight threads (planner.cpp): same tables, bounds and
    demodulation bins as one after the other.  (The design caches are per process: beta differs in its last digits.)"""
    from ghost_amd.engine import CwtPlan
    plans = []
    f = np.geomspace(480.0, 1.1, 190)
    for threads in (1, 8):
        option("plan_threads", threads)
        beta = 21.0 * (1.0 + 4e-14 * threads)        # a fresh cache key, the same design to twelve digits
        plans.append(CwtPlan(18000000, 4, 30000.0, f, gamma=3.0, beta=beta))
    a, b = (p.debug_interp() for p in plans)
    np.testing.assert_array_equal(a["demod"], b["demod"])
    made = 0
    for la, lb in zip(a["levels"], b["levels"]):
        assert (la is None) == (lb is None)
        if la is None:
            continue
        made += 1
        assert la["q"] == lb["q"] and la["factor"] == lb["factor"]
        np.testing.assert_allclose(la["coef"], lb["coef"], rtol=0, atol=1e-6)
        # (the bound is taken with the float32-rounded tables: a coefficient that rounds the other way moves a bound of
        # 6e-8 by up to 15 %, between two runs one after the other just as well)
        assert abs(la["err_bound"] - lb["err_bound"]) <= 0.3 * la["err_bound"]
    assert made >= 6
    np.testing.assert_array_equal(plans[0].scale_info()["decimation"], plans[1].scale_info()["decimation"])


def test_exact_precision_plans_no_decimated_scale():
    """precision='exact' (ghostcwt.h: GCWT_PRECISION_EXACT): the default wavelet's scales all go through the block
    convolution or the full-band path; 'high' stays the default; other values are refused."""
    from ghost_amd.engine import CwtPlan
    f = np.geomspace(200.0, 2.0, 100)
    p = CwtPlan(1000000, 8, 1000.0, f, precision="exact")
    m, ln = p.scale_info()["method"], p.scale_info()["length"]
    assert p.info["n_spectral"] == 0 and p.info["n_direct"] == 0
    assert np.all(m[ln <= 1024] == _lib.SCALE_BLOCKCONV) and np.all(m[ln > 1024] == _lib.SCALE_FULLBAND)
    assert all(g["hop"] + g["back"] + 256 <= 4096 and g["back"] >= 256 for g in p.debug_blockconv())   # faded block edges
    assert CwtPlan(1000000, 8, 1000.0, f).info["n_spectral"] == 100
    with pytest.raises(ValueError):
        CwtPlan(1000, 1, 1000.0, f[:3], precision="double")


def test_blockconv_planning():
    """Kernels no decimated band holds, by length: time domain up to 48 taps, overlap-save blocks up to 2560, one
    FFT per segment beyond (planner.h).  The block scales are grouped by length; every scale of a group is whole
    inside the group's window: `back` samples before an output sample, hop + ahead <= 4096 - back."""
    from ghost_amd.engine import CwtPlan
    f = np.geomspace(200.0, 2.0, 100)
    p = CwtPlan(1000000, 128, 1000.0, f, gamma=3, beta=2)
    si, info = p.scale_info(), p.info
    assert info["n_spectral"] == 0 and info["n_fullband"] == 0 and info["n_blockconv"] + info["n_direct"] == 100
    groups = p.debug_blockconv()
    assert 2 <= len(groups) <= 8
    seen = []
    for g in groups:
        ln = si["length"][g["scales"]]
        assert np.all(np.diff(ln) >= 0) and np.all(si["method"][g["scales"]] == _lib.SCALE_BLOCKCONV)
        behind, ahead = ln - 1 - (ln - 1) // 2, (ln - 1) // 2
        assert g["back"] % 64 == 0 and g["hop"] % 64 == 0 and g["hop"] >= 1536
        assert np.all(behind <= g["back"]) and np.all(g["back"] + g["hop"] + ahead <= 4096)
        seen += g["scales"]
    assert sorted(seen) == np.flatnonzero(si["method"] == _lib.SCALE_BLOCKCONV).tolist()
    assert [len(g["scales"]) for g in groups][0] > [len(g["scales"]) for g in groups][-1]     # short kernels share more
    # the longest kernels of Morse(3, 5) at this shape stay on the full-band path
    q = CwtPlan(1000000, 128, 1000.0, f, gamma=3, beta=5)
    ln, m = q.scale_info()["length"], q.scale_info()["method"]
    assert ln.max() > 2560 and np.all(m[ln > 2560] == _lib.SCALE_FULLBAND) and np.all(m[(ln > 48) & (ln <= 2560)] == _lib.SCALE_BLOCKCONV)
    # a recording in 300-sample epochs: a block per epoch would be mostly padding, the time domain keeps its 256 taps
    eb = np.array([[i * 400, i * 400 + 300] for i in range(500)])
    r = CwtPlan(200000, 2, 1000.0, np.array([300.0, 100.0, 30.0, 12.0]), gamma=3, beta=2, epoch_bounds=eb)
    ln, m = r.scale_info()["length"], r.scale_info()["method"]
    assert np.all(m[ln <= 256] == _lib.SCALE_DIRECT) and (ln > 100).any()


def test_decimated_model_for_other_wavelets(golden):
    """Planner decisions + exact bank, float64, against the reference-made G11."""
    from decimated_model import cwt_decimated
    from conftest import rel_err
    g = golden("g11_gamma_beta.npz")
    fs, x, cols = float(g["fs"]), g["x"], g["cols"]
    for gamma, beta in GB_PAIRS:
        tag = "g%d_b%d" % (gamma, beta)
        c = cwt_decimated(x, fs, g["frequencies"], gamma=gamma, beta=beta)
        assert rel_err(c[:, cols], g["complex_cols_" + tag]).max() < 1e-6, tag


def test_time_blocks_tile_the_epochs():
    """Epochs that need a longer FFT than allowed are cut into overlapping time blocks."""
    f = [200.0, 77.0, 20.0]
    p = CwtPlan(40000, 1, 1000.0, f, max_fft_log2=13)
    seg = p.segments()
    assert len(seg) > 4 and all(s[2] == 8192 for s in seg)
    assert seg[0][0] == 0 and seg[-1][1] == 40000
    assert all(a[1] == b[0] for a, b in zip(seg, seg[1:]))
    # whole epochs when they fit; two epochs -> independent segment lists
    p = CwtPlan(40000, 1, 1000.0, f, epoch_bounds=[[0, 5000], [5000, 40000]], max_fft_log2=14)
    seg = p.segments()
    assert seg[0][:2] == (0, 5000) and seg[1][0] == 5000 and seg[-1][1] == 40000
    assert len(CwtPlan(40000, 1, 1000.0, f).segments()) == 1
    with pytest.raises(GhostCwtError):
        CwtPlan(40000, 1, 1000.0, f, max_fft_log2=9)
    with pytest.raises(GhostCwtError):          # kernel of 6974 taps cannot live in 8192-sample blocks
        CwtPlan(100000, 1, 1000.0, [2.0], max_fft_log2=13)
    # the 2^22 limit now only bounds the block, not the recording
    p = CwtPlan(1 << 23, 1, 1000.0, [10.0])
    assert len(p.segments()) == 3
    # low frequencies relative to fs: decimation beyond 256
    p = CwtPlan(1 << 21, 1, 30000.0, [500.0, 6.0, 1.0])
    assert p.scale_info()["decimation"].tolist() == [32, 2048, 16384]


def test_segments_of_equal_fft_length_are_batched(option):
    """Many short epochs (or the time blocks of a long one) become extra 'channels' of one
    launch set, up to 16 at a time and within a workspace budget."""
    fs = 1000.0
    eb = [[i * 3000, i * 3000 + 2000 + 37 * (i % 5)] for i in range(40)]    # P = 4096 each
    p = CwtPlan(120000, 2, fs, [100.0, 40.0], epoch_bounds=eb)
    assert p.debug_batches() == [(0, 16), (16, 16), (32, 8)]
    lv0 = p.debug_levels(epoch=0)
    assert lv0 == p.debug_levels(epoch=15)                  # members share the level grids
    assert p.info["workspace_bytes"] > 16 * 2 * 4096 * 8
    # epochs of different FFT lengths never share a batch
    eb2 = [[0, 3000], [3000, 6000], [6000, 26000], [26000, 29000]]
    p2 = CwtPlan(30000, 1, fs, [100.0, 40.0], epoch_bounds=eb2)
    assert p2.debug_batches() == [(0, 2), (2, 1), (3, 1)]
    # time blocks of one long epoch batch too, until the budget says stop
    p3 = CwtPlan(200000, 1, fs, [100.0, 40.0], max_fft_log2=13)
    assert p3.debug_batches()[0] == (0, 16) and len(p3.segments()) > 16
    option("batch_bytes", 1)
    p4 = CwtPlan(200000, 1, fs, [100.0, 40.0], max_fft_log2=13)
    assert all(c == 1 for _, c in p4.debug_batches())


def test_output_adapter():
    from ghost_amd.formats import output_numpy_or_asa
    d = np.zeros((10, 3))
    assert output_numpy_or_asa(None, d) is d
    with pytest.raises(TypeError):
        output_numpy_or_asa(None, [1, 2, 3])
    with pytest.raises(TypeError):
        output_numpy_or_asa(None, d, output_type="pandas")
    with pytest.raises(ModuleNotFoundError):       # nelpy is not installed here
        output_numpy_or_asa(FakeASA(np.zeros((1, 10)), 10.0, [10]), d, output_type="asa")


def test_low_cut_lies_below_every_gain():
    """precision = high: every decimation cuts its slice of the spectrum below `low_cut` -- the planner promises
    that every scale reading that x_R answers with less than 2e-8 of its peak anywhere below it (checked here on
    the exact response, between the planner's own probe points too), that heavy-tailed wavelets (side lobes above
    that floor down to zero frequency) get no cut, and that precision = 'fast' has none."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd._lib import lib, check
    import ctypes as C
    f = np.geomspace(200.0, 2.0, 100)
    p = CwtPlan(1000000, 1, 1000.0, f)
    si, levels = p.scale_info(), p.debug_levels()
    assert all(lv["low_cut"] > 0 for lv in levels)
    om = 2 * np.pi * f / 1000.0
    assert np.all(si["theta_lo"] > 0.2 * om) and np.all(si["theta_lo"] < 0.35 * om)
    b = 1 << 24
    for lv in levels:
        members = np.flatnonzero(si["decimation"] == lv["decimation"])
        assert lv["low_cut"] <= si["theta_lo"][members].min() * (1 + 1e-12)
        a = np.unique(np.round(np.linspace(0, lv["low_cut"], 301) * b / (2 * np.pi)).astype(np.int64))
        a = a[a * 2 * np.pi / b <= lv["low_cut"]]
        for i in (members[0], members[-1]):
            g = np.zeros(a.size)
            check(lib.gcwt_debug_exact_gain(p._handle, int(i), a.ctypes.data_as(C.POINTER(C.c_int64)), b, a.size,
                                            g.ctypes.data_as(C.POINTER(C.c_double))))
            assert np.abs(g).max() <= 2.5e-8 * 2.0, (lv["decimation"], i, np.abs(g).max())   # peak gain is 2
    heavy = CwtPlan(1000000, 1, 1000.0, f, gamma=3, beta=4)
    assert all(lv["low_cut"] == 0 for lv in heavy.debug_levels()) and not heavy.scale_info()["theta_lo"].any()
    fast = CwtPlan(1000000, 1, 1000.0, f, precision="fast")
    assert all(lv["low_cut"] == 0 for lv in fast.debug_levels())
    with pytest.raises(ValueError):
        CwtPlan(1000, 1, 1000.0, [100.0], precision="double")


def test_long_mode_planning():
    """Kernels of millions of taps (below 0.13 Hz at 30 kHz, down to the reference's own floor, morse.py:93-106): the
    default plan moves to FFTs of 2^23 / 2^24 points by itself, keeps 2^22 for everything that fits it, and says why
    when a plan cannot be served that way."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd._lib import GhostCwtError
    fs, n = 30000.0, 18000000
    ordinary = CwtPlan(n, 1, fs, np.geomspace(500.0, 1.0, 20))
    assert all(s[2] == 1 << 22 for s in ordinary.segments())                   # config 5 is untouched
    deep = CwtPlan(n, 1, fs, np.array([500.0, 1.0, 0.1165]))
    assert all(s[2] == 1 << 23 for s in deep.segments()) and deep.info["fft_length"] == 1 << 23
    segs = deep.segments()
    assert segs[0][0] == 0 and segs[-1][1] == n and all(a[1] == b[0] for a, b in zip(segs[:-1], segs[1:]))
    assert deep.scale_info()["length"][-1] > 3500000 and deep.info["n_spectral"] == 3
    forced = CwtPlan(n, 1, fs, np.array([500.0, 0.1165]), max_fft_log2=24)
    assert all(s[2] == 1 << 24 for s in forced.segments())
    with pytest.raises(GhostCwtError, match="precision = high"):
        CwtPlan(n, 1, fs, np.array([500.0, 0.1165]), precision="fast")
    with pytest.raises(GhostCwtError, match="R >= 2"):                           # a scale at R = 4 beside a 3.6 M-tap kernel
        CwtPlan(n, 1, fs, np.array([3000.0, 0.1165]), max_fft_log2=24)
    with pytest.raises(GhostCwtError):
        CwtPlan(n, 1, fs, np.array([500.0]), max_fft_log2=25)


def test_named_options_replace_the_environment(monkeypatch):
    """The product library reads no GHOSTCWT_* variable beyond its two budgets: kernel and layout choices are
    named options (gcwt_debug_set_option), accuracy-changing ones exist in the measure build only, unknown names
    are refused, and an old caller's struct (garbage in the new fields) is told so."""
    import subprocess
    from ghost_amd import _lib
    from ghost_amd.engine import CwtPlan, set_option
    from ghost_amd._lib import GhostCwtError
    f = np.geomspace(190.0, 3.0, 41)
    n0 = CwtPlan(30000, 1, 1000.0, f).info["n_levels"]
    monkeypatch.setenv("GHOSTCWT_SPLIT_LEVELS", "1")             # the environment alone changes nothing ...
    assert CwtPlan(30000, 1, 1000.0, f).info["n_levels"] == n0
    try:
        set_option("split_levels", 1)                            # ... the named option does
        assert CwtPlan(30000, 1, 1000.0, f).info["n_levels"] > n0
    finally:
        set_option("split_levels", None)
    assert CwtPlan(30000, 1, 1000.0, f).info["n_levels"] == n0
    with pytest.raises(GhostCwtError):
        set_option("no_such_switch", 1)
    if not _lib.lib.gcwt_debug_measure_build():
        for name in ("halo_margin", "interp_q", "interp_min_r", "prune_inputs", "synth_drop_stores"):
            with pytest.raises(GhostCwtError) as ei:
                set_option(name, 0)
            assert ei.value.code == _lib.ERR_UNSUPPORTED
        names = subprocess.run(["strings", _lib.LIB_PATH], capture_output=True, text=True).stdout
        assert len([l for l in names.splitlines() if l.startswith("GHOSTCWT_") and len(l) > 9]) == 0
    p = _lib.Params()
    p.n_samples, p.n_channels, p.n_freqs, p.fs, p.gamma, p.beta = 1000, 1, 1, 1000.0, 3.0, 20.0
    fr = (C_double * 1)(100.0)
    p.freqs_hz = fr
    p.reserved0 = 7
    h = C_void_p()
    assert _lib.lib.gcwt_plan_create(C_byref(h), C_byref(p)) == _lib.ERR_INVALID
    assert b"reserved0" in _lib.lib.gcwt_last_error()


from ctypes import byref as C_byref, c_double as C_double, c_void_p as C_void_p   # noqa: E402


def test_pinned_result_pool_falls_back_without_a_device():
    """ghost_amd.hostmem hands out page-locked arrays when it can; without a GPU (or over its limit) it returns None and
    callers allocate pageable memory instead -- nothing raises, nothing is kept."""
    from ghost_amd import hostmem
    from ghost_amd.engine import device_count
    a = hostmem.empty((4, 1000), np.float64)
    if device_count() == 0:
        assert a is None
    else:
        assert a.shape == (4, 1000) and hostmem.is_pinned(a) and hostmem.is_pinned(a[1:3, ::2])
        del a
    old = hostmem.limit_bytes
    try:
        hostmem.limit_bytes = 100
        assert hostmem.empty((1000,), np.float32) is None
    finally:
        hostmem.limit_bytes = old
    assert not hostmem.is_pinned(np.zeros(3))
    assert hostmem.empty((0, 5), np.float32) is None
    hostmem.trim()


def test_devices_list_makes_one_plan_per_slot_and_needs_a_device():
    """``devices=[...]`` (ghost_amd/multi.py): the channel blocks are ``dist.shard_channels``' (contiguous, sizes differ
    by one at most), a slot never stays empty, planning is host work -- and executing without a GPU fails as loudly as
    one plan does (no CPU path)."""
    from ghost_amd.multi import ShardedPlan
    from ghost_amd.wave import ContinuousWaveletTransform
    f = np.geomspace(200.0, 5.0, 12)
    sp = ShardedPlan(20000, 10, 1000.0, f, [0, 1, 2, 3], output="power")
    assert sp.blocks == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [p.n_channels for p in sp.plans] == [3, 3, 2, 2] and sp.out_shape == (10, 12, 20000)
    assert sp.info["workspace_bytes"] == sum(p.info["workspace_bytes"] for p in sp.plans)
    sp.close()
    sp = ShardedPlan(20000, 2, 1000.0, f, [0, 1, 2, 3])
    assert sp.devices == [0, 1] and sp.blocks == [(0, 1), (1, 2)]
    x = np.zeros((2, 20000), np.float32)
    if not _lib_has_device():
        with pytest.raises(GhostCwtError) as e:
            sp.execute_resident(x)
        assert e.value.code == _lib.ERR_NO_DEVICE
    sp.close()
    cwt = ContinuousWaveletTransform()
    for bad in (dict(devices=[]), dict(devices=[-1, 0]), dict(devices=[0, 1], device=0)):
        with pytest.raises(ValueError):
            cwt.transform(x, fs=1000.0, multichannel=True, **bad)


def _lib_has_device():
    import ctypes
    n = ctypes.c_int(0)
    return _lib.lib.gcwt_device_count(ctypes.byref(n)) == 0 and n.value > 0


def test_precision_report_says_where_the_detector_does_not_look():
    """precision='auto' watches the decimated path's float32 stages through band sums made in the forward row pass;
    plans whose segments take FFTs of 2^23 / 2^24 points (kernels of millions of taps, combined from interleaved
    transforms) make none: the report says so ("watched": False) and the public call logs it, instead of a silence
    that reads as "nothing was at risk" (the reference is float64 end to end: transforms.py:142-143)."""
    long_plan = CwtPlan(1 << 20, 1, 30000.0, [0.13])
    assert long_plan.segments()[0][2] == 1 << 23
    assert long_plan.precision_report()["watched"] is False
    plain = CwtPlan(1 << 20, 1, 1000.0, [10.0, 5.0])
    rep = plain.precision_report()
    assert rep["watched"] is True and rep["rerouted"] == 0
    from ghost_amd.multi import ShardedPlan
    sp = ShardedPlan(1 << 20, 2, 30000.0, [0.13], [0, 0])
    assert sp.precision_report()["watched"] is False
    sp.close()
