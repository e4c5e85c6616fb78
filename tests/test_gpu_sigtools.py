"""GPU FFT-convolution operators against scipy.signal.convolve -- the reference's own
tests of this layer (tests/test_convolution.py:6-42), same shapes."""
import numpy as np
import pytest
from scipy.fft import fft
from scipy.signal import convolve

pytestmark = pytest.mark.gpu


def _close(a, b):
    """The gate metric of the float32 operators: the largest error against the result's peak."""
    return np.abs(a - b).max() <= 1e-5 * np.abs(b).max()


def test_fastconv_time_domain():
    from ghost_amd.sigtools import fastconv_hip
    rng = np.random.default_rng(1)
    x = rng.random(10000)
    y = rng.random(1000)
    for mode in ("full", "same", "valid"):
        ref = convolve(x, y, mode=mode)
        got = fastconv_hip(x, y, mode=mode)
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert _close(got, ref), mode
    # complex (Morse) kernel, long signal (two-pass FFT with P1 = 256), default mode 'same'
    from oracle import ghost_oracle as orc
    psi, _ = orc.morse_kernel(1395, orc.hz_to_rad(10.0, 1000.0))
    x = rng.standard_normal(700000)
    ref = orc.overlap_add_convolve(x, psi)
    got = fastconv_hip(x, psi)
    assert got.dtype == np.complex64 and _close(got, ref)
    with pytest.raises(ValueError):
        fastconv_hip(x[:10], np.ones(20), mode="valid")
    with pytest.raises(ValueError):
        fastconv_hip(x, psi, mode="circular")
    with pytest.raises(ValueError):
        fastconv_hip(x.reshape(2, -1), psi)


def test_reference_operator_names_run_on_the_device():
    """`from ghost.sigtools import fastconv_scipy, fastconv_fftw, ...` (the reference's own names,
    tests/test_convolution.py:1-2 of the reference imports exactly these) through the alias package: the
    reference's test shapes (tests/test_convolution.py:6-42, tests/test_hilbert.py:4-11), every mode, held to
    the reference's own assertion -- np.allclose at its defaults, element by element -- which float64 arithmetic
    on the device meets and float32 cannot (the edges of a 'full' convolution are 1e-3 of its peak)."""
    from ghost.sigtools import (fastconv_scipy, fastconv_fftw, fastconv_freq_scipy, fastconv_freq_fftw,
                                analytic_signal_fftw, analytic_signal_scipy, chirpz_dft)
    from scipy.signal import hilbert
    rng = np.random.default_rng(4)
    x, y = rng.random(10000), rng.random(1000)
    Y = fft(y, n=3000)
    for mode in ("full", "same", "valid"):
        ref = convolve(x, y, mode=mode)
        for fn in (fastconv_scipy, fastconv_fftw):
            got = fn(x, y, mode=mode, fft_length=2048)
            assert got.dtype == np.float64 and np.allclose(got, ref), (fn.__name__, mode)
        for fn in (fastconv_freq_scipy, fastconv_freq_fftw):
            got = fn(x, Y, len(y), mode=mode)
            assert got.dtype == np.float64 and np.allclose(got, ref), (fn.__name__, mode)
    assert np.allclose(fastconv_fftw(x, y, n_threads=4), convolve(x, y, mode="same"))
    # tests/test_hilbert.py: 30 000 x 60 samples (not a power of two: the chirp-z path in float64)
    sig = rng.random(30000 * 60)
    ref = hilbert(sig)
    for fn in (analytic_signal_fftw, analytic_signal_scipy):
        got = fn(sig)
        assert got.dtype == np.complex128 and np.allclose(got, ref), fn.__name__
    assert np.allclose(analytic_signal_fftw(x[:5000] - 0.5, fft_length=8192, n_threads=2), hilbert(x[:5000] - 0.5, N=8192)[:5000])
    for n in (777, 1024, 10007):
        got = chirpz_dft(x[:n])
        assert got.dtype == np.complex128 and np.allclose(got, fft(x[:n])), n
    # complex kernel, complex signal, and the float32 operators beside them at their own gate
    from ghost_amd.sigtools import fastconv_hip, chirpz_idft_hip
    from oracle import ghost_oracle as orc
    psi, _ = orc.morse_kernel(1395, orc.hz_to_rad(10.0, 1000.0))
    xs = rng.standard_normal(200000)
    ref = orc.overlap_add_convolve(xs, psi)
    hi = fastconv_hip(xs, psi, precision="high")
    assert hi.dtype == np.complex128 and np.allclose(hi, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
    assert _close(fastconv_hip(xs, psi), ref)
    z = rng.standard_normal(5000) + 1j * rng.standard_normal(5000)
    assert np.allclose(chirpz_idft_hip(z, precision="high"), np.fft.ifft(z), rtol=1e-10, atol=1e-13)
    with pytest.raises(ValueError):
        fastconv_hip(xs, psi, precision="exact")
    # a recording longer than one 2^24-point transform holds (9.3 minutes at 30 kHz): overlap-add over chunks of the
    # signal, the kernel's spectrum made once, as convolution.py:70-77 does -- every mode, real and complex kernels
    from scipy.signal import fftconvolve
    xl = rng.standard_normal((1 << 24) + 300001)
    for kern in (psi, y):
        for mode in ("same", "full", "valid"):
            ref = fftconvolve(xl, kern, mode=mode)
            got = fastconv_scipy(xl, kern, mode=mode)
            assert got.shape == ref.shape and got.dtype == ref.dtype
            assert np.allclose(got, ref, rtol=1e-9, atol=1e-11 * np.abs(ref).max()), (mode, kern.dtype)
            del ref, got


def test_fastconv_freq_domain():
    from ghost_amd.sigtools import fastconv_freq_hip
    rng = np.random.default_rng(2)
    x = rng.random(10000)
    y = rng.random(1000)
    Y = fft(y, n=3000)
    for mode in ("full", "same", "valid"):
        assert _close(fastconv_freq_hip(x, Y, len(y), mode=mode), convolve(x, y, mode=mode)), mode


def test_conv_plan_is_a_reusable_batched_operator():
    """gcwt_conv_plan_*: one plan, many executions; (C, N) batches; kernel_fd used as it is
    when it sits on the plan's FFT grid (convolution.py:218-402 with a power-of-two
    kernel_fd, chunked exactly as the reference chunks); device-resident in / out; signals
    beyond one 2^22-point FFT (overlap-save chunks); the reference's fft_length argument."""
    from ghost_amd.engine import DeviceBuffer
    from ghost_amd.sigtools import ConvPlan, fastconv_hip, fastconv_freq_hip
    rng = np.random.default_rng(4)
    C, n, m = 3, 50000, 777
    x = rng.standard_normal((C, n))
    k = rng.standard_normal(m) + 1j * rng.standard_normal(m)
    plan = ConvPlan(n, m, C)
    assert plan.fft_length == 65536 and plan.n_chunks == 1
    plan.set_kernel(k)
    for mode in ("full", "same", "valid"):
        got = plan.execute(x, mode=mode)
        for c in range(C):
            assert _close(got[c], convolve(x[c], k, mode=mode)), (mode, c)
    # a second kernel on the same plan, real this time -> float32 result
    kr = rng.random(m)
    got = plan.set_kernel(kr).execute(x)
    assert got.dtype == np.float32 and _close(got[1], convolve(x[1], kr, mode="same"))
    # the reference's chunking: fft_length 4096 -> chunks of 4096 - m + 1 samples
    small = ConvPlan(n, m, C, fft_length=4096).set_kernel(k)
    assert small.chunk == 4096 - m + 1 and small.n_chunks == -(-(n + m - 1) // small.chunk)
    got = small.execute(x, mode="full")
    assert _close(got[2], convolve(x[2], k, mode="full"))
    # kernel given by its DFT on a power-of-two grid: consumed directly
    Y = fft(kr, n=8192)
    for mode in ("full", "same", "valid"):
        assert _close(fastconv_freq_hip(x[0], Y, m, mode=mode), convolve(x[0], kr, mode=mode)), mode
    Yc = fft(k, n=4096)
    got = fastconv_freq_hip(x[0], Yc, m)
    assert got.dtype == np.complex64 and _close(got, convolve(x[0], k, mode="same"))
    # device-resident execution
    xb = DeviceBuffer(4 * C * n)
    xb.upload(x.astype(np.float32))
    ob = DeviceBuffer(8 * C * n)
    plan.set_kernel(k).execute_device(xb, ob, mode="same")
    dev = ob.download((C, n), np.complex64)
    np.testing.assert_array_equal(dev, plan.execute(x, mode="same"))
    # longer than one 2^22-point FFT: overlap-save chunks, kernel of 30 001 taps
    n2, m2 = 9000000, 30001
    x2 = rng.standard_normal(n2)
    k2 = rng.standard_normal(m2) * np.hanning(m2)
    got = fastconv_hip(x2, k2, mode="same")
    from oracle import ghost_oracle as orc
    ref = orc.overlap_add_convolve(x2, k2, mode="same").real
    assert got.shape == (n2,) and np.abs(got - ref).max() <= 5e-6 * np.abs(ref).max()
    # any fft_length the reference accepts is taken (rounded up to a power of two >= 4096: the
    # result does not depend on it); one shorter than the kernel is refused as the reference does
    odd = ConvPlan(n, m, C, fft_length=3000).set_kernel(k)
    assert odd.fft_length == 4096 and _close(odd.execute(x, mode="same")[0], convolve(x[0], k, mode="same"))
    assert ConvPlan(n, m, C, fft_length=5000).fft_length == 8192
    with pytest.raises(ValueError):
        fastconv_hip(x[0], k, fft_length=512)
    # one chunk holds the whole convolution whenever n + 2 (m - 1) fits the FFT
    tight = ConvPlan(65000, 500, 1)
    assert tight.fft_length == 131072 and tight.n_chunks == 1
    # shapes are checked before anything is reshaped
    with pytest.raises(ValueError, match="signals must have shape"):
        plan.execute(x[0])
    with pytest.raises(ValueError, match="signals must have shape"):
        plan.execute(x[:, :100])


def test_chirpz_dft():
    """tests/test_fourier.py:4-18 of the reference: even and odd (prime) lengths vs np.fft."""
    from ghost_amd.sigtools import chirpz_dft_hip, chirpz_idft_hip
    rng = np.random.default_rng(3)
    for n in (1009, 1010, 1, 2, 4096, 100003):
        x = rng.random(n)
        ref = np.fft.fft(x)
        got = chirpz_dft_hip(x)
        assert got.dtype == np.complex64 and got.shape == ref.shape
        assert np.abs(got - ref).max() <= 3e-6 * np.abs(ref).max(), n
    z = rng.standard_normal(5003) + 1j * rng.standard_normal(5003)
    ref = np.fft.fft(z)
    assert np.abs(chirpz_dft_hip(z) - ref).max() <= 3e-6 * np.abs(ref).max()
    back = chirpz_idft_hip(ref)
    assert np.abs(back - z).max() <= 3e-6 * np.abs(z).max()
    with pytest.raises(ValueError):
        chirpz_dft_hip(np.zeros((2, 8)))


def test_analytic_signal():
    """tests/test_hilbert.py:4-11 of the reference (30 kHz x 60 s of uniform noise) against
    scipy.signal.hilbert, plus padded, odd and tiny lengths and a large DC offset."""
    from scipy.signal import hilbert
    from ghost_amd.sigtools import analytic_signal_hip
    rng = np.random.default_rng(4)
    x = rng.random(30000 * 60)
    ref = hilbert(x)
    got = analytic_signal_hip(x)
    assert got.dtype == np.complex64 and got.shape == ref.shape
    # the real part is the input itself; the Hilbert part carries the arithmetic
    assert np.abs(got.real - x).max() <= 2e-6
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    for n, f in ((1000, None), (1001, None), (1000, 1500), (1001, 2003), (1, None), (2, None),
                 (3, None), (50000, 65536)):
        x = rng.standard_normal(n) + 1000.0
        ref = hilbert(x, N=f)[:n]
        got = analytic_signal_hip(x, fft_length=f)
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref - 1000.0).max() + 1e-3, (n, f)
    with pytest.raises(ValueError):
        analytic_signal_hip(x.astype(complex))
    with pytest.raises(ValueError):
        analytic_signal_hip(np.zeros(0))
    with pytest.raises(ValueError):
        analytic_signal_hip(np.zeros((4, 4)))
    with pytest.raises(ValueError):
        analytic_signal_hip(np.zeros(8), fft_length=4)


def test_morlet_scale_through_fastconv():
    """One Morlet scale = the signal convolved with Morlet.get_wavelet() (the reference's
    Morlet is a kernel factory for fastconv_*; ghost/wave/morlet.py)."""
    from ghost_amd.sigtools import fastconv_hip
    from ghost_amd.wave import Morlet
    rng = np.random.default_rng(5)
    fs = 1000.0
    x = rng.standard_normal(50000) + np.sin(2 * np.pi * 40.0 * np.arange(50000) / fs)
    k = Morlet(freq=40.0, fs=fs).get_wavelet()
    ref = convolve(x, k, mode="same")
    got = fastconv_hip(x, k)
    assert got.dtype == np.complex64 and _close(got, ref)
