"""GPU FFT-convolution operators against scipy.signal.convolve -- the reference's own
tests of this layer (tests/test_convolution.py:6-42), same shapes."""
import numpy as np
import pytest
from scipy.fft import fft
from scipy.signal import convolve

pytestmark = pytest.mark.gpu


def _close(a, b):
    return np.abs(a - b).max() <= 2e-6 * np.abs(b).max() + 1e-6


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


def test_fastconv_freq_domain():
    from ghost_amd.sigtools import fastconv_freq_hip
    rng = np.random.default_rng(2)
    x = rng.random(10000)
    y = rng.random(1000)
    Y = fft(y, n=3000)
    for mode in ("full", "same", "valid"):
        assert _close(fastconv_freq_hip(x, Y, len(y), mode=mode), convolve(x, y, mode=mode)), mode
