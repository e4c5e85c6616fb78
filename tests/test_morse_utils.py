"""The Morse utility layer against goldens generated from the
reference by tests/golden/make_golden_morse.py."""
import numpy as np
import pytest

from ghost_amd.wave import Morse
from ghost_amd.wave import morseutils as mu


@pytest.fixture(scope="module")
def g10(golden):
    return golden("g10_morse_utils.npz")


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def test_morsewave_orders_and_normalisations(g10):
    for i in range(int(g10["wave_n"])):
        n, ga, be, k, energy = g10["wave%d_args" % i]
        psi, psif = mu.morsewave(int(n), ga, be, g10["wave%d_freqs" % i], n_wavelets=int(k),
                                 normalization="energy" if energy else "bandpass")
        assert psi.shape == g10["wave%d_psi" % i].shape and psi.dtype == np.complex128
        assert _rel(psif, g10["wave%d_psif" % i]) < 1e-12, i
        assert _rel(psi, g10["wave%d_psi" % i]) < 1e-12, i
    # bandpass: the first-order spectrum peaks at 2 (sampled peak just below)
    _, psif = mu.morsewave(4096, 3.0, 20.0, 0.5)
    assert 1.999 < psif.max() <= 2.0
    with pytest.raises(ValueError):
        mu.morsewave(64, 3, 20, 0.5, normalization="unit")
    with pytest.raises(ValueError):
        mu.morsewave(64, 3, 20, 0.5, n_wavelets=0)
    with pytest.raises(ValueError):
        mu.morsewave(64, -1, 20, 0.5)


def test_morse_call_normalisations(g10):
    m = Morse(fs=1000.0)
    m.norm_radian_freq = 0.4
    for norm in ("bandpass", "energy"):
        psi, psif = m(300, normalization=norm)
        assert psi.shape == (300,) and psif.shape == (300,)
        assert _rel(psi, g10["call_%s_psi" % norm]) < 1e-12
        assert _rel(psif, g10["call_%s_psif" % norm]) < 1e-12
    with pytest.raises(ValueError):
        m(300, normalization="peak")
    with pytest.raises(ValueError):
        m(0)


def test_scalar_functions(g10):
    for j, (ga, be) in enumerate(g10["pairs"]):
        assert _rel(mu.morsefreq(ga, be, nout=4), g10["morsefreq4"][j]) < 1e-12
        assert mu.morsefreq(ga, be) == mu.morsefreq(ga, be, nout=2)[0]
        for p in range(4):
            assert _rel(mu.morsemom(p, ga, be, nout=4), g10["morsemom"][j, p]) < 1e-11
        assert _rel(mu.morsef(ga, be), g10["morsef"][j]) < 1e-14
        assert _rel(mu.morseafunc(ga, be), g10["afunc_bandpass"][j]) < 1e-14
        for o in (1, 2, 3):
            assert _rel(mu.morseafunc(ga, be, normalization="energy", order=o),
                        g10["afunc_energy"][j, o - 1]) < 1e-13
        assert _rel(mu.morselow(ga, be, 5, 1000), g10["morselow"][j]) < 1e-15
        assert mu.morsehigh(ga, be, 0.25) == g10["morsehigh_eta"][j]
    for k in range(4):
        assert _rel(mu.laguerre(g10["laguerre_x"], k, 2.5), g10["laguerre"][k]) < 1e-13
    with pytest.raises(ValueError):
        mu.morsefreq(3, 20, nout=5)
    with pytest.raises(ValueError):
        mu.morsemom(-1, 3, 20)
    with pytest.raises(ValueError):
        mu.morseafunc(3, 20, normalization="test")


def test_morsespace(g10):
    a = mu.morsespace(3.0, 20.0, 1000)
    assert a.shape == g10["space_default"].shape and _rel(a, g10["space_default"]) < 1e-13
    b = mu.morsespace(3.0, 20.0, 5000, high=2.0, eta=0.2, pack_num=3, low=0.01, density=4)
    assert b.shape == g10["space_opts"].shape and _rel(b, g10["space_opts"]) < 1e-13
    c = mu.morsespace(2.0, 8.0, 777, density=1)
    assert c.shape == g10["space_g2"].shape and _rel(c, g10["space_g2"]) < 1e-13
    assert np.all(np.diff(a) > 0)
    for bad in (dict(eta=2), dict(high=4), dict(pack_num=0), dict(low=-1), dict(density=0)):
        with pytest.raises(ValueError):
            mu.morsespace(3.0, 20.0, 1000, **bad)
    with pytest.raises(ValueError):
        mu.morsespace(3.0, 20.0, 1)


def test_morlet_kernels(g10):
    from ghost_amd.wave import Morlet
    assert _rel(Morlet().get_wavelet(), g10["morlet_default"]) < 1e-13
    mo = Morlet(w0=6, freq=40.0, fs=1000.0)
    assert mo.get_wavelet().shape == g10["morlet_40hz"].shape
    assert _rel(mo.get_wavelet(), g10["morlet_40hz"]) < 1e-13
    assert abs(mo.scale - float(g10["morlet_40hz_scale"])) < 1e-15
    mo.freq = 12.5
    mo.w0 = 7.0
    assert _rel(mo.get_wavelet(), g10["morlet_retuned"]) < 1e-13
    assert repr(mo) == "Morlet" and mo.copy().freq == 12.5
    for attr in ("fs", "w0", "freq"):
        with pytest.raises(ValueError):
            setattr(mo, attr, 0)
