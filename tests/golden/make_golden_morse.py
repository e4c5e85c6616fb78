"""Golden fixtures for the Morse utility layer, from the UNMODIFIED
reference.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        PYTHONPATH=/root/reference:/root/repo python3 tests/golden/make_golden_morse.py

Stores inputs and outputs of ghost.wave.morseutils functions -- no reference code.
"""
import os
import sys
import warnings

import numpy as np

warnings.simplefilter("ignore")

import ghost as _ref_pkg                                          # refuses the alias package at this repo's root:
assert os.path.realpath(_ref_pkg.__file__).startswith("/root/reference/"), \
    "fixtures must come from the reference: put /root/reference FIRST on PYTHONPATH"
from ghost.wave import Morse, Morlet            # reference
from ghost.wave import morseutils as mu         # reference

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = {}
    # morsewave(N, gamma, beta, freqs, n_wavelets, normalization)
    cases = [(256, 3.0, 20.0, [0.5, 1.0], 1, "bandpass"),
             (255, 3.0, 20.0, [0.3], 3, "bandpass"),
             (128, 3.0, 20.0, [0.8, -0.8], 2, "energy"),
             (64, 2.0, 8.0, [1.2], 1, "energy"),
             (101, 4.0, 2.5, [0.9, -0.4, 0.2], 2, "bandpass")]
    g["wave_n"] = len(cases)
    for i, (n, ga, be, fr, k, norm) in enumerate(cases):
        psi, psif = mu.morsewave(n, ga, be, np.array(fr), n_wavelets=k, normalization=norm)
        g["wave%d_args" % i] = np.array([n, ga, be, k, 0 if norm == "bandpass" else 1])
        g["wave%d_freqs" % i] = np.array(fr)
        g["wave%d_psi" % i] = psi
        g["wave%d_psif" % i] = psif
    # Morse.__call__ with both normalisations
    m = Morse(fs=1000.0)
    m.norm_radian_freq = 0.4
    for norm in ("bandpass", "energy"):
        psi, psif = m(300, normalization=norm)
        g["call_%s_psi" % norm] = psi
        g["call_%s_psif" % norm] = psif
    # scalars
    pairs = [(3.0, 20.0), (2.0, 8.0), (3.0, 1.5), (1.0, 4.0)]
    g["pairs"] = np.array(pairs)
    g["morsefreq4"] = np.array([mu.morsefreq(a, b, nout=4) for a, b in pairs])
    g["morsemom"] = np.array([[mu.morsemom(p, a, b, nout=4) for p in range(4)] for a, b in pairs])
    g["morsef"] = np.array([mu.morsef(a, b) for a, b in pairs])
    g["afunc_bandpass"] = np.array([mu.morseafunc(a, b) for a, b in pairs])
    g["afunc_energy"] = np.array([[mu.morseafunc(a, b, normalization="energy", order=o)
                                   for o in (1, 2, 3)] for a, b in pairs])
    g["morselow"] = np.array([mu.morselow(a, b, 5, 1000) for a, b in pairs])
    g["morsehigh_eta"] = np.array([mu.morsehigh(a, b, 0.25) for a, b in pairs])
    g["laguerre_x"] = np.linspace(0.0, 6.0, 25)
    g["laguerre"] = np.array([mu._laguerre(g["laguerre_x"], k, 2.5) for k in range(4)])
    g["space_default"] = mu.morsespace(3.0, 20.0, 1000)
    g["space_opts"] = mu.morsespace(3.0, 20.0, 5000, high=2.0, eta=0.2, pack_num=3, low=0.01,
                                    density=4)
    g["space_g2"] = mu.morsespace(2.0, 8.0, 777, density=1)
    # Morlet kernels (ghost/wave/morlet.py): default, and a 1 kHz / 40 Hz one re-tuned by setters
    g["morlet_default"] = Morlet().get_wavelet()
    mo = Morlet(w0=6, freq=40.0, fs=1000.0)
    g["morlet_40hz"] = mo.get_wavelet()
    g["morlet_40hz_scale"] = mo.scale
    mo.freq = 12.5
    mo.w0 = 7.0
    g["morlet_retuned"] = mo.get_wavelet()
    path = os.path.join(HERE, "g10_morse_utils.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    sys.exit(main())
