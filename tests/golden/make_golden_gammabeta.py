"""G11: goldens for Morse wavelets other than the default (gamma, beta) = (3, 20), made
from the UNMODIFIED reference (build container only; the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        PYTHONPATH=/root/reference:/root/repo python3 tests/golden/make_golden_gammabeta.py

Drivers as in make_golden.py:
  D1  ContinuousWaveletTransform(wavelet=Morse(gamma=, beta=)).transform(...) -> amplitude
  D2  Morse(gamma=, beta=) + compute_lengths + fastconv_scipy -> complex coefficients
      (the objects transform() itself uses, ghost/wave/transforms.py:194-204)

Only inputs and outputs are stored -- no reference code.
"""
import logging
import os
import sys

import numpy as np

logging.disable(logging.WARNING)

import ghost as _ref_pkg                                          # refuses the alias package at this repo's root:
assert os.path.realpath(_ref_pkg.__file__).startswith("/root/reference/"), \
    "fixtures must come from the reference: put /root/reference FIRST on PYTHONPATH"
from ghost.wave import ContinuousWaveletTransform, Morse          # reference
from ghost.sigtools import fastconv_scipy                         # reference

from ghost_amd.synthetic import lfp_channel                       # this repo

HERE = os.path.dirname(os.path.abspath(__file__))
PAIRS = [(3, 8), (3, 4), (3, 2), (2, 8), (4, 30), (1, 5)]


def d2(x, fs, freqs, gamma, beta):
    x = np.asarray(x, dtype=np.float64)
    xc = x - np.mean(x)                          # transforms.py:142-143
    out = np.zeros((len(freqs), x.size), dtype=np.complex128)
    lengths = []
    for i, f in enumerate(freqs):
        m = Morse(gamma=gamma, beta=beta)
        m.fs = fs
        w = f / (fs / 2.0) * np.pi               # transforms.py:408-410
        m.norm_radian_freq = w
        L = int(m.compute_lengths(np.array([w]))[0])
        k, _ = m(L)
        lengths.append(L)
        out[i] = fastconv_scipy(xc, k)
    return out, np.array(lengths)


def main():
    fs = 1000.0
    n = 12000
    x32 = lfp_channel(n, fs, channel=21)
    x = x32.astype(np.float64)
    t = np.arange(n) / fs
    freqs = np.array([330.0, 210.0, 140.0, 77.0, 40.0, 23.0, 11.0, 6.5])
    cols = np.unique(np.concatenate([np.arange(96), np.arange(0, n, 29), np.arange(n - 96, n)]))
    arrays = dict(x=x32, fs=fs, frequencies=freqs, cols=cols, pairs=np.array(PAIRS, dtype=np.float64))
    for gamma, beta in PAIRS:
        tag = "g%d_b%d" % (gamma, beta)
        c, lengths = d2(x, fs, freqs, gamma, beta)
        arrays["lengths_" + tag] = lengths
        arrays["complex_cols_" + tag] = c[:, cols]
        arrays["rowmax_" + tag] = np.abs(c).max(axis=1)
        # public API, the wavelet handed to the constructor (transforms.py:42-46)
        cwt = ContinuousWaveletTransform(wavelet=Morse(gamma=gamma, beta=beta))
        cwt.transform(x, fs=fs, timestamps=t, freq_limits=[8, 300], voices_per_octave=4)
        f1 = cwt.frequencies.copy()
        c1, _ = d2(x, fs, f1, gamma, beta)
        assert np.allclose(np.abs(c1), cwt.amplitude, rtol=0, atol=1e-12)
        arrays["d1_frequencies_" + tag] = f1
        arrays["d1_amplitude_cols_" + tag] = cwt.amplitude[:, cols]
        arrays["d1_rowmax_" + tag] = cwt.amplitude.max(axis=1)
        print(tag, "lengths", lengths.tolist(), "D1 scales", f1.size, "%.2f..%.2f Hz" % (f1[-1], f1[0]))
    path = os.path.join(HERE, "g11_gamma_beta.npz")
    np.savez_compressed(path, **arrays)
    print("wrote g11_gamma_beta.npz", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    sys.exit(main())
