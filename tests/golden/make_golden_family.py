"""G12: goldens for the other members of the Morse family -- orthogonal wavelets of higher
order and the 'energy' normalisation -- made from the UNMODIFIED reference (build container
only):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        PYTHONPATH=/root/reference:/root/repo python3 tests/golden/make_golden_family.py

transform() itself only ever uses the first 'bandpass' wavelet (ghost/wave/morse.py:84-91);
the rest of the family is reached through ``morsewave(..., n_wavelets=, normalization=)``
(ghost/wave/morseutils.py:22-91).  Driver: the reference's own ``morsewave`` for the kernels
(lengths from ``Morse.compute_lengths``) and its ``fastconv_scipy`` for the convolution,
i.e. the inner loop of transforms.py:194-204 with the kernel of another family member.
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
from ghost.wave import Morse                                      # reference
from ghost.wave.morseutils import morsewave                       # reference
from ghost.sigtools import fastconv_scipy                         # reference

from ghost_amd.synthetic import lfp_channel                       # this repo

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(3, 20, "bandpass", 3), (3, 20, "energy", 3), (2, 8, "bandpass", 2), (4, 30, "energy", 2)]


def main():
    fs, n = 1000.0, 9000
    x32 = lfp_channel(n, fs, channel=31)
    x = x32.astype(np.float64)
    xc = x - x.mean()
    freqs = np.array([310.0, 150.0, 40.0, 12.0, 5.0])
    cols = np.unique(np.concatenate([np.arange(80), np.arange(0, n, 23), np.arange(n - 80, n)]))
    arrays = dict(x=x32, fs=fs, frequencies=freqs, cols=cols,
                  cases=np.array([[g, b, norm == "energy", k] for g, b, norm, k in CASES], dtype=np.float64))
    for gamma, beta, norm, n_w in CASES:
        tag = "g%d_b%d_%s" % (gamma, beta, norm)
        m = Morse(gamma=gamma, beta=beta)
        m.fs = fs
        lengths = m.compute_lengths(freqs / (fs / 2.0) * np.pi).astype(int)
        coeffs = np.zeros((n_w, freqs.size, n), dtype=np.complex128)
        for i, f in enumerate(freqs):
            w = f / (fs / 2.0) * np.pi
            psi, psif = morsewave(int(lengths[i]), gamma, beta, w, n_wavelets=n_w, normalization=norm)
            for k in range(n_w):
                coeffs[k, i] = fastconv_scipy(xc, psi[:, 0, k])
            if i == 2:
                arrays["psi_" + tag] = psi[:, 0, :]
                arrays["psif_" + tag] = psif[:, 0, :]
        arrays["lengths_" + tag] = lengths
        arrays["complex_cols_" + tag] = coeffs[:, :, cols]
        arrays["rowmax_" + tag] = np.abs(coeffs).max(axis=2)
        print(tag, "lengths", lengths.tolist(), "row maxima order 0:", np.abs(coeffs[0]).max(axis=1))
    path = os.path.join(HERE, "g12_family.npz")
    np.savez_compressed(path, **arrays)
    print("wrote g12_family.npz", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    sys.exit(main())
