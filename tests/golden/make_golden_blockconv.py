"""G15: heavy-tailed Morse wavelets with kernels of 60 .. 2400 taps -- the regime of the block convolution (round 4:
overlap-save over 4096-sample blocks) -- through the UNMODIFIED reference (build container only; the reference never
travels):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        PYTHONPATH=/root/reference:/root/repo python3 tests/golden/make_golden_blockconv.py

D2 of SURVEY.md 8c -- Morse(gamma, beta), compute_lengths, the kernel, fastconv_scipy per epoch on the mean-removed
float64 copy (ghost/wave/transforms.py:142-143, :187-204) -- for the complex coefficients of two epochs, and D1, the
public call, for the amplitude.  Gives the GPU tests of the new path reference values that do not come from this
repository's own code, and pins the oracle on these kernels.  Only inputs and outputs are stored -- no reference code.
"""
import logging
import os

import numpy as np

logging.disable(logging.WARNING)

import ghost as _ref_pkg                                          # refuses the alias package at this repo's root:
assert os.path.realpath(_ref_pkg.__file__).startswith("/root/reference/"), \
    "fixtures must come from the reference: put /root/reference FIRST on PYTHONPATH"
from ghost.wave import ContinuousWaveletTransform, Morse          # reference
from ghost.sigtools import fastconv_scipy                         # reference

from ghost_amd.synthetic import lfp_channel                       # this repo (workload data)

HERE = os.path.dirname(os.path.abspath(__file__))
FS, N = 1000.0, 40000
FREQS = np.array([150.0, 61.0, 27.0, 11.0, 5.3, 3.1])
EPOCHS = np.array([[0, 17000], [17011, N]])
PAIRS = [(3.0, 2.0), (1.0, 5.0), (3.0, 5.0)]


def inner_loop(x64, f_hz, gamma, beta):
    xc = x64 - np.mean(x64)                                        # transforms.py:142-143 (the global mean)
    om = f_hz / (FS / 2) * np.pi                                   # transforms.py:408-410
    m = Morse(gamma=gamma, beta=beta)
    m.fs = FS
    m.norm_radian_freq = om
    length = int(m.compute_lengths(np.array([om]))[0])
    kernel, _ = m(length)
    out = np.zeros(N, dtype=np.complex128)
    for a, b in EPOCHS:                                            # transforms.py:187-204: every epoch on its own
        out[a:b] = fastconv_scipy(xc[a:b], kernel)
    return out, length


def main():
    x32 = (lfp_channel(N, FS, 5) * 3.0 + 0.7).astype(np.float32)
    x64 = x32.astype(np.float64)
    cols = np.unique(np.concatenate([np.arange(160), np.arange(0, N, 37), np.arange(16900, 17100), np.arange(N - 160, N)]))
    arrays = {"fs": FS, "frequencies": FREQS, "cols": cols, "x": x32, "epochs": EPOCHS,
              "pairs": np.array(PAIRS)}
    for gamma, beta in PAIRS:
        tag = "%g_%g" % (gamma, beta)
        coeffs, lengths = [], []
        for f in FREQS:
            w, length = inner_loop(x64, f, gamma, beta)
            coeffs.append(w)
            lengths.append(length)
        coeffs = np.array(coeffs)
        arrays.update({"complex_cols_" + tag: coeffs[:, cols], "rowmax_" + tag: np.abs(coeffs).max(axis=1),
                       "lengths_" + tag: np.array(lengths)})
        print(tag, "lengths", lengths)
    cwt = ContinuousWaveletTransform(wavelet=Morse(gamma=3.0, beta=2.0))
    cwt.transform(x64[:17000], fs=FS, timestamps=np.arange(17000) / FS, freq_limits=[4, 120], voices_per_octave=4)
    arrays.update({"api_frequencies": cwt.frequencies.copy(), "api_amplitude_cols": cwt.amplitude[:, cols[cols < 17000]]})
    path = os.path.join(HERE, "g15_blockconv.npz")
    np.savez_compressed(path, **arrays)
    print("wrote g15_blockconv.npz", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
