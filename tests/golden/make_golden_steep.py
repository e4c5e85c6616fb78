"""G14: recordings whose spectrum is far from flat -- 1/f^3 noise with an offset, 1/f^2 noise, LFP with 60 Hz at 30 x
its spread -- through the UNMODIFIED reference (build container only; the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        PYTHONPATH=/root/reference:/root/repo python3 tests/golden/make_golden_steep.py

Both drivers of SURVEY.md 8c: D2, the inner loop of transform() -- Morse(), compute_lengths, the kernel, fastconv_scipy on
the mean-removed float64 copy (ghost/wave/transforms.py:142-143, :187-204) -- for the complex coefficients, and D1, the
public call, for the amplitude.  These are the inputs on which a float32 front end loses a quiet band's low bits (round
4); the fixture pins that the oracle restates the reference on them too, and gives the GPU test reference values that
do not come from this repository's own code.  Only inputs and outputs are stored -- no reference code.
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

from ghost_amd.synthetic import power_law_noise, spectrum_class   # this repo (workload data)

HERE = os.path.dirname(os.path.abspath(__file__))
FS, N = 1000.0, 32768
FREQS = np.array([200.0, 120.0, 61.0, 33.0, 17.0, 8.7])


def inner_loop(x64, f_hz):
    xc = x64 - np.mean(x64)                                        # transforms.py:142-143
    om = f_hz / (FS / 2) * np.pi                                   # transforms.py:408-410
    m = Morse()
    m.fs = FS
    m.norm_radian_freq = om
    length = int(m.compute_lengths(np.array([om]))[0])
    kernel, _ = m(length)
    return fastconv_scipy(xc, kernel), length


def main():
    inputs = {
        "f3_offset": (power_law_noise(N, 3.0, 301) + 40.0).astype(np.float32),
        "brown": power_law_noise(N, 2.0, 302).astype(np.float32),
        "line30": spectrum_class("line30", N, FS, channel=9),
    }
    cols = np.unique(np.concatenate([np.arange(128), np.arange(0, N, 48), np.arange(N - 128, N)]))
    arrays = {"fs": FS, "frequencies": FREQS, "cols": cols, "names": np.array(list(inputs))}
    for name, x32 in inputs.items():
        x64 = x32.astype(np.float64)
        coeffs, lengths = [], []
        for f in FREQS:
            w, length = inner_loop(x64, f)
            coeffs.append(w)
            lengths.append(length)
        coeffs = np.array(coeffs)
        arrays.update({"x_" + name: x32, "complex_cols_" + name: coeffs[:, cols], "rowmax_" + name: np.abs(coeffs).max(axis=1),
                       "lengths_" + name: np.array(lengths)})
        if name == "f3_offset":                                    # the public call too, on the steepest one
            cwt = ContinuousWaveletTransform()
            cwt.transform(x64, fs=FS, timestamps=np.arange(N) / FS, freq_limits=[9, 200], voices_per_octave=4)
            arrays.update({"api_frequencies_" + name: cwt.frequencies.copy(), "api_amplitude_cols_" + name: cwt.amplitude[:, cols]})
        print(name, "x std %.3g" % x64.std(), "row max / std", np.round(np.abs(coeffs).max(axis=1) / x64.std(), 5))
    path = os.path.join(HERE, "g14_steep.npz")
    np.savez_compressed(path, **arrays)
    print("wrote g14_steep.npz", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
