"""G13: the public call over a grid of shapes -- sampling rates, odd lengths, voices per octave
from 4 to 48, frequency limits that the reference clamps to the wavelet's bounds, timestamp
gaps that split the recording into epochs -- made from the UNMODIFIED reference (build
container only; the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        PYTHONPATH=/root/reference:/root/repo python3 tests/golden/make_golden_api_grid.py

Driver D1 of SURVEY.md 8c: ContinuousWaveletTransform().transform(x, fs=, timestamps=,
freq_limits=, voices_per_octave=) -> frequencies, amplitude (ghost/wave/transforms.py:59-231).
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
from ghost.wave import ContinuousWaveletTransform                 # reference

from ghost_amd.synthetic import lfp_channel                       # this repo

HERE = os.path.dirname(os.path.abspath(__file__))

# (tag, fs, n, freq_limits, voices_per_octave, gaps): gaps = sample indices after which 0.25 s of
# recording are missing (the timestamps jump, preprocessing.py:78-114 cuts an epoch there)
CASES = [
    ("a", 1000.0, 5000, [10, 200], 4, []),
    ("b", 1000.0, 12345, [1, 600], 16, []),             # both limits clamped (transforms.py:412-434)
    ("c", 1250.0, 8191, [20, 300], 48, []),
    ("d", 30000.0, 40000, [100, 9000], 8, []),
    ("e", 200.0, 6001, [2, 60], 10, [2500]),
    ("f", 1000.0, 20000, [15, 250], 12, [7000, 13001]),
]


def main():
    arrays = {"tags": np.array([c[0] for c in CASES])}
    for k, (tag, fs, n, limits, v, gaps) in enumerate(CASES):
        x32 = lfp_channel(n, fs, channel=40 + k)
        t = np.arange(n) / fs
        for g in gaps:
            t[g:] += 0.25
        cwt = ContinuousWaveletTransform()
        cwt.transform(x32.astype(np.float64), fs=fs, timestamps=t, freq_limits=list(limits), voices_per_octave=v)
        f, a = cwt.frequencies.copy(), cwt.amplitude
        cols = np.unique(np.concatenate([np.arange(64), np.arange(0, n, 37), np.arange(n - 64, n)] +
                                        [np.arange(max(0, g - 48), min(n, g + 48)) for g in gaps]))
        arrays.update({"x_" + tag: x32, "t_" + tag: t, "fs_" + tag: fs, "limits_" + tag: np.array(limits, dtype=np.float64),
                       "voices_" + tag: v, "frequencies_" + tag: f, "cols_" + tag: cols,
                       "amplitude_cols_" + tag: a[:, cols], "rowmax_" + tag: a.max(axis=1)})
        print(tag, "fs", fs, "n", n, "scales", f.size, "%.3f..%.3f Hz" % (f[-1], f[0]), "epochs", len(gaps) + 1)
    path = os.path.join(HERE, "g13_api_grid.npz")
    np.savez_compressed(path, **arrays)
    print("wrote g13_api_grid.npz", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    sys.exit(main())
