"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Run in the build container only (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        PYTHONPATH=/root/reference:/root/repo python3 tests/golden/make_golden.py

Drivers:
  D1  public API  ContinuousWaveletTransform.transform(...)  -> frequencies, amplitude
  D2  inner loop  Morse + compute_lengths + fastconv_scipy   -> complex coefficients
      (line-for-line use of the same reference objects transform() uses at
      ghost/wave/transforms.py:194-204; needed because transform() discards
      the complex coefficients and cannot produce arbitrary frequency lists).

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
from ghost.wave import morseutils                                 # reference
from ghost.sigtools import fastconv_scipy                         # reference
from ghost.utils import get_contiguous_segments                   # reference

from ghost_amd.synthetic import lfp_channel                       # this repo

HERE = os.path.dirname(os.path.abspath(__file__))


def d2(x, fs, freqs, epoch_bounds=None):
    """Complex coefficients via the reference's own objects (driver D2)."""
    x = np.asarray(x, dtype=np.float64)
    xc = x - np.mean(x)                          # transforms.py:142-143
    if epoch_bounds is None:
        epoch_bounds = [[0, x.size]]
    out = np.zeros((len(freqs), x.size), dtype=np.complex128)
    lengths = []
    for i, f in enumerate(freqs):
        m = Morse()
        m.fs = fs
        w = f / (fs / 2.0) * np.pi               # transforms.py:408-410
        m.norm_radian_freq = w
        L = int(m.compute_lengths(np.array([w]))[0])
        k, _ = m(L)
        lengths.append(L)
        for s, e in epoch_bounds:
            out[i, s:e] = fastconv_scipy(xc[s:e], k)
    return out, np.array(lengths)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote", name, os.path.getsize(path), "bytes")


def main():
    fs = 1000.0

    # ---- G1: BASELINE config 1, 1 ch x 16384 @ 1 kHz, 32 scales, driver D1 + D2
    n = 16384
    x32 = lfp_channel(n, fs, channel=0)
    x = x32.astype(np.float64)
    t = np.arange(n) / fs
    cwt = ContinuousWaveletTransform()
    cwt.transform(x, fs=fs, timestamps=t, freq_limits=[5, 200], voices_per_octave=6)
    freqs = cwt.frequencies.copy()
    amp = cwt.amplitude
    assert freqs.size == 32
    cols = np.unique(np.concatenate([np.arange(256), np.arange(0, n, 64),
                                     np.arange(n - 256, n)]))
    cplx, lengths = d2(x, fs, freqs)
    assert np.allclose(np.abs(cplx), amp, rtol=0, atol=1e-13)
    save("g1_config1.npz", x=x32, fs=fs, frequencies=freqs, lengths=lengths,
         cols=cols, amplitude_cols=amp[:, cols], complex_cols=cplx[:, cols],
         amplitude_rowmax=amp.max(axis=1), amplitude_sum=amp.sum())

    # ---- G2: full complex output, N=2048, 8 scales with odd and even L
    n2 = 2048
    x2_32 = lfp_channel(n2, fs, channel=3)
    f2 = np.array([200.0, 140.0, 100.0, 77.0, 50.0, 40.0, 36.0])
    c2, l2 = d2(x2_32.astype(np.float64), fs, f2)
    assert (l2 % 2 == 0).any() and (l2 % 2 == 1).any()
    save("g2_complex_small.npz", x=x2_32, fs=fs, frequencies=f2, lengths=l2,
         coeffs=c2)

    # ---- G3: kernels psi, psif at matching omega
    g3 = {}
    for L, f in [(36, 391.0), (70, 200.0), (279, 50.0), (1163, 12.0), (1395, 10.0),
                 (40, 350.0)]:
        m = Morse()
        m.fs = fs
        w = f / (fs / 2.0) * np.pi
        m.norm_radian_freq = w
        Lc = int(m.compute_lengths(np.array([w]))[0])
        assert Lc == L, (Lc, L)
        psi, psif = m(L)
        g3["psi_%d" % L] = psi
        g3["psif_%d" % L] = psif
        g3["omega_%d" % L] = w
    save("g3_kernels.npz", **g3)

    # ---- G4: scalar known answers
    m = Morse()
    g4 = dict(
        morsefreq=morseutils.morsefreq(3, 20),
        morsehigh=morseutils.morsehigh(3, 20),
        bounds_16384=np.array(m.compute_freq_bounds(16384)),
        bounds_1e6=np.array(m.compute_freq_bounds(1000000)),
        bounds_18e6=np.array(m.compute_freq_bounds(18000000)),
        bounds_4096=np.array(m.compute_freq_bounds(4096)),
        len_freqs_hz=np.array([1, 2, 5.568, 10, 12, 50, 100, 200, 391, 500.0]),
    )
    m.fs = fs
    g4["lengths_1khz"] = m.compute_lengths(g4["len_freqs_hz"] / (fs / 2) * np.pi)
    f30 = np.array([1.0, 2.0, 200.0, 500.0])
    g4["len_freqs_30k"] = f30
    g4["lengths_30khz"] = m.compute_lengths(f30 / 15000.0 * np.pi)
    # default grid on 16384 samples
    cwt = ContinuousWaveletTransform()
    cwt.transform(x, fs=fs, timestamps=t)
    g4["default_grid_16384"] = cwt.frequencies.copy()
    # beta/gamma variants
    g4["morsefreq_g2_b8"] = morseutils.morsefreq(2, 8)
    g4["morsehigh_g2_b8"] = morseutils.morsehigh(2, 8)
    save("g4_scalars.npz", **g4)

    # ---- G4b: deterministic two-tone signal, D1 + D2
    nb = 4096
    tb = np.arange(nb) / fs
    xb = np.sin(2 * np.pi * 50 * tb) + 0.5 * np.sin(2 * np.pi * 12 * tb)
    cwt = ContinuousWaveletTransform()
    cwt.transform(xb, fs=fs, timestamps=tb, freq_limits=[10, 100], voices_per_octave=4)
    cb, lb = d2(xb, fs, np.array([50.0, 12.0]))
    save("g4b_two_tone.npz", frequencies=cwt.frequencies.copy(),
         amplitude_col2048=cwt.amplitude[:, 2048], amplitude_sum=cwt.amplitude.sum(),
         w_idx=np.array([0, 1000, 2048, 4095]),
         w50=cb[0, [0, 1000, 2048, 4095]], w12=cb[1, [0, 1000, 2048, 4095]],
         lengths=lb)

    # ---- G5: two epochs (6000 + 4000 samples, 10 s gap), driver D1 + D2
    n5 = 10000
    x5_32 = lfp_channel(n5, fs, channel=5)
    t5 = np.arange(n5) / fs
    t5[6000:] += 10.0
    cwt = ContinuousWaveletTransform()
    cwt.transform(x5_32.astype(np.float64), fs=fs, timestamps=t5)
    eb = get_contiguous_segments(t5, step=1 / fs, assume_sorted=False, index=True,
                                 inclusive=False)
    c5, l5 = d2(x5_32.astype(np.float64), fs, cwt.frequencies, eb)
    assert np.allclose(np.abs(c5), cwt.amplitude, rtol=0, atol=1e-13)
    cols5 = np.unique(np.concatenate([np.arange(0, n5, 16), np.arange(5900, 6100)]))
    save("g5_two_epochs.npz", x=x5_32, fs=fs, timestamps=t5, epoch_bounds=eb,
         frequencies=cwt.frequencies.copy(), lengths=l5, cols=cols5,
         complex_cols=c5[:, cols5], amplitude_rowmax=cwt.amplitude.max(axis=1))

    # ---- G6: near-Nyquist scales (closed form invalid), N=4096
    n6 = 4096
    x6_32 = lfp_channel(n6, fs, channel=6)
    f6 = np.array([391.0, 350.0, 320.0, 300.0, 280.0])
    c6, l6 = d2(x6_32.astype(np.float64), fs, f6)
    save("g6_near_nyquist.npz", x=x6_32, fs=fs, frequencies=f6, lengths=l6, coeffs=c6)

    # ---- G8: multichannel (each channel = one reference call), odd N, default grid
    n8 = 5001
    chans = np.stack([lfp_channel(n8, fs, channel=c) for c in (10, 11, 12)])
    t8 = np.arange(n8) / fs
    amps = []
    for c in range(3):
        cwt = ContinuousWaveletTransform()
        cwt.transform(chans[c].astype(np.float64), fs=fs, timestamps=t8,
                      freq_limits=[20, 250], voices_per_octave=4)
        amps.append(cwt.amplitude.copy())
    save("g8_multichannel.npz", x=chans, fs=fs, frequencies=cwt.frequencies.copy(),
         amplitude=np.stack(amps).astype(np.float32))

    # ---- G9: BASELINE config 2 shape, reduced: 100 scales geomspace(200,2),
    #      N=65536 (long enough for the 2 Hz kernel, L=6974), decimated columns
    n9 = 65536
    x9_32 = lfp_channel(n9, fs, channel=9)
    f9 = np.geomspace(200.0, 2.0, 100)
    c9, l9 = d2(x9_32.astype(np.float64), fs, f9)
    cols9 = np.unique(np.concatenate([np.arange(0, n9, 97), np.arange(64),
                                      np.arange(n9 - 64, n9)]))
    save("g9_config2_reduced.npz", x=x9_32, fs=fs, frequencies=f9, lengths=l9,
         cols=cols9, complex_cols=c9[:, cols9].astype(np.complex64),
         amplitude_rowmax=np.abs(c9).max(axis=1))


if __name__ == "__main__":
    sys.exit(main())
