"""Synthetic multichannel LFP used by the tests and by bench.py.

Per channel c: unit-variance pink (1/f power) noise from a fixed seed, an 8 Hz
rhythm with a channel-dependent phase and 40 Hz bursts; cast to float32.  This
is workload data, not reference code.
"""
import numpy as np

__all__ = ["lfp_channel", "lfp"]


def lfp_channel(n_samples, fs=1000.0, channel=0, seed=1234):
    rng = np.random.default_rng(seed + channel)
    white = rng.standard_normal(n_samples)
    spec = np.fft.rfft(white)
    k = np.arange(spec.size)
    spec /= np.sqrt(np.maximum(k, 1))
    pink = np.fft.irfft(spec, n=n_samples)
    pink /= pink.std()
    t = np.arange(n_samples) / fs
    phase = 2 * np.pi * rng.random()
    burst = (np.sin(2 * np.pi * 0.5 * t + phase) > 0.6).astype(np.float64)
    x = pink + 0.5 * np.sin(2 * np.pi * 8 * t + phase) \
        + 0.2 * np.sin(2 * np.pi * 40 * t) * burst
    return x.astype(np.float32)


def lfp(n_channels, n_samples, fs=1000.0, seed=1234):
    """(n_channels, n_samples) float32, C-contiguous."""
    out = np.empty((n_channels, n_samples), dtype=np.float32)
    for c in range(n_channels):
        out[c] = lfp_channel(n_samples, fs, c, seed)
    return out
