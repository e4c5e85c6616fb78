"""Synthetic multichannel LFP used by the tests and by bench.py.

Per channel c: unit-variance pink (1/f power) noise from a fixed seed, an 8 Hz
rhythm with a channel-dependent phase and 40 Hz bursts; cast to float32.  This
is workload data, not reference code.
"""
import numpy as np

__all__ = ["lfp_channel", "lfp", "power_law_noise", "spectrum_class", "SPECTRUM_CLASSES"]


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


def power_law_noise(n_samples, exponent, seed=0):
    """Unit-variance Gaussian noise whose POWER spectrum falls like 1/f**exponent
    (1: pink, 2: brown / random walk, 3: steeper than most LFP); float64."""
    rng = np.random.default_rng(seed)
    spec = np.fft.rfft(rng.standard_normal(n_samples))
    k = np.arange(spec.size, dtype=np.float64)
    spec /= np.maximum(k, 1.0) ** (0.5 * exponent)
    spec[0] = 0.0
    x = np.fft.irfft(spec, n=n_samples)
    return x / x.std()


SPECTRUM_CLASSES = ("pink_lfp", "brown", "f3", "line30", "line100", "drift1000")


def spectrum_class(name, n_samples, fs=1000.0, channel=0):
    """Recordings that stress the dynamic range of the transforms (round 4): what real LFP looks
    like next to the pink workload data -- steep 1/f^2 .. 1/f^3 backgrounds, mains interference far
    above the signal, electrode drift.  float32, the engine's input type."""
    t = np.arange(n_samples) / fs
    base = lfp_channel(n_samples, fs, channel).astype(np.float64)
    if name == "pink_lfp":
        x = base
    elif name == "brown":
        x = power_law_noise(n_samples, 2.0, 77 + channel)
    elif name == "f3":
        x = power_law_noise(n_samples, 3.0, 78 + channel)
    elif name in ("line30", "line100"):
        x = base + float(name[4:]) * base.std() * np.sin(2 * np.pi * 60.0 * t + 0.3)
    elif name == "drift1000":
        x = base + 1000.0 * base.std() * np.sin(2 * np.pi * 0.5 * t + 1.0)
    else:
        raise ValueError(name)
    return x.astype(np.float32)
