"""NumPy model of the algorithm the HIP engine runs (test helper, float64).

Decimate-then-block formulation:
  X      = FFT_P(x - mean, zero padded)                      one per channel/epoch
  x_R    = IFFT_{P/R}(X[0:P/R]) / R                          analytic low-pass, rate fs/R
  XB_b   = FFT_B(x_R[b*hop - Lh : b*hop - Lh + B])           block spectra (circular index)
  y[R*m + r] = IFFT_B(XB_b * H_s(2 pi k/(B R)) * exp(2 pi i k r/(B R)))[m]
valid for Lh <= m < B - Lh.  Used on the CPU to check the maths and the planner
against the literal oracle before any kernel runs.
"""
import math

import numpy as np
from scipy.fft import fft, ifft

from oracle import ghost_oracle as orc


def band_edges(gamma, beta, eps):
    """u_lo < 1 < u_hi with Psi(u*w0)/Psi(w0) = eps (bisection on the log gain)."""
    w0c = beta / gamma  # w0**gamma

    def lg(u):
        return beta * math.log(u) - w0c * (u ** gamma - 1.0)

    target = math.log(eps)
    lo, hi = 1e-9, 1.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if lg(mid) < target:
            lo = mid
        else:
            hi = mid
    u_lo = lo
    lo, hi = 1.0, 64.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if lg(mid) < target:
            hi = mid
        else:
            lo = mid
    return u_lo, hi


def plan_scale(omega, length, p_big, B, gamma, beta, eps):
    """(method, R, Lh, hop) for one scale."""
    _, u_hi = band_edges(gamma, beta, eps)
    if u_hi * omega > math.pi:
        return ("direct", 1, 0, 0)
    r = 1
    while u_hi * omega * (2 * r) <= math.pi * 2 and 2 * r <= p_big // B:
        r *= 2
    # r is the largest power of two with u_hi*omega <= 2 pi / r
    lh = max(int(math.ceil(0.82 * length / (2.0 * r))) + 2, 16)
    hop = B - 2 * lh
    return ("spectral", r, lh, hop)


def cwt_decimated(x, fs, freqs_hz, epoch_bounds=None, gamma=3.0, beta=20.0,
                  B=256, eps=1e-9):
    x = np.asarray(x).squeeze().astype(np.float64)
    x = x - x.mean()
    n = x.size
    if epoch_bounds is None:
        epoch_bounds = np.array([[0, n]])
    freqs_hz = np.atleast_1d(np.asarray(freqs_hz, dtype=np.float64))
    omegas = orc.hz_to_rad(freqs_hz, fs)
    lengths = orc.morse_lengths(omegas, gamma, beta)
    out = np.zeros((len(freqs_hz), n), dtype=np.complex128)
    for start, stop in epoch_bounds:
        ne = stop - start
        p_big = max(1 << int(math.ceil(math.log2(ne + int(lengths.max())))), 2 * B)
        X = fft(x[start:stop], n=p_big)
        xr_cache = {}
        for i, (om, L) in enumerate(zip(omegas, lengths)):
            method, R, lh, hop = plan_scale(om, L, p_big, B, gamma, beta, eps)
            if method == "direct":
                psi, _ = orc.morse_kernel(L, om, gamma, beta)
                out[i, start:stop] = orc.overlap_add_convolve(x[start:stop], psi)
                continue
            assert hop > 0, (R, lh)
            M = p_big // R
            if R not in xr_cache:
                xr_cache[R] = ifft(X[:M]) / R
            xr = xr_cache[R]
            k = np.arange(B)
            H = orc.spectral_filter(2 * np.pi * k / (B * R), om, L, gamma, beta)
            tw = np.exp(2j * np.pi * np.outer(k, np.arange(R)) / (B * R))
            nblk = int(math.ceil(math.ceil(ne / R) / hop))
            y = np.zeros(R * nblk * hop, dtype=np.complex128)
            for b in range(nblk):
                idx = (b * hop - lh + np.arange(B)) % M
                XB = fft(xr[idx])
                blk = ifft((XB * H)[:, None] * tw, axis=0)        # [m, r]
                y[R * b * hop: R * (b + 1) * hop] = blk[lh:lh + hop].reshape(-1)
            out[i, start:stop] = y[:ne]
    return out
