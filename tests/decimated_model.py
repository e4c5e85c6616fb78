"""NumPy model of the algorithm the HIP engine runs (test helper, float64).

Per scale the engine's planner picks one of three evaluations of the SAME quantity, the
convolution with the reference's L-tap kernel:

  spectral   X = FFT_P(x - mean, zero padded); x_R = IFFT_{P/R}(T_R X[0:P/R]) / R, T_R the level's low
             cut (precision = high: zero below theta_cut / 2, half a cosine up to one at theta_cut, below
             the band of every scale of the decimation; ``low_cut`` of ``debug_levels``)  -- for a level whose
             band starts `shift` bins of its 256-point grid below zero frequency (heavy-tailed
             wavelets), IFFT_{P/R}(X[-U : P/R - U]) with U = shift (P/R)/256 and every bin k
             below standing for the frequency k - shift;
             XB_b = FFT_B(x_R[b*hop - Lh : b*hop - Lh + B]) (circular index);
             y[R*m + r] = IFFT_B(XB_b * H_s(2 pi k/(B R)) * exp(2 pi i k r/(B R)))[m],
             kept for Lh <= m < B - Lh
  full-band  y = IFFT_P(X * H_s(2 pi k/P))[0:N_e]
  direct     the literal kernel, time domain

  interpolated (amplitude / power at R >= 16, csrc/synthi.hip): only q of the R phases go through
             the block transform, z[q m + p] = IFFT_B(XB_b * G_s * exp(2 pi i (k - k_c)(q m + p)/(B q)))[m]
             -- bins counted from the scale's demodulation bin k_c, real gain G_s -- and
             |y[I m' + rho]| = |sum_j c_rho[j] z[m' + j - 3]|, I = R/q, with the planner's 8-tap
             coefficient tables (tau = rho/I for odd kernel lengths, (rho - 1/2)/I for even ones)

with H_s the exact response of the reference's kernel (``exact_gain`` below, the
closed form csrc/morse_exact.h evaluates).  The decisions (method, R, halo, hop) are
taken from the planner itself (``CwtPlan.scale_info``), so this model checks on the CPU,
against the literal oracle, both the maths and what the planner decided.
"""
import math

import numpy as np
from scipy.fft import fft, ifft

from oracle import ghost_oracle as orc


def kept_bins(omega, length, gamma, beta, normalization="bandpass", order=0):
    """(j, A_j): the spectrum samples the reference kernel is built from
    (morseutils.py:117, :130-131, :178, :181-196), without the ones below 1e-18 of the
    largest."""
    _, psif = orc.morse_kernel(int(length), omega, gamma, beta, normalization, order)
    amp = psif[:round(int(length) / 2)]
    keep = np.abs(amp) > np.abs(amp).max() * 1e-18 if amp.size else np.zeros(0, bool)
    j = np.flatnonzero(keep)
    return j, amp[j]


def exact_gain(theta, omega, length, gamma=3.0, beta=20.0, normalization="bandpass", order=0):
    """G(theta), real: H = exp(-i theta d) G is the response of the reference's kernel;
    G(theta) = (1/L) sum_j A_j sin(L (theta_j - theta)/2) / sin((theta_j - theta)/2)."""
    L = int(length)
    j, amp = kept_bins(omega, L, gamma, beta, normalization, order)
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    d = 2 * np.pi * j[None, :] / L - theta[:, None]
    num, den = np.sin(L * d / 2), np.sin(d / 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(np.abs(den) < 1e-13, L * np.cos(L * d / 2) / np.cos(d / 2), num / den)
    return (ratio * amp[None, :]).sum(axis=1) / L


def exact_response(theta, omega, length, gamma=3.0, beta=20.0, normalization="bandpass", order=0):
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    d = (length - 1) / 2 - (length - 1) // 2
    return exact_gain(theta, omega, length, gamma, beta, normalization, order) * np.exp(-1j * theta * d)


def low_cut(theta_cut, p_big, m):
    """T_R on bins 0 .. m-1 of a P-point grid (csrc/kernels.hip: row_taper)."""
    if not theta_cut > 0:
        return np.ones(m)
    k1 = theta_cut * p_big / (2 * np.pi)
    k0 = 0.5 * k1
    if k1 - k0 < 1:
        return np.ones(m)
    u = np.clip((np.arange(m) - k0) / (k1 - k0), 0.0, 1.0)
    return 0.5 - 0.5 * np.cos(np.pi * u)


def cwt_decimated(x, fs, freqs_hz, epoch_bounds=None, gamma=3.0, beta=20.0, B=256, plan=None,
                  normalization="bandpass", order=0):
    """Model output, complex128 (S, N).  ``plan``: a CwtPlan for the same layout (made
    here when omitted; planning needs no GPU)."""
    from ghost_amd.engine import CwtPlan
    x = np.asarray(x).squeeze().astype(np.float64)
    x = x - x.mean()
    n = x.size
    if epoch_bounds is None:
        epoch_bounds = np.array([[0, n]])
    freqs_hz = np.atleast_1d(np.asarray(freqs_hz, dtype=np.float64))
    if plan is None:
        plan = CwtPlan(n, 1, fs, freqs_hz, gamma=gamma, beta=beta, epoch_bounds=epoch_bounds,
                       normalization=normalization, order=order)
    si = plan.scale_info()
    shift_of = {lv["decimation"]: lv["band_shift"] for lv in plan.debug_levels()}
    cut_of = {lv["decimation"]: lv["low_cut"] for lv in plan.debug_levels()}
    omegas = orc.hz_to_rad(freqs_hz, fs)
    lengths = orc.morse_lengths(omegas, gamma, beta)
    assert np.array_equal(lengths, si["length"])
    out = np.zeros((len(freqs_hz), n), dtype=np.complex128)
    fft_len = {(a, b): p for (a, b, p) in plan.segments()}
    for start, stop in epoch_bounds:
        ne = stop - start
        p_big = fft_len[(start, stop)]                 # whole epochs only (no time blocks here)
        lead = start - (start & ~63)                   # segments start on multiples of 64 samples
        X = fft(np.concatenate([np.zeros(lead), x[start:stop]]), n=p_big)
        xr_cache = {}
        for i, (om, L) in enumerate(zip(omegas, lengths)):
            method = si["method"][i]
            if method in (1, 3):                     # time domain / block convolution: the literal kernel
                psi, _ = orc.morse_kernel(L, om, gamma, beta, normalization, order)
                out[i, start:stop] = orc.overlap_add_convolve(x[start:stop], psi)
                continue
            if method == 2:
                H = exact_response(2 * np.pi * np.arange(p_big) / p_big, om, L, gamma, beta,
                                   normalization, order)
                out[i, start:stop] = ifft(X * H)[lead:lead + ne]
                continue
            R, lh, hop = int(si["decimation"][i]), int(si["halo"][i]), int(si["hop"][i])
            M = p_big // R
            shift = shift_of[R]
            if R not in xr_cache:
                U = shift * M // B
                sl = X[(np.arange(M) - U) % p_big]
                if not shift:
                    sl = sl * low_cut(cut_of[R], p_big, M)
                xr_cache[R] = ifft(sl) / R                                 # the engine's x_R: bin u is frequency u - U
            xr = xr_cache[R]
            k = np.arange(B)
            H = exact_response(2 * np.pi * (k - shift) / (B * R), om, L, gamma, beta, normalization, order)
            tw = np.exp(2j * np.pi * np.outer(k - shift, np.arange(R)) / (B * R))
            nblk = int(math.ceil(math.ceil((lead + ne) / R) / hop))
            y = np.zeros(R * nblk * hop, dtype=np.complex128)
            for b in range(nblk):
                idx = (b * hop - lh + np.arange(B)) % M
                XB = fft(xr[idx])
                blk = ifft((XB * H)[:, None] * tw, axis=0)        # [m, r]
                if shift:   # the carrier of the shifted band: block position and sample within the block
                    blk = blk * np.exp(-2j * np.pi * shift * ((b * hop - lh) + np.arange(B)) / B)[:, None]
                y[R * b * hop: R * (b + 1) * hop] = blk[lh:lh + hop].reshape(-1)
            out[i, start:stop] = y[lead:lead + ne]
    return out


def amplitude_interpolated(x, fs, freqs_hz, epoch_bounds=None, gamma=3.0, beta=20.0, B=256, plan=None):
    """|W| (S, N) float64 the way the engine makes it for amplitude output: the levels the
    planner designed an interpolator for (``plan.debug_interp()``) through q phases + the 8-tap
    FIR with the planner's own float32 coefficients and demodulation bins, every other scale
    as in ``cwt_decimated``.  Whole epochs only."""
    from ghost_amd.engine import CwtPlan
    x = np.asarray(x).squeeze().astype(np.float64)
    n = x.size
    if epoch_bounds is None:
        epoch_bounds = np.array([[0, n]])
    freqs_hz = np.atleast_1d(np.asarray(freqs_hz, dtype=np.float64))
    if plan is None:
        plan = CwtPlan(n, 1, fs, freqs_hz, gamma=gamma, beta=beta, epoch_bounds=epoch_bounds, output="amplitude")
    out = np.abs(cwt_decimated(x, fs, freqs_hz, epoch_bounds, gamma, beta, B, plan=plan))
    si, di, levels = plan.scale_info(), plan.debug_interp(), plan.debug_levels()
    by_r = {lv["decimation"]: d for lv, d in zip(levels, di["levels"])}
    shift_of = {lv["decimation"]: lv["band_shift"] for lv in levels}
    cut_of = {lv["decimation"]: lv["low_cut"] for lv in levels}
    omegas = orc.hz_to_rad(freqs_hz, fs)
    xc = x - x.mean()
    fft_len = {(a, b): p for (a, b, p) in plan.segments()}
    T = 8
    for start, stop in epoch_bounds:
        ne = stop - start
        p_big = fft_len[(start, stop)]
        lead = start - (start & ~63)
        X = fft(np.concatenate([np.zeros(lead), xc[start:stop]]), n=p_big)
        for i, (om, L) in enumerate(zip(omegas, si["length"])):
            R = int(si["decimation"][i])
            d = by_r.get(R) if si["method"][i] == 0 else None
            if d is None:
                continue
            q, I, lh, hop = d["q"], d["factor"], int(si["halo"][i]), int(si["hop"][i])
            coef = d["coef"][1 if int(L) % 2 == 0 else 0].astype(np.float64)   # [I][8]
            kc = int(di["demod"][i])
            assert kc % q == 0, (kc, q)
            M = p_big // R
            shift = shift_of[R]
            sl = X[(np.arange(M) - shift * M // B) % p_big]
            if not shift:
                sl = sl * low_cut(cut_of[R], p_big, M)
            xr = ifft(sl) / R
            k = np.arange(B)
            G = exact_gain(2 * np.pi * (k - shift) / (B * R), om, int(L), gamma, beta)
            tw = np.exp(2j * np.pi * np.outer(k, np.arange(q)) / (B * q))            # phase p of bin k
            demod = np.exp(-2j * np.pi * kc * np.arange(B * q) / (B * q))           # bins counted from k_c
            nblk = int(math.ceil(math.ceil((lead + ne) / R) / hop))
            y = np.zeros(R * nblk * hop)
            for b in range(nblk):
                idx = (b * hop - lh + np.arange(B)) % M
                XB = fft(xr[idx])
                z = ifft((XB * G)[:, None] * tw, axis=0).reshape(-1) * demod        # z[q m + p]
                m0 = lh * q
                acc = np.zeros((hop * q, I), dtype=np.complex128)
                for j in range(T):
                    seg = z[m0 + j - (T // 2 - 1): m0 + j - (T // 2 - 1) + hop * q]
                    acc += seg[:, None] * coef[None, :, j]
                y[R * b * hop: R * (b + 1) * hop] = np.abs(acc.reshape(-1))
            out[i, start:stop] = y[lead:lead + ne]
    return out
