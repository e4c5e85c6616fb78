"""CPU restatement ("oracle") of ``ghost.wave.ContinuousWaveletTransform.transform``.

TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP engine in
``ghost_amd/``; the product never imports it.

Parity pin
----------
PINNED.  The reference is pure Python and imports in the build container, so
every function below was checked there against the unmodified reference and the
resulting vectors are committed under ``tests/golden/`` together with the script
that made them (``tests/golden/make_golden.py``).  ``tests/test_oracle.py``
re-checks this file against those vectors on every run (no reference needed).
The reference's own tests hold no fixtures for this path;
its only test of the layer below (``tests/test_convolution.py:6-21``, fastconv
vs ``scipy.signal.convolve``) is reproduced in ``tests/test_oracle.py`` too.

Arithmetic below the seam is third-party FFT (scipy/pocketfft here, FFTW3 via
pyfftw upstream; ``setup.py:28-31`` pins only ``scipy>=1.6.1``).  DFT results are
defined mathematically, so parity is against exact DFT semantics, float64.

Two forms are given:

* literal  -- what the reference does: L-point time-domain Morse kernel,
  chunked overlap-add FFT convolution, ``'same'`` crop, ``abs``.
* spectral -- the algebraically equivalent frequency-domain form the GPU engine
  is built on (one FFT of the zero-padded epoch, closed-form one-sided filter).
  It equals the literal form to ~2e-8 while the filter is negligible at
  Nyquist (f < ~0.28 fs); it is checked against the literal form in the tests.

All citations are ``file:line`` relative to the reference repository root.
"""
import math
from multiprocessing.pool import ThreadPool

import numpy as np
from scipy.fft import fft as _fft, ifft as _ifft

__all__ = [
    "morse_peak_freq", "morse_high_cutoff", "morse_base_length",
    "morse_freq_bounds", "morse_lengths", "morse_kernel",
    "overlap_add_convolve", "contiguous_segments", "frequency_grid",
    "hz_to_rad", "rad_to_hz", "cwt_complex", "cwt_amplitude",
    "spectral_filter", "cwt_complex_spectral", "kernel_response",
]

# --------------------------------------------------------------------------
# Morse scalars
# --------------------------------------------------------------------------

def morse_peak_freq(gamma=3.0, beta=20.0):
    """Peak radian frequency of the mother wavelet, (beta/gamma)**(1/gamma).

    ghost/wave/morseutils.py:315 (``morsefreq`` with nout=1)."""
    return float(np.exp((np.log(beta) - np.log(gamma)) / gamma))


def morse_high_cutoff(gamma=3.0, beta=20.0, eta=0.1):
    """Highest usable peak frequency (rad/sample): first point of a 10 000-point
    grid on [1e-12, pi] at which the wavelet at Nyquist has dropped below
    ``eta`` of its peak.  ghost/wave/morseutils.py:612-624 (``morsehigh``)."""
    grid = np.linspace(1e-12, np.pi, 10000)
    w = morse_peak_freq(gamma, beta) * np.pi / grid
    ln_psi = (beta / gamma) * np.log(np.exp(1) * gamma / beta) \
        + beta * np.log(w) - w ** gamma
    first = np.argwhere(np.log(eta) - ln_psi < 0).squeeze()[0]
    return float(grid[first])


def morse_base_length(gamma=3.0, beta=20.0):
    """Four mother-wavelet footprints, in samples at unit scale.

    ghost/wave/morse.py:101 and :115-116."""
    w0 = morse_peak_freq(gamma, beta)
    return (2 * np.sqrt(2) * np.sqrt(gamma * beta)) / w0 * 4


def morse_freq_bounds(n_samples, gamma=3.0, beta=20.0, p=5):
    """[lowest, highest] analysable peak frequency in rad/sample for a segment
    of ``n_samples``.  ghost/wave/morse.py:93-106 (``compute_freq_bounds``)."""
    wh = morse_high_cutoff(gamma, beta)
    w0 = morse_peak_freq(gamma, beta)
    max_length = int(np.floor(n_samples / p))
    max_scale = max_length / morse_base_length(gamma, beta)
    return [w0 / max_scale, wh]


def morse_lengths(norm_radian_freqs, gamma=3.0, beta=20.0):
    """Kernel length per analysis frequency: ceil(w0/omega * base_length).

    ghost/wave/morse.py:108-122 (``compute_lengths``)."""
    w0 = morse_peak_freq(gamma, beta)
    scale = w0 / np.asarray(norm_radian_freqs, dtype=np.float64)
    return np.ceil(scale * morse_base_length(gamma, beta)).astype(int)


# --------------------------------------------------------------------------
# Morse time-domain kernel (literal)
# --------------------------------------------------------------------------

def morse_kernel(length, omega, gamma=3.0, beta=20.0, normalization="bandpass", order=0):
    """``Morse.__call__(length)``: one wavelet of the family, time domain and spectrum.

    Returns ``(psi, psif)``: complex128 (L,) time-domain kernel and float64 (L,)
    one-sided spectrum sampled on the L-point grid.  The defaults are what
    ``transform()`` uses ('bandpass', first wavelet: morse.py:84-91).

    Follows ghost/wave/morse.py:84-91 -> ghost/wave/morseutils.py:93-151
    (``_morsewave``) and :153-198 (``_morsewave_first_family``).  For the first wavelet
    with 'bandpass' normalisation the Laguerre factor and ``coeff`` are exactly 1
    (:188-196, :266-271); the spectrum is kept on bins 0..round(L/2)-1 only
    (:178, Python banker's ``round``) and is zero elsewhere (:177, :196).
    ``order`` k >= 1 multiplies by ``coeff * L_k^(c)(2 w^gamma)`` (:181-196, :256-273);
    'energy' uses ``exp(beta ** log(w) - w ** gamma)`` exactly as the reference writes it
    (:124) and ``coeff = sqrt(1/fact) * morseafunc(order=k+1)`` (:186-189, :249-251).
    """
    from scipy.special import gammaln, gamma as gammafunc
    L = int(length)
    w0 = morse_peak_freq(gamma, beta)
    fact = omega / w0                                        # :116
    w = 2 * np.pi * np.linspace(0, 1 - 1 / L, L) / fact      # :117
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if normalization == "energy":
            psizero = np.exp(-w ** gamma) if beta == 0 else np.exp(beta ** np.log(w) - w ** gamma)  # :121-124
        elif beta == 0:
            psizero = 2 * np.exp(-w ** gamma)
        else:
            psizero = 2 * np.exp(-beta * np.log(w0) + w0 ** gamma
                                 + beta * np.log(w) - w ** gamma)  # :130-131
    psizero[0] /= 2                                          # :133
    r = (2 * beta + 1) / gamma                               # :173-174
    c = r - 1
    index = np.arange(round(L / 2))                          # :178
    if normalization == "energy":
        a = np.sqrt(2 * np.pi * gamma * (2 ** r) * np.exp(gammaln(order + 1) - gammaln(order + r)))  # :249-251
        coeff = np.sqrt(1 / fact) * a                        # :186-189
    elif beta != 0:
        coeff = np.sqrt(np.exp(gammaln(r) + gammaln(order + 1) - gammaln(order + r)))  # :190-193
    else:
        coeff = 1
    x = 2 * w[index] ** gamma
    lag = np.zeros(index.size)
    for m in range(order + 1):                               # :266-271
        f = np.exp(gammaln(order + c + 1) - gammaln(c + m + 1) - gammaln(order - m + 1))
        lag += (-1) ** m * f * x ** m / gammafunc(m + 1)
    poly = np.zeros(L)
    poly[index] = lag                                        # :194
    with np.errstate(invalid="ignore"):
        psif = coeff * psizero * poly                        # :196
    psif[np.isinf(psif)] = 0                                 # :142
    centred = psif * np.exp(1j * w * (L + 1) / 2 * fact)     # :147
    psi = np.fft.ifft(centred)                               # :149
    return psi, psif


# --------------------------------------------------------------------------
# FFT overlap-add convolution (literal)
# --------------------------------------------------------------------------

def overlap_add_convolve(signal, kernel, mode="same", fft_length=None):
    """``fastconv_scipy``: chunked overlap-add linear convolution.

    ghost/sigtools/convolution.py:16-87.  Default FFT length 65 536, x4 until it
    holds the kernel (:64-66); chunk = F - M + 1 (:70); per chunk two forward
    FFTs, a product, one inverse (:74-76), accumulated into a complex128 buffer
    of N + M - 1 samples (:68, :77); centre crop for 'same' (:79-87)."""
    signal = np.asarray(signal)
    kernel = np.asarray(kernel)
    n, m = signal.shape[-1], kernel.shape[-1]
    total = n + m - 1
    if fft_length is None:
        fft_length = 65536
        while fft_length < m:
            fft_length *= 4
    elif fft_length < m:
        raise ValueError("FFT length must be at least the kernel size")
    if mode == "valid" and n < m:
        raise ValueError("input shorter than kernel in 'valid' mode")
    acc = np.zeros(total, dtype="<c16")
    chunk = min(fft_length - m + 1, n)
    kernel_fd = _fft(kernel, n=fft_length)   # the reference recomputes this per chunk
    for start in range(0, n, chunk):
        length = min(chunk, n - start)
        prod = _fft(signal[start:start + length], n=fft_length) * kernel_fd
        piece = _ifft(prod)[:length + m - 1]
        acc[start:start + len(piece)] += piece
    size = {"full": total, "same": n, "valid": n - m + 1}[mode]
    first = (total - size) // 2
    return acc[first:first + size]


# --------------------------------------------------------------------------
# Driver pieces of transform()
# --------------------------------------------------------------------------

def hz_to_rad(f_hz, fs):
    """ghost/wave/transforms.py:408-410."""
    return np.array(f_hz) / (fs / 2.0) * np.pi


def rad_to_hz(w, fs):
    """ghost/wave/transforms.py:404-406."""
    return np.array(w) / np.pi * fs / 2.0


def contiguous_segments(timestamps, fs):
    """Index ranges [start, stop) of runs whose successive timestamps differ by
    less than two sample periods.  ghost/utils.py:3-42 with ``index=True,
    inclusive=False`` as called from ghost/formats/preprocessing.py:170-174."""
    t = np.asarray(timestamps, dtype=np.float64)
    if not np.all(t[:-1] <= t[1:]):
        t = np.sort(t)
    breaks = np.flatnonzero(np.diff(t) >= 2.0 / fs)
    starts = np.concatenate(([0], breaks + 1)).astype(int)
    stops = np.concatenate((breaks, [t.size - 1])).astype(int) + 1
    return np.stack((starts, stops), axis=1)


def frequency_grid(fs, shortest_epoch, freq_limits=None, voices_per_octave=10,
                   gamma=3.0, beta=20.0):
    """Descending log-spaced analysis frequencies in Hz.

    ghost/wave/transforms.py:147-175: bounds from the shortest epoch (:147-149),
    requested limits clamped into them (:412-434), then
    f_j = f_high / 2**(j/v), j = 0..floor(log2(f_high/f_low)*v)."""
    lo_ref, hi_ref = rad_to_hz(morse_freq_bounds(shortest_epoch, gamma, beta), fs)
    if freq_limits is None:
        f_low, f_high = lo_ref, hi_ref
    else:
        f_low, f_high = np.sort(freq_limits)
        f_low = max(f_low, lo_ref)
        f_high = min(f_high, hi_ref)
    n_oct = np.log2(f_high / f_low)
    j = np.arange(np.floor(n_oct * voices_per_octave) + 1)
    return f_high / 2 ** (j / voices_per_octave)


def cwt_complex(x, fs, freqs_hz, epoch_bounds=None, gamma=3.0, beta=20.0,
                n_threads=1, keep="complex", normalization="bandpass", order=0):
    """Complex wavelet coefficients, literal path, one channel.

    Mirror of the closure ``wavelet_conv`` (ghost/wave/transforms.py:187-204) and
    the setup before it: float64 copy with the GLOBAL mean removed (:142-143),
    per-frequency kernel of ``compute_lengths`` samples (:181-182, :194-197),
    per-epoch 'same' convolution (:202-203).  The reference keeps ``abs`` only
    (:204); ``keep='abs'`` does the same, ``keep='complex'`` returns the
    coefficients the parity gate is written on.  ``n_threads > 1`` maps the
    per-frequency loop over a ThreadPool exactly as ``parallel=True`` does
    (:206-218)."""
    x = np.asarray(x).squeeze().astype(np.float64)
    x = x - np.mean(x)
    n = x.shape[-1]
    if epoch_bounds is None:
        epoch_bounds = np.array([[0, n]])
    freqs_hz = np.atleast_1d(np.asarray(freqs_hz, dtype=np.float64))
    omegas = hz_to_rad(freqs_hz, fs)
    lengths = morse_lengths(omegas, gamma, beta)
    out = np.zeros((len(freqs_hz), n),
                   dtype=np.complex128 if keep == "complex" else np.float64)

    def one(idx):
        kernel, _ = morse_kernel(lengths[idx], omegas[idx], gamma, beta, normalization, order)
        for start, stop in epoch_bounds:
            res = overlap_add_convolve(x[start:stop], kernel)
            out[idx, start:stop] = res if keep == "complex" else np.abs(res)

    if n_threads > 1:
        pool = ThreadPool(n_threads)
        pool.map(one, range(len(freqs_hz)), chunksize=1)
        pool.close()
        pool.join()
    else:
        for idx in range(len(freqs_hz)):
            one(idx)
    return out


def cwt_amplitude(x, fs, freqs_hz, epoch_bounds=None, gamma=3.0, beta=20.0,
                  n_threads=1):
    """What ``transform`` stores in ``_amplitude`` (transforms.py:204, :231)."""
    return cwt_complex(x, fs, freqs_hz, epoch_bounds, gamma, beta,
                       n_threads=n_threads, keep="abs")


# --------------------------------------------------------------------------
# Spectral form (what the GPU engine evaluates) -- SURVEY.md Appendix A.2
# --------------------------------------------------------------------------

def spectral_filter(theta, omega, length, gamma=3.0, beta=20.0):
    """One-sided frequency response of the scale whose peak is ``omega``
    rad/sample, evaluated at radian frequencies ``theta`` in [0, pi):

        H(theta) = 2 exp(-b ln w0 + w0^g + b ln w - w^g) * exp(-i theta d),
        w = theta * w0 / omega,  d = (L-1)/2 - (L-1)//2  (0 odd L, 1/2 even L).

    This is the continuous spectrum the L-point kernel of ``morse_kernel``
    samples (morseutils.py:117, :130-131) together with the linear phase that its
    centring (:147) and the 'same' crop (convolution.py:85) leave behind."""
    theta = np.asarray(theta, dtype=np.float64)
    w0 = morse_peak_freq(gamma, beta)
    w = theta * (w0 / omega)
    with np.errstate(divide="ignore", invalid="ignore"):
        amp = 2 * np.exp(-beta * np.log(w0) + w0 ** gamma
                         + beta * np.log(w) - w ** gamma)
    amp = np.where(theta > 0, amp, 0.0)
    d = (length - 1) / 2 - (length - 1) // 2
    return amp * np.exp(-1j * theta * d)


def kernel_response(theta, omega, length, gamma=3.0, beta=20.0, normalization="bandpass", order=0):
    """Frequency response of the L-tap kernel the reference convolves with, as the
    'same' crop positions it: H(theta) = sum_n psi[n] exp(-i theta (n - (L-1)//2)), psi from
    ``morse_kernel`` (morseutils.py:149) and the offset from convolution.py:85.  Direct
    O(L) sum per frequency.  For (gamma, beta) = (3, 20) it equals ``spectral_filter`` to
    ~6e-9 of the peak; for heavier-tailed wavelets only this one is what the reference
    applies."""
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    psi, _ = morse_kernel(length, omega, gamma, beta, normalization, order)
    n = np.arange(int(length)) - (int(length) - 1) // 2
    return np.exp(-1j * np.outer(theta, n)) @ psi


def cwt_complex_spectral(x, fs, freqs_hz, epoch_bounds=None, gamma=3.0,
                         beta=20.0):
    """Spectral-form coefficients: per epoch one FFT of the zero-padded,
    mean-removed samples, times ``spectral_filter`` on the positive-frequency
    half, inverse FFT, first N_e samples."""
    x = np.asarray(x).squeeze().astype(np.float64)
    x = x - np.mean(x)
    n = x.shape[-1]
    if epoch_bounds is None:
        epoch_bounds = np.array([[0, n]])
    freqs_hz = np.atleast_1d(np.asarray(freqs_hz, dtype=np.float64))
    omegas = hz_to_rad(freqs_hz, fs)
    lengths = morse_lengths(omegas, gamma, beta)
    out = np.zeros((len(freqs_hz), n), dtype=np.complex128)
    for start, stop in epoch_bounds:
        ne = stop - start
        p = 1 << int(math.ceil(math.log2(ne + int(lengths.max()))))
        spec = _fft(x[start:stop], n=p)
        theta = 2 * np.pi * np.arange(p // 2) / p
        for i, (om, ln) in enumerate(zip(omegas, lengths)):
            prod = np.zeros(p, dtype=np.complex128)
            prod[:p // 2] = spec[:p // 2] * spectral_filter(theta, om, ln, gamma, beta)
            out[i, start:stop] = _ifft(prod)[:ne]
    return out
