"""``ghost.sigtools.analytic`` under the reference's names (ghost/sigtools/analytic.py:3)."""
import ghost_amd.sigtools.analytic as _gpu
from ghost_amd.sigtools.analytic import *             # noqa: F401,F403
from ghost_amd.sigtools.analytic import analytic_signal_hip

__all__ = ['analytic_signal_scipy', 'analytic_signal_fftw'] + list(_gpu.__all__)


def analytic_signal_fftw(signal, *, fft_length=None, n_threads=None):
    """analytic.py:22-112 on the device (``n_threads`` has no meaning here)."""
    return analytic_signal_hip(signal, fft_length=fft_length, precision='high')


def analytic_signal_scipy(signal):
    """analytic.py:15-19 forwards to ``scipy.signal.hilbert``; the same numbers from the device."""
    return analytic_signal_hip(signal, precision='high')
