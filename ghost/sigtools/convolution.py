"""``ghost.sigtools.convolution`` under the reference's names (ghost/sigtools/convolution.py:3-4)."""
import ghost_amd.sigtools.convolution as _gpu
from ghost_amd.sigtools.convolution import *          # noqa: F401,F403
from ghost_amd.sigtools.convolution import fastconv_hip, fastconv_freq_hip

__all__ = ['fastconv_scipy', 'fastconv_fftw', 'fastconv_freq_scipy', 'fastconv_freq_fftw'] + list(_gpu.__all__)


def fastconv_scipy(signal, kernel, *, mode=None, fft_length=None):
    """convolution.py:16-87 on the device."""
    return fastconv_hip(signal, kernel, mode=mode, fft_length=fft_length, precision='high')


def fastconv_fftw(signal, kernel, *, mode=None, fft_length=None, n_threads=None):
    """convolution.py:89-216 on the device (``n_threads`` has no meaning here)."""
    return fastconv_hip(signal, kernel, mode=mode, fft_length=fft_length, precision='high')


def fastconv_freq_scipy(signal_td, kernel_fd, kernel_len, *, mode=None):
    """convolution.py:218-285 on the device."""
    return fastconv_freq_hip(signal_td, kernel_fd, kernel_len, mode=mode, precision='high')


def fastconv_freq_fftw(signal_td, kernel_fd, kernel_len, *, mode=None, n_threads=None):
    """convolution.py:287-402 on the device (``n_threads`` has no meaning here)."""
    return fastconv_freq_hip(signal_td, kernel_fd, kernel_len, mode=mode, precision='high')
