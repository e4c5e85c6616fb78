"""The reference's ``ghost.sigtools`` names on the GPU operators (ghost/sigtools/__init__.py:3-5):
``fastconv_scipy`` / ``fastconv_fftw`` / ``fastconv_freq_scipy`` / ``fastconv_freq_fftw``,
``analytic_signal_fftw`` and ``chirpz_dft`` resolve to the ``*_hip`` operators of ``ghost_amd.sigtools``
-- same arguments, argument meaning and exceptions; ``n_threads`` is accepted and ignored (the device is
the parallelism).  Under these names the operators run in float64 on the device (precision='high') and
return float64 / complex128 like the reference; the ``*_hip`` names default to float32."""
from .analytic import *      # noqa: F401,F403
from .convolution import *   # noqa: F401,F403
from .fourier import *       # noqa: F401,F403
