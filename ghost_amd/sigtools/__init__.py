"""Signal tools on the GPU (reference: ghost/sigtools): the FFT-convolution operators the
transform is built from, the analytic signal and the arbitrary-length DFT."""
from .analytic import *      # noqa: F401,F403
from .convolution import *   # noqa: F401,F403
from .fourier import *       # noqa: F401,F403
