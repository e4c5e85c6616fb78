"""Signal tools on the GPU (reference: ghost/sigtools): the FFT-convolution operators the
transform is built from."""
from .convolution import *   # noqa: F401,F403
