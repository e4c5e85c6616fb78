"""ghost_amd -- MI355X-native engine for the hot path of nelpy/ghost:
``ghost.wave.ContinuousWaveletTransform.transform``.

Same import surface as the reference for that path (ghost/__init__.py:3-6,
ghost/wave/__init__.py:3-5):

    from ghost_amd.wave import ContinuousWaveletTransform, Morse

The arithmetic runs in ``libghostcwt.so`` (hand-written HIP for gfx950) through
ctypes; there is no CPU fallback.
"""
from .wave.wavelet import *      # noqa: F401,F403
from .formats import *           # noqa: F401,F403
from .version import __version__  # noqa: F401
