"""Opt-in alias: the reference's import name for this engine.

With the repository root on ``sys.path`` (and the reference itself not installed),

    from ghost.wave import ContinuousWaveletTransform, Morse

is the reference's own import line (ghost/wave/__init__.py:3-5, ghost/__init__.py:3-6) and
resolves to ``ghost_amd`` -- same classes, same module layout (``ghost.wave.transforms``,
``ghost.wave.morse``, ``ghost.wave.morseutils``, ``ghost.sigtools.convolution``,
``ghost.formats.preprocessing`` ...).  Nothing is copied: every name is the ``ghost_amd`` object.
``ghost.sigtools`` is a small package of its own: the reference's operator names (``fastconv_scipy``,
``fastconv_fftw``, ``fastconv_freq_scipy``, ``fastconv_freq_fftw``, ``analytic_signal_fftw``,
``chirpz_dft``: ghost/sigtools/convolution.py:3-4, analytic.py:3, fourier.py:9) bound to the GPU operators.
"""
import importlib
import sys

from ghost_amd import *                    # noqa: F401,F403
from ghost_amd import __version__          # noqa: F401

for _name in ("wave", "wave.wavelet", "wave.morse", "wave.morseutils", "wave.morlet", "wave.transforms",
              "formats", "formats.preprocessing", "formats.postprocessing", "utils", "version"):
    _mod = importlib.import_module("ghost_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    if "." not in _name:
        setattr(sys.modules[__name__], _name, _mod)
del _name, _mod
from . import sigtools                   # noqa: E402,F401  (the reference's operator names)
