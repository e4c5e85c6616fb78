"""``ghost.sigtools.fourier`` under the reference's names (ghost/sigtools/fourier.py:9)."""
import ghost_amd.sigtools.fourier as _gpu
from ghost_amd.sigtools.fourier import *              # noqa: F401,F403
from ghost_amd.sigtools.fourier import chirpz_dft_hip

__all__ = ['chirpz_dft'] + list(_gpu.__all__)


def chirpz_dft(x):
    """fourier.py:9-52 on the device."""
    return chirpz_dft_hip(x, precision='high')
