"""Default import for the wave module (reference: ghost/wave/__init__.py:3-5)."""
from .wavelet import *      # noqa: F401,F403
from .morse import *        # noqa: F401,F403
from .transforms import *   # noqa: F401,F403
from .morlet import *       # noqa: F401,F403
