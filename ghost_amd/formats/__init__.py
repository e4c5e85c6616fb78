"""Input / output adapters (reference: ghost/formats)."""
from .preprocessing import *    # noqa: F401,F403
from .postprocessing import *   # noqa: F401,F403
