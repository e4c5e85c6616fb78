"""Input adapters (reference: ghost/formats)."""
from .preprocessing import *   # noqa: F401,F403
