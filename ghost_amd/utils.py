"""Host-side helpers (reference: ghost/utils.py)."""
import numpy as np

__all__ = ["get_contiguous_segments", "is_sorted"]


def is_sorted(x, chunk_size=None):
    """True if ``x`` is monotonically non-decreasing (ghost/utils.py:44-67)."""
    if not isinstance(x, (tuple, list, np.ndarray)):
        raise TypeError("Unsupported type {}".format(type(x)))
    x = np.atleast_1d(np.array(x).squeeze())
    if x.ndim > 1:
        raise ValueError("Input x must have only one non-singleton dimension")
    return bool(np.all(x[:-1] <= x[1:]))


def get_contiguous_segments(data, *, step=None, assume_sorted=None, index=False,
                            inclusive=False):
    """Runs of samples whose spacing stays below two steps.

    Same contract as ghost/utils.py:3-42: with ``index=True`` returns integer
    ``[start, stop)`` pairs (``[start, stop]`` if ``inclusive``), otherwise the
    ``[first, last + step)`` values; a break is any gap ``>= 2*step``."""
    if inclusive and not index:
        raise AssertionError("option 'inclusive' can only be used with 'index=True'")
    data = np.asarray(data)
    if not assume_sorted and not is_sorted(data):
        data = np.sort(data)
    if step is None:
        step = np.median(np.diff(data))
    breaks = np.flatnonzero(np.diff(data) >= 2 * step)
    starts = np.concatenate(([0], breaks + 1)).astype(int)
    stops = np.concatenate((breaks, [len(data) - 1])).astype(int)
    if index:
        last = stops if inclusive else stops + 1
        return np.stack((starts, last), axis=1).astype(int)
    return np.stack((data[starts], data[stops] + step), axis=1)
