"""nelpy / numpy input adapter (reference: ghost/formats/preprocessing.py:15-195).

``standardize_asa`` is a decorator with the reference's signature.  It turns the
first data argument -- a numpy array or a nelpy ``RegularlySampledAnalogSignalArray``
(duck-typed, so nelpy itself is optional) -- into the keyword set the wrapped
function receives: data, ``fs``, timestamps and ``epoch_bounds``.

Deliberate differences from the reference ("deviations"):
* default timestamps work on current numpy (reference uses ``np.float``, :145-147);
* nelpy epochs are the cumulative sum of ``lengths`` (reference forgets the
  cumsum, :102-103);
* ``n_signals=None`` on the decorator lets multi-signal input through, which the
  multichannel extension of ``transform`` uses;
* timestamps the adapter makes itself are a regular grid: their one segment is known without the scan, and with
  ``defer_abscissa=True`` the grid is handed on as a ``RegularGrid`` that becomes an array when somebody reads it.
"""
import logging
from functools import wraps

import numpy as np

from ..utils import get_contiguous_segments

__all__ = ["standardize_asa", "is_asa_like", "RegularGrid"]

_ASA_ATTRS = ("n_signals", "fs", "abscissa_vals", "lengths", "_data_colsig", "_data_rowsig")


def is_asa_like(obj):
    """A nelpy RegularlySampledAnalogSignalArray, or anything shaped like one."""
    return all(hasattr(obj, a) for a in _ASA_ATTRS)


class RegularGrid:
    """Timestamps ``arange(n) / rate`` that nobody has asked for yet (``standardize_asa(defer_abscissa=True)``):
    a callee that only keeps them for later -- ``ContinuousWaveletTransform.time`` -- need not pay for a million
    divisions per call.  ``np.asarray(grid)`` makes them."""
    __slots__ = ("n", "rate")

    def __init__(self, n, rate):
        self.n, self.rate = int(n), rate

    def __len__(self):
        return self.n

    def __array__(self, dtype=None, copy=None):
        t = np.arange(self.n, dtype=np.float64) / self.rate
        return t if dtype is None else t.astype(dtype, copy=False)


def standardize_asa(func=None, *, x, abscissa_vals=None, fs=None, n_signals=None,
                    rowsig=None, class_method=None, defer_abscissa=False):
    logger = logging.getLogger("ghost")

    if not isinstance(x, str):
        raise TypeError("'x' decorator argument must be a string")
    if n_signals is not None:
        try:
            ok = float(n_signals).is_integer() and int(n_signals) > 0
        except (TypeError, ValueError):
            ok = False
        if not ok:
            raise ValueError("'n_signals' must be a positive integer")
        n_signals = int(n_signals)
    for name, val in (("abscissa_vals", abscissa_vals), ("fs", fs)):
        if val is not None and not isinstance(val, str):
            raise TypeError("'{}' decorator argument must be a string".format(name))
    rowsig = False if rowsig is None else rowsig
    if rowsig not in (True, False):
        raise ValueError("'rowsig' decorator argument must be True or False")
    class_method = False if class_method is None else class_method
    if class_method not in (True, False):
        raise ValueError("'class_method' decorator argument must be True or False")
    pos = 1 if class_method else 0

    def decorate(function):
        fname = getattr(function, "_public_name", function.__name__)

        @wraps(function)
        def wrapped(*args, **kwargs):
            want_signals = kwargs.pop("_n_signals_override", n_signals)
            data_in = kwargs.pop(x, None)
            fs_in = kwargs.pop(fs, None) if fs is not None else None
            t_in = kwargs.pop(abscissa_vals, None) if abscissa_vals is not None else None
            positional = False
            if data_in is None:
                if len(args) <= pos:
                    raise TypeError("{}() missing 1 required positional argument: '{}'"
                                    .format(fname, x))
                data_in = args[pos]
                positional = True

            if is_asa_like(data_in):                      # preprocessing.py:78-114
                if want_signals is not None and data_in.n_signals != want_signals:
                    raise ValueError("Input object '{}'.n_signals=={}, but expected {}"
                                     .format(x, data_in.n_signals, want_signals))
                if fs is not None:
                    if fs_in is not None:
                        logger.warning("'%s' was passed in, but will be overwritten by the "
                                       "input object's 'fs' attribute", fs)
                    kwargs[fs] = data_in.fs
                if abscissa_vals is not None:
                    if t_in is not None:
                        logger.warning("'%s' was passed in, but will be overwritten by the "
                                       "input object's 'abscissa_vals' attribute", abscissa_vals)
                    kwargs[abscissa_vals] = np.asarray(data_in.abscissa_vals)
                data_out = data_in._data_rowsig if rowsig else data_in._data_colsig
                edges = np.concatenate(([0], np.cumsum(np.asarray(data_in.lengths)))).astype(int)
                kwargs["epoch_bounds"] = np.stack((edges[:-1], edges[1:]), axis=1)
            else:                                          # preprocessing.py:120-177
                if not isinstance(data_in, np.ndarray):
                    raise TypeError("Input was not a nelpy.RegularlySampledAnalogSignalArray"
                                    " so expected a numpy ndarray but got {}"
                                    .format(type(data_in)))
                data_out = np.atleast_1d(data_in.squeeze())
                if data_out.ndim == 1:
                    data_out = data_out.reshape((-1, 1))
                elif want_signals is None and data_out.ndim == 2:
                    # multichannel extension: rows are channels on input -> column signals
                    data_out = data_out.T
                if want_signals is not None and data_out.shape[-1] != want_signals:
                    raise ValueError("Expected {} number of signals but got {}"
                                     .format(want_signals, data_out.shape[0]))
                if fs is not None:
                    if fs_in is None:
                        raise TypeError("{}() missing 1 required keyword argument: '{}'"
                                        .format(fname, fs))
                    kwargs[fs] = fs_in
                if abscissa_vals is not None:
                    rate = 1 if fs_in is None else fs_in
                    generated = t_in is None
                    if generated:
                        logger.info("'%s' not passed in; generating from data", abscissa_vals)
                        t_in = (RegularGrid(data_out.shape[0], rate) if defer_abscissa
                                else np.arange(data_out.shape[0], dtype=np.float64) / rate)
                    else:
                        if not isinstance(t_in, np.ndarray):
                            raise TypeError("Expected '{}' to be a numpy.ndarray but got {}"
                                            .format(abscissa_vals, type(t_in)))
                        if t_in.ndim != 1:
                            raise ValueError("'{}' should have at most one non-singleton"
                                             " dimension".format(abscissa_vals))
                        if t_in.shape[0] != data_out.shape[0]:
                            raise ValueError("The argument '{}' has {} sample points, but the "
                                             "data '{}' has {}".format(
                                                 abscissa_vals, t_in.shape[0], x,
                                                 data_out.shape[0]))
                    if fs_in is None:
                        logging.warning("'%s' not passed in; assuming default of 1 Hz", fs)
                    kwargs[abscissa_vals] = t_in
                    if generated and len(t_in) > 0:
                        # a regular grid made here has no gaps: what get_contiguous_segments would find, without
                        # the scan (sortedness test + two differences over every sample: 0.8 ms per million)
                        kwargs["epoch_bounds"] = np.array([[0, len(t_in)]], dtype=int)
                    else:
                        kwargs["epoch_bounds"] = get_contiguous_segments(
                            t_in, step=1 / rate, assume_sorted=fs_in is None, index=True,
                            inclusive=False)

            if positional:
                args = tuple(data_out if i == pos else a for i, a in enumerate(args))
            else:
                kwargs[x] = data_out
            return function(*args, **kwargs)

        return wrapped

    if func:
        return decorate(func)
    return decorate
