"""Base class of a wavelet (reference: ghost/wave/wavelet.py:7-21) and the validated
attribute the concrete wavelets build their parameters from."""
import abc

__all__ = ["Wavelet"]


class Wavelet(metaclass=abc.ABCMeta):
    """A wavelet prints as its class name and can be deep-copied (``copy``)."""

    @abc.abstractmethod
    def copy(self):
        """A new, independent wavelet with the same parameters."""

    def __repr__(self):
        return type(self).__name__


class Positive:
    """Data descriptor for a strictly positive parameter stored as ``_<name>``.

    ``message`` is the ValueError text (``{}`` receives the rejected value, ``{fs}`` the
    owner's sampling rate); ``after`` names a method of the owner to call once the new
    value is in place (derived quantities)."""

    def __init__(self, message, after=None, store=None):
        self.message, self.after, self.store = message, after, store

    def __set_name__(self, owner, name):
        self.slot = self.store or "_" + name

    def __get__(self, obj, owner=None):
        return self if obj is None else getattr(obj, self.slot)

    def __set__(self, obj, value):
        if not value > 0:
            raise ValueError(self.message.format(value, fs=getattr(obj, "_fs", None)))
        setattr(obj, self.slot, value)
        if self.after:
            getattr(obj, self.after)(value)
