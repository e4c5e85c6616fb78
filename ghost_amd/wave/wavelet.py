"""Base class of a wavelet (reference: ghost/wave/wavelet.py:7-21)."""
from abc import ABC, abstractmethod

__all__ = ["Wavelet"]


class Wavelet(ABC):

    def __repr__(self):
        return self.__class__.__name__

    @abstractmethod
    def copy(self):
        pass
