"""Output adapter (reference: ghost/formats/postprocessing.py:13-65): hand a result back
as a numpy array or wrapped in a nelpy AnalogSignalArray built around the input object.
nelpy is optional; without it only the numpy form is available."""
import logging

import numpy as np

from .preprocessing import is_asa_like

__all__ = ["output_numpy_or_asa"]


def output_numpy_or_asa(obj, data, *, output_type=None, labels=None):
    """``data`` (n_samples, n_signals) as is, or as ``nelpy.AnalogSignalArray`` with the
    abscissa, sampling rate and support of ``obj`` when ``output_type='asa'``."""
    if not isinstance(data, np.ndarray):
        raise TypeError("data must be a numpy ndarray")
    if data.size == 0:
        logging.warning("Output data is empty")
    if output_type is not None and output_type != "asa":
        raise TypeError("Invalid output type {} specified".format(output_type))
    if output_type == "asa":
        try:
            import nelpy as nel
        except ImportError:
            raise ModuleNotFoundError("You must have nelpy installed for output type {}"
                                      .format(output_type))
        if not (isinstance(obj, nel.RegularlySampledAnalogSignalArray) or is_asa_like(obj)):
            raise TypeError("You specified output type {} but the input object was not a nelpy"
                            " object. Cannot form an ASA around the input object"
                            .format(output_type))
        # ASAs are (n_signals, n_samples)
        return nel.AnalogSignalArray(data.T, abscissa_vals=obj.abscissa_vals, fs=obj.fs,
                                     support=obj.support, labels=labels)
    return data
