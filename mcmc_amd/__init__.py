"""mcmc_amd -- MI355X-native Metropolis-Hastings likelihood path.

Python is only the host-side mirror of the C interface: the hot path lives in
``lib/libmhlik.so`` (hand-written HIP for gfx950, C-ABI in ``include/mhlik.h``)
and the samplers in ``lib/libmcmc_amd.so`` (C, reference signatures).  Nothing
here computes a likelihood on the CPU; if the HIP library or a GPU is missing
the calls raise.
"""
from ._lib import MhlikError, load_mhlik, lib_path  # noqa: F401
from .likelihood import Likelihood, LAYOUT_TILED, LAYOUT_COLMAJOR, F64, F32  # noqa: F401

__all__ = ["Likelihood", "MhlikError", "load_mhlik", "lib_path", "LAYOUT_TILED", "LAYOUT_COLMAJOR", "F64", "F32"]
