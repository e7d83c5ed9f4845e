"""1d-spectral-optimal-transport_amd -- MI355X-native 1-D spectral optimal-transport (SOT) loss.

A hand-written HIP (gfx950) implementation of the hot path of
bernardo-torres/1d-spectral-optimal-transport (losses.Wasserstein1D / wasserstein_1d), behind the
reference's own Python interface.  Import it as ``sot_amd`` (the directory name is not a valid
Python identifier): ``from sot_amd.losses import Wasserstein1D``.
"""
from . import _native, build, distributed, losses, spectra  # noqa: F401
from .losses import MixOfLosses, Wasserstein1D, quantile_function, safe_divide, wasserstein_1d  # noqa: F401

__all__ = ["Wasserstein1D", "wasserstein_1d", "quantile_function", "MixOfLosses", "safe_divide",
           "losses", "distributed", "spectra", "build"]
