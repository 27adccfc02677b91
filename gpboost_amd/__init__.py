"""gpboost_amd -- MI355X (gfx950) hot path of GPBoost: Vecchia GP likelihood/gradient + feature histograms.

The numerical work lives in ``lib_gpboost_amd.so`` (hand-written HIP kernels behind a C ABI, see
``include/gpb_hip.h`` and ``include/gpboost_c_api_subset.h``).  This package is only the host-side
mirror of the reference's Python binding for that path.
"""
from .basic import GPBoostError, GPModel, device_count, selftest, set_device   # noqa: F401
from . import parallel, shim   # noqa: F401

__all__ = ["GPModel", "GPBoostError", "device_count", "selftest", "set_device", "shim", "parallel"]
