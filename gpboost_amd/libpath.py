"""Locate and load lib_gpboost_amd.so (mirrors python-package/gpboost/libpath.py:22-37 of the reference)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "lib_gpboost_amd.so"


def find_lib_path():
    override = os.environ.get("GPBOOST_AMD_LIB")   # development builds (e.g. a single-MT library)
    cand = ([override] if override else []) + [os.path.join(_HERE, LIB_NAME), os.path.join(_HERE, "csrc", LIB_NAME)]
    found = [p for p in cand if os.path.isfile(p)]
    if not found:
        raise FileNotFoundError(
            "Cannot find %s (looked in %s). Build it with `make -C gpboost_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`; there is no pure-Python or CPU fallback." %
            (LIB_NAME, ", ".join(cand)))
    return found[0]


def load_lib():
    return ctypes.CDLL(find_lib_path())
