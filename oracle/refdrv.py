"""ctypes access to oracle/_ref/libref_driver.so (the unmodified reference behind oracle/ref_driver.cpp).
TEST INFRASTRUCTURE ONLY.  ``available()`` is False on machines without a prebuilt oracle/_ref."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_driver.so")
_L = None


def available():
    return os.path.exists(_PATH) and os.path.exists(os.path.join(_HERE, "_ref", "lib_gpboost_ref.so"))


def _lib():
    global _L
    if _L is None:
        C.CDLL(os.path.join(_HERE, "_ref", "lib_gpboost_ref.so"), mode=C.RTLD_GLOBAL)
        _L = C.CDLL(_PATH)
        _L.refdrv_create.restype = C.c_void_p
        _L.refdrv_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_double, C.c_int, C.c_char_p,
                                     C.c_int, C.c_char_p, C.c_int]
    return _L


def _P(a):
    return a.ctypes.data_as(C.c_void_p)


class RefModel(object):
    """One Gaussian Vecchia GP built by the reference's own REModel constructor."""

    def __init__(self, coords, cov_function="exponential", shape=0.5, m=30, ordering="random", seed=0, threads=8):
        cm = np.asfortranarray(coords, dtype=np.float64)
        self.n, self.d = cm.shape
        self.m = min(m, self.n - 1)
        h = _lib().refdrv_create(self.n, _P(cm), self.d, cov_function.encode(), float(shape), int(m),
                                 ordering.encode(), int(seed), b"gaussian", int(threads))
        if not h:
            raise RuntimeError("reference model creation failed")
        self.h = C.c_void_p(h)

    def __del__(self):
        try:
            _lib().refdrv_free(self.h)
        except Exception:
            pass

    def perm(self):
        p = np.empty(self.n, dtype=np.int32)
        _lib().refdrv_get_perm(self.h, _P(p))
        return p

    def neighbors(self):
        nn = np.empty((self.n, self.m), dtype=np.int32)
        _lib().refdrv_get_neighbors(self.h, C.c_int(self.m), _P(nn))
        return nn

    def nll_grad(self, y, cov_pars):
        y = np.ascontiguousarray(y, dtype=np.float64)
        cp = np.ascontiguousarray(cov_pars, dtype=np.float64)
        nll = C.c_double(0); g = np.empty(3); pt = np.empty(3)
        rc = _lib().refdrv_nll_grad(self.h, _P(y), _P(cp), C.byref(nll), _P(g), _P(pt))
        if rc != 0:
            raise RuntimeError("refdrv_nll_grad failed")
        return nll.value, g, pt

    def factor(self):
        """(A, D, y_aux_vecchia_order) of the last nll_grad call."""
        A = np.empty((self.n, self.m)); Di = np.empty(self.n); ya = np.empty(self.n)
        rc = _lib().refdrv_get_factor(self.h, C.c_int(self.m), _P(A), _P(Di), _P(ya))
        if rc != 0:
            raise RuntimeError("refdrv_get_factor failed")
        return A, 1. / Di, ya
