"""Stage-by-stage comparison of a Laplace gradient with the oracle's (shared by the GPU test and by a CPU test of the tolerances)."""
import numpy as np


def check_stages(g, parts, gref, oparts):
    """U = (Sigma^-1 + W)^-1 Z comes from a block CG that stops when the MEAN residual norm drops below 1e-2: the stopping test compares a
    rounded number with a threshold, so two correct implementations may differ by one iteration (tests/test_laplace_gpu.py allows +-1).
    One iteration moves d logdet / d mode by ~1e-4 of its scale and the gradient by up to 4e-5 (tests/test_oracle_golden.py measures it by
    capping the oracle's iteration count); with equal counts the agreement is ~3e-6 / 1e-7.  The tolerances carry a 10x margin over that;
    the 1e-5 pin on the gradient is the reference fixture (tests/golden/laplace_grad_ref.npz)."""
    sc = np.abs(oparts["dlogdet_dmode"]).max()
    np.testing.assert_allclose(parts["dlogdet_dmode"], oparts["dlogdet_dmode"], rtol=0, atol=1e-3 * sc)
    np.testing.assert_allclose(parts["implicit_solve"], oparts["implicit_solve"], rtol=0, atol=2e-2 * np.abs(oparts["implicit_solve"]).max())
    np.testing.assert_allclose(parts["per_par"][:, 0], oparts["per_par"][:, 0], rtol=1e-5)        # mode' SigmaI_deriv mode: only the mode itself in it
    np.testing.assert_allclose(parts["per_par"][:, 1:3], oparts["per_par"][:, 1:3], rtol=1e-3)
    np.testing.assert_allclose(g, gref, rtol=2e-4, atol=1e-5)
