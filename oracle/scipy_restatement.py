"""A second CPU baseline, shaped like the reference's own software stack.  TEST INFRASTRUCTURE, like everything under oracle/:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may import it.

The reference solves every voxel in Python: scipy's Lawson-Hanson `nnls` on the augmented system [D; sqrt(lambda) L] inside
scipy's bounded Brent `fminbound` (intravoxel_algorithms/algorithms.py:55-82, :211-233, :262-269), one image row per joblib task
on a multiprocessing pool (motor/motor_recon_met2_real_data.py:113-162, :427-441).  This file states that pipeline again with the
SciPy and joblib that ARE in this image (scipy.optimize.nnls of 1.15, where the reference pinned the private `__nnls` of ~1.8) --
written by the builder from the behaviour recorded in SURVEY.md section 8(a) rows N1, N2, X1, V1; it is not the reference's
text and the reference's files never travel to the GPU box.  tests/test_oracle_golden.py pins it to the reference's own outputs
(golden_S1.npz).  bench.py times it on a few hundred voxels as `cpu_baseline_scipy` beside the C port.
"""
import numpy as np
from scipy.optimize import fminbound, nnls as _sp_nnls


def nnls(A, b):
    """N1: x >= 0 minimising ||A x - b||, and the residual norm (Lawson-Hanson, iteration cap 3 n)."""
    A = np.asarray_chkfinite(A, dtype=np.float64)
    b = np.asarray_chkfinite(b, dtype=np.float64)
    x, rnorm = _sp_nnls(A, b, maxiter=3 * A.shape[1])
    return x, rnorm


def _augmented(D, M, L, lam):
    return np.vstack((D, np.sqrt(lam) * L)), np.concatenate((M, np.zeros(L.shape[0])))


def nnls_tik(D, M, L, lam):
    """N2: the Tikhonov-regularised solve at a fixed lambda."""
    A, b = _augmented(D, M, L, lam)
    return nnls(A, b)[0]


def nnls_x2(D, M, L, factor=1.02):
    """X1: lambda such that the misfit grows to `factor` times the unregularised one -> (f, lambda, achieved ratio)."""
    f0, _ = nnls(D, M)
    sse0 = float(np.sum((D @ f0 - M) ** 2))

    def cost(lam):
        f = nnls_tik(D, M, L, lam)
        return abs(float(np.sum((D @ f - M) ** 2)) - factor * sse0) / sse0

    lam = fminbound(cost, 0.0, 10.0, xtol=1e-5, maxfun=300, full_output=False, disp=0)
    f = nnls_tik(D, M, L, lam)
    return f, lam, float(np.sum((D @ f - M) ** 2)) / sse0


def fit_row(method, D, L, rows, factor=1.02, t2sparc_lambda=1.8):
    """V1 for one image row: gate (sum > 0 and first echo > 0), normalise by the first echo, solve, un-normalise.
    -> (fsol [nx, nT2], signal [nx, nTE], reg [nx])"""
    nx = rows.shape[0]
    fsol = np.zeros((nx, D.shape[1])); sig = np.zeros((nx, D.shape[0])); reg = np.zeros(nx)
    for i in range(nx):
        M = rows[i]
        if not (M.sum() > 0 and M[0] > 0):
            continue
        km = M[0]
        Mn = M / km
        if method == "X2":
            f, _, reg[i] = nnls_x2(D, Mn, L, factor)
        elif method == "T2SPARC":
            f = nnls_tik(D, Mn, L, t2sparc_lambda); reg[i] = t2sparc_lambda
        elif method == "NNLS":
            f = nnls(D, Mn)[0]
        else:
            raise ValueError("the SciPy restatement covers NNLS, T2SPARC and X2")
        fsol[i] = f * km
        sig[i] = (D @ f) * km
    return fsol, sig, reg


def fit_rows(method, D, L, data, n_rows=8, n_jobs=1, factor=1.02):
    """The driver's step 3: the voxel list cut into `n_rows` image rows, one joblib task each on a multiprocessing pool."""
    data = np.ascontiguousarray(data, dtype=np.float64)
    parts = [p for p in np.array_split(np.arange(data.shape[0]), max(1, n_rows)) if p.size]
    if n_jobs == 1:
        outs = [fit_row(method, D, L, data[p], factor) for p in parts]
    else:
        from joblib import Parallel, delayed
        outs = Parallel(n_jobs=n_jobs, backend="multiprocessing")(delayed(fit_row)(method, D, L, data[p], factor) for p in parts)
    return tuple(np.concatenate([o[k] for o in outs], axis=0) for k in range(3))
