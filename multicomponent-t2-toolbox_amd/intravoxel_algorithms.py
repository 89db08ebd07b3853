"""Drop-ins for intravoxel_algorithms/algorithms.py and bayesian_interpolation.py of the reference
(same names, arguments and return values), each one running on the MI355X through the C ABI.

A single call solves a single voxel -- one wavefront of a 256-CU device -- so these exist for API
compatibility and testing; throughput comes from `motor.fitting_slice_T2` / `Met2Plan.fit`, which
hand the device whole rows or volumes.  Non-finite input raises ValueError like the reference's
np.asarray_chkfinite (algorithms.py:56).

The device entry normalises every voxel by its first echo and un-normalises the result
(motor:129-155); for these single-signal calls that is an exact round trip up to rounding, and it
requires M[0] > 0 and sum(M) > 0 -- the driver's own gates (motor:124,131)."""
import numpy as np
import torch

from ._cache import plan_for


def _chk(*arrs):
    out = []
    for a in arrs:
        a = np.ascontiguousarray(a, dtype=np.float64)
        if not np.isfinite(a).all():
            raise ValueError("array must not contain infs or NaNs")
        out.append(a)
    return out


def _fit1(method, D, M, L=None, lambda_reg=None, **options):
    plan = plan_for(D, L, lambda_reg)
    data = torch.as_tensor(M[None, :], device=plan.device)
    # `factor` / `reg_opt` are per-call arguments in the reference; the cached plan is shared with other callers of the same
    # (D, L, grid), so its options are put back after the call
    saved = plan.get_options(*options) if options else {}
    try:
        if options:
            plan.set_options(**options)
        out = plan.fit(method, data, want_maps=False, want_lambda=True)
    finally:
        if saved:
            plan.set_options(**saved)
    if int(out["status"][0].item()) == 0:
        raise ValueError("signal fails the driver's gates (need M[0] > 0 and sum(M) > 0)")
    return (out["fsol"][0].cpu().numpy(), out["sig"][0].cpu().numpy(), float(out["reg"][0].item()), float(out["lam"][0].item()))


def nnls(A, b):
    """algorithms.py:55-82 -> (x, rnorm)"""
    A, b = _chk(A, b)
    f, sig, _, _ = _fit1("NNLS", A, b)
    return f, float(np.sqrt(np.sum((sig - b) ** 2)))


def nnls_tik(Dic_i, M, Laplac, reg_opt):
    """algorithms.py:262-269 -> f"""
    D, M, L = _chk(Dic_i, M, Laplac)
    return _fit1("T2SPARC", D, M, L, t2sparc_lambda=float(reg_opt))[0]


def nnls_x2(Dic_i, M, Laplac, factor):
    """algorithms.py:211-223 -> (f, reg_opt, k_est)"""
    D, M, L = _chk(Dic_i, M, Laplac)
    f, _, kest, lam = _fit1("X2", D, M, L, x2_factor=float(factor))
    return f, lam, kest


def nnls_lcurve_wrapper(D, y, Laplac_mod, lambda_reg):
    """algorithms.py:88-113 -> reg_opt"""
    D, y, L, lg = _chk(D, y, Laplac_mod, lambda_reg)
    return _fit1("L_curve", D, y, L, lg)[3]


def nnls_gcv(Dic_i, M, L):
    """algorithms.py:276-283 -> (f, reg_opt)"""
    D, M, L = _chk(Dic_i, M, L)
    f, _, _, lam = _fit1("GCV", D, M, L)
    return f, lam


def BayesReg_nnls(Dic_i, M, L):
    """bayesian_interpolation.py:84-105 -> (f, reg_sol)"""
    D, M, L = _chk(Dic_i, M, L)
    f, _, _, lam = _fit1("BayesReg", D, M, L)
    return f, lam
