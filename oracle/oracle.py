"""ctypes wrapper around oracle/libmet2_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

Function names and argument meaning follow the reference
(intravoxel_algorithms/algorithms.py, bayesian_interpolation.py, epg/epg.py,
flip_angle_algorithms/fa_estimation.py, motor/motor_recon_met2_real_data.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

METHODS = {"NNLS": 0, "T2SPARC": 1, "X2": 2, "L_curve": 3, "GCV": 4, "BayesReg": 5}
PENALTY_ORDER = {"I": 0, "L1": 1, "L2": 2, "InvT2": 3}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(force=False):
    so = os.path.join(_HERE, "libmet2_oracle.so")
    src = os.path.join(_HERE, "met2_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libmet2_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmet2_oracle.so")
        if not os.path.exists(so):
            so = build()
        _LIB = C.CDLL(so)
        _LIB.met2o_fminbound_poly.restype = C.c_double
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


# ------------------------------------------------------------------ EPG / dictionary
def epg_signal(n, tau, R1vec, R2vec, alpha, alpha_exc):
    R1 = _d(np.atleast_1d(R1vec)); R2 = _d(np.atleast_1d(R2vec))
    H = np.zeros((int(n), R2.shape[0]))
    lib().met2o_epg_signal(int(n), C.c_double(tau), R2.shape[0], _p(R1), _p(R2), C.c_double(alpha), C.c_double(alpha_exc), _p(H))
    return H


def dictionary_fa_major(Npc, T2s, T1s, nEchoes, tau, alpha_values, TR):
    """D[fa][te][t2] (oracle/device layout)."""
    T2s = _d(T2s); T1s = _d(T1s); al = _d(alpha_values)
    D = np.zeros((al.shape[0], int(nEchoes), int(Npc)))
    lib().met2o_dictionary(int(nEchoes), int(Npc), al.shape[0], _p(T2s), _p(T1s), C.c_double(tau), _p(al), C.c_double(TR), _p(D))
    return D


def create_Dic_3D(Npc, T2s, T1s, nEchoes, tau, alpha_values, TR):
    """Reference layout [nEchoes, Npc, nFA] (epg/epg.py:155-162)."""
    return np.ascontiguousarray(np.transpose(dictionary_fa_major(Npc, T2s, T1s, nEchoes, tau, alpha_values, TR), (1, 2, 0)))


def create_met2_design_matrix_epg(Npc, T2s, T1s, nEchoes, tau, flip_angle, TR):
    return dictionary_fa_major(Npc, T2s, T1s, nEchoes, tau, [flip_angle], TR)[0]


def penalty(Npc, name, T2s=None):
    L = np.zeros((Npc, Npc))
    t2 = _d(T2s) if T2s is not None else np.zeros(Npc)
    lib().met2o_penalty(int(Npc), PENALTY_ORDER[name], _p(t2), _p(L))
    return L


# ------------------------------------------------------------------ single-voxel solvers
def nnls(A, b):
    A = _d(A); b = _d(b)
    m, n = A.shape
    x = np.zeros(n); rn = C.c_double(0.0)
    lib().met2o_nnls(m, n, _p(A), _p(b), _p(x), C.byref(rn))
    return x, rn.value


def _solve(method, D, M, L, param=0.0, lam_grid=None, trace=0):
    D = _d(D); M = _d(M)
    m, n = D.shape
    L = _d(L) if L is not None else np.zeros((n, n))
    lg = _d(lam_grid) if lam_grid is not None else np.zeros(1)
    f = np.zeros(n); reg = C.c_double(0.0); extra = C.c_double(0.0)
    tx = np.full(max(trace, 1), np.nan); tf = np.full(max(trace, 1), np.nan); tn = C.c_int(0)
    st = lib().met2o_solve(METHODS[method], m, n, _p(D), _p(M), _p(L), C.c_double(param), _p(lg), lg.shape[0],
                           _p(f), C.byref(reg), C.byref(extra), int(trace), _p(tx), _p(tf), C.byref(tn))
    return f, reg.value, extra.value, st, tx[: tn.value], tf[: tn.value]


def nnls_tik(Dic_i, M, Laplac, reg_opt):
    return _solve("T2SPARC", Dic_i, M, Laplac, reg_opt)[0]


def nnls_x2(Dic_i, M, Laplac, factor, trace=0):
    f, lam, kest, st, tx, tf = _solve("X2", Dic_i, M, Laplac, factor, trace=trace)
    return (f, lam, kest, tx, tf) if trace else (f, lam, kest)


def nnls_lcurve_wrapper(D, y, Laplac_mod, lambda_reg, curves=False):
    nl = len(lambda_reg)
    f, lam, _, st, tx, tf = _solve("L_curve", D, y, Laplac_mod, 0.0, lambda_reg, trace=nl if curves else 0)
    return (lam, tx, tf) if curves else lam


def nnls_gcv(Dic_i, M, L, trace=0):
    f, lam, _, st, tx, tf = _solve("GCV", Dic_i, M, L, trace=trace)
    return (f, lam, tx, tf) if trace else (f, lam)


def BayesReg_nnls(Dic_i, M, L, trace=0):
    f, lam, _, st, tx, tf = _solve("BayesReg", Dic_i, M, L, trace=trace)
    return (f, lam, tx, tf) if trace else (f, lam)


def objective(method, D, M, L, lams):
    D = _d(D); M = _d(M); L = _d(L); lams = _d(lams)
    vals = np.zeros(lams.shape[0])
    lib().met2o_objective(METHODS[method], D.shape[0], D.shape[1], _p(D), _p(M), _p(L), lams.shape[0], _p(lams), _p(vals))
    return vals


def select_corner(x, y):
    x = _d(x); y = _d(y)
    sc = np.zeros(2 * x.shape[0])
    c = lib().met2o_select_corner(x.shape[0], _p(x), _p(y), _p(sc))
    return c, sc.reshape(2, -1)


def fminbound_poly(c, r, p, x1, x2, xatol=1e-5, maxfun=300, cap=400):
    c = _d(c); r = _d(r); p = _d(p)
    xs = np.zeros(cap); fs = np.zeros(cap); nf = C.c_int(0)
    xf = lib().met2o_fminbound_poly(c.shape[0], _p(c), _p(r), _p(p), C.c_double(x1), C.c_double(x2), C.c_double(xatol),
                                    int(maxfun), cap, _p(xs), _p(fs), C.byref(nf))
    return xf, xs[: nf.value], fs[: nf.value]


# ------------------------------------------------------------------ batches
def fit_batch(method, D_fa_major, L, data, fa_index, mask, lambda_reg=None, x2_factor=1.02, t2sparc_lambda=1.8, nthreads=1,
              want_lambda=False, intervals=None):
    """intervals: (x2_lo, x2_hi, gcv_lo, gcv_hi, bayes_lo, bayes_hi) of the lambda searches; None = the reference's literals."""
    if intervals is not None:
        iv = _d(np.asarray(intervals, dtype=np.float64).reshape(6))
        lib().met2o_set_intervals(_p(iv))
        try:
            return fit_batch(method, D_fa_major, L, data, fa_index, mask, lambda_reg, x2_factor, t2sparc_lambda, nthreads, want_lambda)
        finally:
            lib().met2o_set_intervals(None)
    D = _d(D_fa_major)
    nfa, nte, nt2 = D.shape
    data = _d(data); nvox = data.shape[0]
    fa = _d(fa_index); mk = _d(mask)
    L = _d(L) if L is not None else np.zeros((nt2, nt2))
    lg = _d(lambda_reg) if lambda_reg is not None else np.zeros(1)
    fsol = np.zeros((nvox, nt2)); sig = np.zeros((nvox, nte)); reg = np.zeros(nvox)
    status = np.zeros(nvox, dtype=np.int32)
    lam = np.zeros(nvox)
    rc = lib().met2o_fit_batch_lam(METHODS[method], nte, nt2, nfa, _p(D), _p(L), _p(lg), lg.shape[0], C.c_double(x2_factor),
                                   C.c_double(t2sparc_lambda), C.c_int64(nvox), _p(data), _p(fa), _p(mk), _p(fsol), _p(sig), _p(reg),
                                   _p(lam), status.ctypes.data_as(_ip), int(nthreads))
    if rc != 0:
        raise IndexError("FA index outside the dictionary")
    return (fsol, sig, reg, status, lam) if want_lambda else (fsol, sig, reg, status)


def fitting_slice_T2(mask_1d, data_1d, FA_index_1d, nx, Dic_3D, lambda_reg, T2dim, nEchoes, reg_method, Laplac, dist_x_prior=None):
    """motor/motor_recon_met2_real_data.py:113-162 (Dic_3D in the reference layout [te,t2,fa])."""
    D = np.ascontiguousarray(np.transpose(np.asarray(Dic_3D, dtype=np.float64), (2, 0, 1)))
    fsol, sig, reg, _ = fit_batch(reg_method, D, Laplac, data_1d, FA_index_1d, mask_1d, lambda_reg)
    return fsol, sig, reg


def metrics(fsol, T2s, mask, t2_myelin=40.0, t2_ie=200.0):
    fsol = _d(fsol); T2s = _d(T2s); mask = _d(mask)
    nvox, nt2 = fsol.shape
    maps = np.zeros((6, nvox))
    lib().met2o_metrics(nt2, _p(T2s), C.c_double(t2_myelin), C.c_double(t2_ie), C.c_int64(nvox), _p(fsol), _p(mask), _p(maps))
    return dict(zip(("MWF", "IEWF", "FWF", "T2_M", "T2_IE", "TWC"), maps))


def fa_bruteforce(D_fa_major, data, mask, nthreads=1, want_resid=False):
    D = _d(D_fa_major); nfa, nte, nt2 = D.shape
    data = _d(data); nvox = data.shape[0]; mk = _d(mask)
    idx = np.zeros(nvox); km = np.zeros(nvox); sse = np.zeros(nvox); f = np.zeros((nvox, nt2))
    resid = np.zeros((nvox, nfa)) if want_resid else None
    lib().met2o_fa_bruteforce(nte, nt2, nfa, _p(D), C.c_int64(nvox), _p(data), _p(mk), _p(idx), _p(km), _p(sse), _p(f),
                              _p(resid), int(nthreads))
    return (idx, km, sse, f, resid) if want_resid else (idx, km, sse, f)


def compute_optimal_FA(M, Dic_3D, alpha_values):
    """flip_angle_algorithms/fa_estimation.py:74-90."""
    D = np.ascontiguousarray(np.transpose(np.asarray(Dic_3D, dtype=np.float64), (2, 0, 1)))
    idx, km, sse, f = fa_bruteforce(D, np.asarray(M)[None, :], np.ones(1))
    i = int(idx[0])
    return i, alpha_values[i], km[0], sse[0], f[0]


def fa_spline(D_lr_fa_major, alpha_lr, D_hr_fa_major, alpha_hr, data, mask, nthreads=1):
    """fa_estimation.py:35-70 over a flat voxel list -> (idx, km, xmin)"""
    Dl = _d(D_lr_fa_major); Dh = _d(D_hr_fa_major); al = _d(alpha_lr); ah = _d(alpha_hr)
    nlr, nte, nt2 = Dl.shape
    data = _d(data); nvox = data.shape[0]; mk = _d(mask)
    idx = np.zeros(nvox); km = np.zeros(nvox); xm = np.zeros(nvox)
    lib().met2o_fa_spline(nte, nt2, nlr, _p(Dl), _p(al), Dh.shape[0], _p(Dh), _p(ah), C.c_int64(nvox), _p(data), _p(mk), _p(idx), _p(km),
                          _p(xm), int(nthreads))
    return idx, km, xm


def spline_weights(x):
    x = _d(x); n = x.shape[0]
    W = np.zeros((n, n))
    lib().met2o_spline_weights(n, _p(x), _p(W))
    return W


def nesma(data, mask, nthreads=1):
    """motor:305-333 on data [nx,ny,nz,nt] (already masked and clipped) -> filtered volume"""
    data = _d(data); mask = _d(mask)
    nx, ny, nz, nt = data.shape
    out = np.zeros_like(data)
    lib().met2o_nesma(nx, ny, nz, nt, _p(data), _p(mask), _p(out), int(nthreads))
    return out


def gaussian_kernel1d(sigma, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, int(truncate * sigma + 0.5)) -- the reference calls
    gaussian_filter(vol, 2.0, 0) (motor:342); returns (radius, weights)"""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return radius, phi / phi.sum()


def gaussian_smooth(data, sigma=2.0, nthreads=1):
    """motor:337-343: every echo volume of data [nx,ny,nz,nt] through scipy.ndimage.gaussian_filter(sigma)"""
    data = _d(data)
    nx, ny, nz, nt = data.shape
    r, w = gaussian_kernel1d(sigma)
    w = _d(w)
    out = np.zeros_like(data)
    lib().met2o_gaussian_smooth(nx, ny, nz, nt, r, _p(w), _p(data), _p(out), int(nthreads))
    return out


def max_threads():
    return lib().met2o_max_threads()
