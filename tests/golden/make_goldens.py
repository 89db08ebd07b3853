#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by *running the reference itself*.

Runs only in the build container (needs /root/reference, read-only).  Nothing from
the reference is copied: this script imports the reference's modules and records
inputs/outputs of its functions as .npz data.

The reference does not import as shipped on this image (ordinary Python errors, see
SURVEY.md §8c): it imports `numba` (never used) and `scipy.optimize.__nnls` (a private
module that existed in SciPy ~1.8-1.11).  The shims below are this repo's own few-line
stand-ins: an empty `numba` module, and an adapter that forwards the old private
`__nnls.nnls(A,m,n,b,w,zz,index,maxiter)` call to the installed SciPy 1.15.3
Lawson-Hanson (`scipy.optimize._cython_nnls._nnls`).  For the end-to-end driver run,
`nibabel`, `progressbar` and `skimage.restoration` are replaced by in-memory stubs and
the Python-2 builtin `xrange` (fa_estimation.py:99) is aliased to `range`.

Usage:  python tests/golden/make_goldens.py            (writes tests/golden/*.npz)
"""
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MET2_REFERENCE", "/root/reference")


# --------------------------------------------------------------------------- shims
class _MemImage:
    def __init__(self, arr, affine=None):
        self._arr = np.asarray(arr)
        self.affine = np.eye(4) if affine is None else affine

    def get_fdata(self):
        return np.array(self._arr, dtype=np.float64)


_NIB_FILES = {}   # path -> ndarray (both "loaded" inputs and "saved" outputs)


def install_shims():
    import builtins

    sys.modules.setdefault("numba", types.ModuleType("numba"))

    import scipy.optimize as so
    from scipy.optimize import _cython_nnls

    adapter = types.ModuleType("scipy.optimize.__nnls")

    def nnls(A, m, n, b, w, zz, index, maxiter):
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        it = 3 * n if maxiter == -1 else int(maxiter)
        x, rnorm, info = _cython_nnls._nnls(A, b, it)
        return x, rnorm, (1 if info != -1 else 3)

    adapter.nnls = nnls
    sys.modules["scipy.optimize.__nnls"] = adapter
    setattr(so, "__nnls", adapter)
    if not hasattr(so, "_nnls"):
        setattr(so, "_nnls", adapter)

    # in-memory nibabel
    nib = types.ModuleType("nibabel")
    nib.load = lambda path: _MemImage(_NIB_FILES[path])
    nib.Nifti1Image = lambda arr, affine: _MemImage(arr, affine)

    def _save(img, path):
        _NIB_FILES[path] = np.array(img._arr)

    nib.save = _save
    sys.modules["nibabel"] = nib

    pb = types.ModuleType("progressbar")
    pb.progressbar = lambda it, **kw: it
    sys.modules["progressbar"] = pb

    if "skimage" not in sys.modules:
        sk = types.ModuleType("skimage")
        skr = types.ModuleType("skimage.restoration")
        skr.denoise_tv_chambolle = None
        skr.estimate_sigma = None
        sk.restoration = skr
        sys.modules["skimage"] = sk
        sys.modules["skimage.restoration"] = skr

    import matplotlib
    matplotlib.use("Agg")

    builtins.xrange = range     # fa_estimation.py:99 is Python-2 code
    if REF not in sys.path:
        sys.path.insert(0, REF)


# --------------------------------------------------------------------------- helpers
def t2_grid(npc):
    return np.logspace(math.log10(10.0), math.log10(2000.0), num=npc, endpoint=True, base=10.0)


def penalties(npc, T2s, create_Laplacian_matrix):
    out = {
        "I": create_Laplacian_matrix(npc, 0),
        "L1": create_Laplacian_matrix(npc, 1),
        "L2": create_Laplacian_matrix(npc, 2),
    }
    # motor/motor_recon_met2_real_data.py:263-269 (inline in the driver, re-evaluated here)
    T2s_mod = np.concatenate((np.array([T2s[0] - 1.0]), T2s[:-1]))
    deltaT2 = T2s - T2s_mod
    deltaT2[0] = deltaT2[1]
    out["InvT2"] = np.diag(1.0 / deltaT2)
    return out


def lambda_grid():
    lam = np.zeros(50)
    lam[1:] = np.logspace(math.log10(1e-8), math.log10(10.0), num=49, endpoint=True, base=10.0)
    return lam


def synth_voxels(rng, nvox, nte, epg_signal, fa_deg=None, snr_lo=50.0, snr_hi=150.0, te=10.0, TR=3000.0):
    """Two-lobe recipe of evaluate_all_methods_two_lobes_SNR50_150.py:156-170,385-394 (seeded)."""
    from scipy.stats import norm
    T2grid = np.linspace(1.0, 300.0, 1000)
    T1grid = 1000.0 * np.ones_like(T2grid)
    rad = np.pi / 180.0
    data = np.zeros((nvox, nte))
    fa_true = np.zeros(nvox)
    cache = {}
    for v in range(nvox):
        MWF = rng.uniform(0.05, 0.25)
        T2m = rng.uniform(15.0, 35.0)
        T2ie = rng.uniform(60.0, 90.0)
        FA = rng.uniform(90.0, 180.0) if fa_deg is None else fa_deg
        SNR = rng.uniform(snr_lo, snr_hi) if np.isfinite(snr_hi) else np.inf
        sm = rng.uniform(1.0, 3.0)
        sie = rng.uniform(6.0, 12.0)
        dist = MWF * norm.pdf(T2grid, T2m, sm) + (1.0 - MWF) * norm.pdf(T2grid, T2ie, sie)
        dist = dist / np.sum(dist)
        if FA not in cache:                  # same values as a fresh call; a constant flip angle is evaluated once
            if len(cache) > 4:
                cache.clear()
            cache[FA] = (1.0 - np.exp(-TR / T1grid)) * epg_signal(nte, te, 1.0 / T1grid, 1.0 / T2grid, FA * rad, FA / 2.0 * rad)
        sig_all = cache[FA]
        S = np.sum(1000.0 * sig_all * dist, axis=1)
        if np.isfinite(SNR):
            s = S[0] / SNR
            S = np.sqrt((S + rng.normal(0, s, S.shape)) ** 2 + rng.normal(0, s, S.shape) ** 2)
        data[v] = S
        fa_true[v] = FA
    return data, fa_true


class Trace:
    """Wrap an objective so that every (x, f(x)) Brent evaluates is recorded."""

    def __init__(self, fn):
        self.fn = fn
        self.xs = []
        self.fs = []

    def __call__(self, x, *args):
        f = self.fn(x, *args)
        self.xs.append(float(x))
        self.fs.append(float(f))
        return f


def pad_traces(traces, width=64):
    out = np.full((len(traces), width), np.nan)
    for i, t in enumerate(traces):
        out[i, : len(t)] = t[:width]
    return out


# --------------------------------------------------------------------------- generators
def gen_shape(tag, nte, npc, nvox, nvox_slow, seed, with_fa_full):
    from epg.epg import create_Dic_3D, epg_signal, create_met2_design_matrix_epg
    import intravoxel_algorithms.algorithms as alg
    import intravoxel_algorithms.bayesian_interpolation as bay
    from flip_angle_algorithms.fa_estimation import compute_optimal_FA
    from motor.motor_recon_met2_real_data import create_Laplacian_matrix, fitting_slice_T2

    rng = np.random.default_rng(seed)
    T2s = t2_grid(npc)
    T1s = 1000.0 * np.ones_like(T2s)
    TR, te = 3000.0, 10.0
    out = {"T2s": T2s, "T1s": T1s, "TR": TR, "tau": te, "nte": nte, "npc": npc}

    # ---- E1/E2/E3: dictionary
    fa_sel = np.array([90.0, 120.0, 150.0, 165.5, 180.0])
    out["fa_sel"] = fa_sel
    out["Dic_sel"] = create_Dic_3D(npc, T2s, T1s, nte, te, fa_sel, TR)          # [nte, npc, 5]
    alpha_values = np.linspace(90.0, 180.0, 91)
    Dic_full = create_Dic_3D(npc, T2s, T1s, nte, te, alpha_values, TR)
    out["alpha_values"] = alpha_values
    if with_fa_full:
        out["Dic_full"] = Dic_full
    # raw epg_signal with an excitation angle that is not alpha/2, two rates at once
    out["epg_raw_in"] = np.array([nte, te, 1.0 / 900.0, 1.0 / 1100.0, 1.0 / 35.0, 1.0 / 80.0, 2.2, 1.3])
    out["epg_raw_out"] = epg_signal(nte, te, np.array([1.0 / 900.0, 1.0 / 1100.0]), np.array([1.0 / 35.0, 1.0 / 80.0]), 2.2, 1.3)

    # ---- P1: penalties
    pens = penalties(npc, T2s, create_Laplacian_matrix)
    for k, v in pens.items():
        out["L_" + k] = v
    lam_grid = lambda_grid()
    out["lambda_grid"] = lam_grid

    # ---- voxels (FA 150 -> index 60 on the 91-grid), normalised by first echo as V1 does
    D = np.ascontiguousarray(Dic_full[:, :, 60])
    out["D150"] = D
    data, _ = synth_voxels(rng, nvox, nte, epg_signal, fa_deg=150.0)
    out["data"] = data
    M = data / data[:, :1]

    # N1: plain NNLS
    xs = np.zeros((nvox, npc)); rn = np.zeros(nvox)
    for v in range(nvox):
        xs[v], rn[v] = alg.nnls(D, M[v])
    out["nnls_x"], out["nnls_rnorm"] = xs, rn

    tik_lams = np.array([1e-6, 1e-3, 1e-1, 1.8])
    out["tik_lams"] = tik_lams
    for name, L in pens.items():
        # N2
        tik = np.zeros((len(tik_lams), nvox, npc))
        for i, lam in enumerate(tik_lams):
            for v in range(nvox):
                tik[i, v] = alg.nnls_tik(D, M[v], L, lam)
        out["tik_" + name] = tik
        # X1
        f = np.zeros((nvox, npc)); lam = np.zeros(nvox); kest = np.zeros(nvox)
        for v in range(nvox):
            f[v], lam[v], kest[v] = alg.nnls_x2(D, M[v], L, 1.02)
        out["x2_f_" + name], out["x2_lam_" + name], out["x2_kest_" + name] = f, lam, kest
        # LC
        lc = np.zeros(nvox); lcf = np.zeros((nvox, npc))
        for v in range(nvox):
            lc[v] = alg.nnls_lcurve_wrapper(D, M[v], L, lam_grid)
            lcf[v] = alg.nnls_tik(D, M[v], L, lc[v])
        out["lc_lam_" + name], out["lc_f_" + name] = lc, lcf
        # GC / BR on fewer voxels at the large shape (slow)
        ns = nvox_slow
        gf = np.zeros((ns, npc)); gl = np.zeros(ns); bf = np.zeros((ns, npc)); bl = np.zeros(ns)
        for v in range(ns):
            gf[v], gl[v] = alg.nnls_gcv(D, M[v], L)
            with np.errstate(all="ignore"):
                bf[v], bl[v] = bay.BayesReg_nnls(D, M[v], L)
        out["gcv_f_" + name], out["gcv_lam_" + name] = gf, gl
        out["bayes_f_" + name], out["bayes_lam_" + name] = bf, bl

    # ---- L-curve internals on one voxel/L2: the two log curves and the corner
    L = pens["L2"]
    le = np.zeros(50); ln = np.zeros(50)
    for i, lam in enumerate(lam_grid):
        x = alg.nnls_tik(D, M[0], L, lam)
        le[i] = np.log(np.sum((D @ x - M[0]) ** 2) + 1e-200)
        ln[i] = np.log(np.sum((L @ x) ** 2) + 1e-200)
    out["lc_logerr0"], out["lc_lognorm0"] = le, ln
    out["lc_corner0"] = alg.select_corner(le, ln)
    xs_, ys_ = alg.scale_curve(le, ln)
    out["lc_scaled0"] = np.stack([xs_, ys_])

    # ---- Brent traces (objective values at every evaluated lambda) for 4 voxels
    ntr = 4
    for meth, mod, objname, call in (
        ("x2", alg, "obj_nnls_x2", lambda v, L: alg.nnls_x2(D, M[v], L, 1.02)),
        ("gcv", alg, "obj_nnls_gcv", lambda v, L: alg.nnls_gcv(D, M[v], L)),
        ("bayes", bay, "obj_BayesReg_nnls", lambda v, L: bay.BayesReg_nnls(D, M[v], L)),
    ):
        for name in ("I", "L2", "InvT2"):
            txs, tfs = [], []
            for v in range(ntr):
                orig = getattr(mod, objname)
                tr = Trace(orig)
                setattr(mod, objname, tr)
                try:
                    with np.errstate(all="ignore"):
                        call(v, pens[name])
                finally:
                    setattr(mod, objname, orig)
                txs.append(tr.xs); tfs.append(tr.fs)
            out["trace_%s_%s_x" % (meth, name)] = pad_traces(txs)
            out["trace_%s_%s_f" % (meth, name)] = pad_traces(tfs)

    # ---- GCV / BayesReg objective values on a fixed lambda grid (2 voxels, all penalties)
    og = np.array([1e-8, 1e-6, 1e-4, 1e-3, 1e-2, 0.1, 0.5, 1.0, 1.9, 5.0])
    out["obj_grid"] = og
    m = nte
    for name, L in pens.items():
        gv = np.zeros((2, len(og))); bv = np.zeros((2, len(og)))
        for v in range(2):
            Maug = np.concatenate((M[v], np.zeros(npc)))
            x0, _ = bay.nnls(D, M[v])
            dof = np.max([m - np.sum(x0 > 0), 1.0])
            beta = 1.0 / (np.sqrt(np.sum((M[v] - D @ x0) ** 2) / dof)) ** 2
            from scipy.linalg import det
            B = D.T @ D; K = L.T @ L; detL = det(L)
            for i, lam in enumerate(og):
                gv[v, i] = alg.obj_nnls_gcv(lam, D, L, Maug, m, np.eye(m))
                with np.errstate(all="ignore"):
                    bv[v, i] = bay.obj_BayesReg_nnls(lam, D, L, Maug, M[v], m, npc, B, detL, beta, K)
        out["gcvobj_" + name], out["bayesobj_" + name] = gv, bv

    # ---- F1: brute-force FA on un-normalised data with per-voxel true FA
    nfa_vox = max(6, nvox // 4)
    dfa, fa_true = synth_voxels(rng, nfa_vox, nte, epg_signal, fa_deg=None)
    idx = np.zeros(nfa_vox); al = np.zeros(nfa_vox); km = np.zeros(nfa_vox); sse = np.zeros(nfa_vox)
    ff = np.zeros((nfa_vox, npc))
    for v in range(nfa_vox):
        idx[v], al[v], km[v], sse[v], ff[v] = compute_optimal_FA(dfa[v], Dic_full, alpha_values)
    out.update(fa_data=dfa, fa_true=fa_true, fa_idx=idx, fa_alpha=al, fa_km=km, fa_sse=sse, fa_f=ff)

    # ---- V1: fitting_slice_T2 rows (gating + per-voxel FA index), every method
    nx = 12
    row, _ = synth_voxels(rng, nx, nte, epg_signal, fa_deg=None)
    mask = np.ones(nx); mask[2] = 0.0
    row[5, :] = 0.0                     # sum(M) == 0 gate
    row[7, 0] = 0.0                     # M[0] == 0 gate
    fa_index = rng.integers(0, 91, size=nx).astype(np.float64)
    out.update(row_data=row, row_mask=mask, row_fa_index=fa_index)
    for meth, pen in (("NNLS", "I"), ("T2SPARC", "InvT2"), ("X2", "L2"), ("X2", "I"), ("L_curve", "L1"),
                      ("GCV", "L2"), ("BayesReg", "InvT2"), ("BayesReg", "L2")):
        with np.errstate(all="ignore"):
            fs, sg, rg = fitting_slice_T2(mask, row, fa_index, nx, Dic_full, lam_grid, npc, nte, meth, pens[pen], None)
        out["row_%s_%s_fsol" % (meth, pen)] = fs
        out["row_%s_%s_sig" % (meth, pen)] = sg
        out["row_%s_%s_reg" % (meth, pen)] = rg

    # ---- degenerate: noise-free voxel (SNR inf) -> Brent end points
    nf, _ = synth_voxels(rng, 1, nte, epg_signal, fa_deg=150.0, snr_lo=np.inf, snr_hi=np.inf)
    Mnf = nf[0] / nf[0, 0]
    out["nf_M"] = Mnf
    with np.errstate(all="ignore"):
        f, lam, kest = alg.nnls_x2(D, Mnf, pens["I"], 1.02)
        out["nf_x2_lam"], out["nf_x2_kest"] = lam, kest
        out["nf_lc_lam"] = alg.nnls_lcurve_wrapper(D, Mnf, pens["I"], lam_grid)
        out["nf_bayes_lam"] = bay.BayesReg_nnls(D, Mnf, pens["I"])[1]

    np.savez_compressed(os.path.join(HERE, "golden_%s.npz" % tag), **out)
    print("wrote golden_%s.npz (%d arrays)" % (tag, len(out)))


def gen_motor(tag, reg_method, reg_matrix, fa_method, seed):
    """End-to-end reference driver on a tiny in-memory volume (metrics M1 + FA + V1 together)."""
    from epg.epg import epg_signal
    import motor.motor_recon_met2_real_data as motor
    import matplotlib
    matplotlib.rcParams["text.usetex"] = False      # the driver asks for LaTeX text; none in this image

    rng = np.random.default_rng(seed)
    nx, ny, nz, nte = 5, 4, 2, 32
    data, _ = synth_voxels(rng, nx * ny * nz, nte, epg_signal, fa_deg=None)
    data = data.reshape(nx, ny, nz, nte)
    mask = np.ones((nx, ny, nz))
    mask[0, 0, 0] = 0
    mask[4, 3, 1] = 0
    data[1, 1, 0, :] = 0.0          # masked-in but empty voxel -> metrics of an all-zero spectrum
    _NIB_FILES["in_data"] = data
    _NIB_FILES["in_mask"] = mask
    TE = 10.0 * np.arange(1, nte + 1)
    os.makedirs("/tmp/met2_golden_png", exist_ok=True)
    prefix = "/tmp/met2_golden_png/" + tag + "_"
    motor.motor_recon_met2(TE, "in_data", "in_mask", prefix, 3000.0, reg_method, reg_matrix, "None", fa_method, "no", 40.0, 1)
    out = {"data": data, "mask": mask, "TE": TE}
    for name in ("MWF", "IEWF", "FWF", "T2_M", "T2_IE", "TWC", "FA", "fsol_4D", "Est_Signal", "reg_param"):
        out[name] = _NIB_FILES[prefix + name + ".nii.gz"]
    np.savez_compressed(os.path.join(HERE, "golden_motor_%s.npz" % tag), **out)
    print("wrote golden_motor_%s.npz" % tag)


def gen_nesma(seed):
    """The NESMA filter of motor:305-333 is inline in the driver.  The driver is run with denoise='NESMA' on a small
    volume and the rows handed to the FA step (motor:354-357: `data_smooth = data.copy()` after the filter) are
    recorded, which gives the filtered volume itself next to the end-to-end outputs."""
    from epg.epg import epg_signal
    import motor.motor_recon_met2_real_data as motor
    import matplotlib
    matplotlib.rcParams["text.usetex"] = False
    rng = np.random.default_rng(seed)
    nx, ny, nz, nte = 9, 8, 5, 32
    # piecewise-constant tissue classes so that the similarity test (RE < 2.5 %) has neighbours to accept
    cls, _ = synth_voxels(rng, 3, nte, epg_signal, fa_deg=150.0, snr_lo=np.inf, snr_hi=np.inf)
    lab = rng.integers(0, 3, (nx, ny, nz))
    data = cls[lab] * (1.0 + 0.004 * rng.standard_normal((nx, ny, nz, nte)))
    mask = np.ones((nx, ny, nz)); mask[0, 0, :] = 0; mask[5, 5, 2] = 0
    mask[8, 7, 4] = 2                     # a mask value other than 1 is skipped by the filter (motor:317)
    _NIB_FILES["nes_data"] = data
    _NIB_FILES["nes_mask"] = mask
    rows = {}
    orig = motor.fitting_slice_FA_brute_force
    state = {"z": -1, "y": 0}

    def spy(mask_1d, data_1d, nx_, Dic_3D, alpha_values):
        if state["y"] == 0:
            state["z"] += 1
        rows[(state["y"], state["z"])] = np.array(data_1d)
        state["y"] = (state["y"] + 1) % ny
        return orig(mask_1d, data_1d, nx_, Dic_3D, alpha_values)

    motor.fitting_slice_FA_brute_force = spy
    TE = 10.0 * np.arange(1, nte + 1)
    os.makedirs("/tmp/met2_golden_png", exist_ok=True)
    prefix = "/tmp/met2_golden_png/nesma_"
    try:
        with np.errstate(all="ignore"):
            motor.motor_recon_met2(TE, "nes_data", "nes_mask", prefix, 3000.0, "X2", "L2", "NESMA", "brute-force", "no", 40.0, 1)
    finally:
        motor.fitting_slice_FA_brute_force = orig
    den = np.zeros_like(data)
    for (y, z), r in rows.items():
        den[:, y, z, :] = r
    out = {"data": data, "mask": mask, "TE": TE, "denoised": den}
    for name in ("MWF", "FA", "fsol_4D", "reg_param"):
        out[name] = _NIB_FILES[prefix + name + ".nii.gz"]
    np.savez_compressed(os.path.join(HERE, "golden_nesma.npz"), **out)
    print("wrote golden_nesma.npz")


def gen_smooth(seed):
    """The CLI's default pipeline (run_real_data_script.py:32-46: FA_method='spline', FA_smooth='yes', denoise='None', X2):
    the driver smooths every echo volume with scipy.ndimage.gaussian_filter(sigma=2) for the FA step only
    (motor:337-343).  The rows handed to the spline FA step are recorded (= the smoothed volume) next to the outputs."""
    from epg.epg import epg_signal
    import motor.motor_recon_met2_real_data as motor
    import matplotlib
    matplotlib.rcParams["text.usetex"] = False
    rng = np.random.default_rng(seed)
    nx, ny, nz, nte = 7, 6, 5, 32
    data, _ = synth_voxels(rng, nx * ny * nz, nte, epg_signal, fa_deg=None)
    data = data.reshape(nx, ny, nz, nte)
    mask = np.ones((nx, ny, nz)); mask[0, 0, 0] = 0; mask[6, 5, 4] = 0; mask[3, 2, :] = 0
    _NIB_FILES["sm_data"] = data
    _NIB_FILES["sm_mask"] = mask
    rows = {}
    orig = motor.fitting_slice_FA_spline_method
    state = {"z": -1, "y": 0}

    def spy(Dic_3D_LR, Dic_3D, data_1d, mask_1d, alpha_values_spline, nx_, alpha_values):
        if state["y"] == 0:
            state["z"] += 1
        rows[(state["y"], state["z"])] = np.array(data_1d)
        state["y"] = (state["y"] + 1) % ny
        return orig(Dic_3D_LR, Dic_3D, data_1d, mask_1d, alpha_values_spline, nx_, alpha_values)

    motor.fitting_slice_FA_spline_method = spy
    TE = 10.0 * np.arange(1, nte + 1)
    os.makedirs("/tmp/met2_golden_png", exist_ok=True)
    prefix = "/tmp/met2_golden_png/smooth_"
    try:
        with np.errstate(all="ignore"):
            motor.motor_recon_met2(TE, "sm_data", "sm_mask", prefix, 3000.0, "X2", "L2", "None", "spline", "yes", 40.0, 1)
    finally:
        motor.fitting_slice_FA_spline_method = orig
    sm = np.zeros_like(data)
    for (y, z), r in rows.items():
        sm[:, y, z, :] = r
    out = {"data": data, "mask": mask, "TE": TE, "smoothed": sm}
    for name in ("MWF", "IEWF", "FWF", "T2_M", "T2_IE", "TWC", "FA", "fsol_4D", "Est_Signal", "reg_param"):
        out[name] = _NIB_FILES[prefix + name + ".nii.gz"]
    np.savez_compressed(os.path.join(HERE, "golden_motor_default_smooth.npz"), **out)
    print("wrote golden_motor_default_smooth.npz", sm.shape, float(np.abs(sm - data * mask[..., None]).max()))


def gen_roi(seed):
    """The ROI-mode driver (motor/motor_recon_met2_real_data_ROI.py:152-498) on a small in-memory volume with three ROI labels
    (one of them partly outside the mask).  joypy (absent) is replaced by a stub that draws nothing; `np.int` (removed from
    numpy 1.24) is aliased to int; the tables the driver writes with np.savetxt are recorded."""
    from epg.epg import epg_signal
    import matplotlib
    import matplotlib.pyplot as plt
    matplotlib.rcParams["text.usetex"] = False
    if "joypy" not in sys.modules:
        jp = types.ModuleType("joypy")
        jp.joyplot = lambda *a, **k: (plt.figure(), None)
        sys.modules["joypy"] = jp
    if not hasattr(np, "int"):
        np.int = int
    import motor.motor_recon_met2_real_data_ROI as roi_mod
    rng = np.random.default_rng(seed)
    nx, ny, nz, nte = 6, 5, 3, 32
    data, _ = synth_voxels(rng, nx * ny * nz, nte, epg_signal, fa_deg=None)
    data = data.reshape(nx, ny, nz, nte)
    mask = np.ones((nx, ny, nz)); mask[0, 0, :] = 0; mask[5, 4, 2] = 0
    rois = np.zeros((nx, ny, nz))
    rois[0:3, :, :] = 3            # label 3 includes masked-out voxels (0,0,:)
    rois[3:5, 0:3, :] = 7
    rois[5, :, 0:2] = 12
    _NIB_FILES["roi_data"] = data
    _NIB_FILES["roi_mask"] = mask
    _NIB_FILES["roi_rois"] = rois
    saved = {}
    orig_savetxt = np.savetxt

    def spy_savetxt(fname, X, *a, **k):
        saved[str(fname)] = np.array(X, dtype=object)
        return orig_savetxt(fname, X, *a, **k)

    TE = 10.0 * np.arange(1, nte + 1)
    os.makedirs("/tmp/met2_golden_png/roi", exist_ok=True)
    prefix = "/tmp/met2_golden_png/roi/"
    fa_rows = {}
    orig_fa = roi_mod.fitting_slice_FA_brute_force
    state = {"z": -1, "y": 0}

    def spy_fa(mask_1d, data_1d, nx_, Dic_3D, alpha_values):
        if state["y"] == 0:
            state["z"] += 1
        r = orig_fa(mask_1d, data_1d, nx_, Dic_3D, alpha_values)
        fa_rows[(state["y"], state["z"])] = np.array(r[1])
        state["y"] = (state["y"] + 1) % ny
        return r

    np.savetxt = spy_savetxt
    roi_mod.fitting_slice_FA_brute_force = spy_fa
    try:
        with np.errstate(all="ignore"):
            roi_mod.motor_recon_met2_ROIs(TE, "roi_data", "roi_mask", "roi_rois", prefix, 3000.0, "L2", "None", "brute-force", "no", 40.0, 1)
    finally:
        np.savetxt = orig_savetxt
        roi_mod.fitting_slice_FA_brute_force = orig_fa
    fa_index = np.zeros((nx, ny, nz))
    for (y, z), r in fa_rows.items():
        fa_index[:, y, z] = r
    out = {"data": data, "mask": mask, "rois": rois, "TE": TE, "FA_index": fa_index,
           "table_MWF": saved[prefix + "table_MWF.csv"].astype(np.float64),
           "table_Spectra": saved[prefix + "table_Spectra.csv"].astype(np.float64),
           "ROI_labels": saved[prefix + "ROI_labels.csv"].astype(np.float64)}
    for lab in out["ROI_labels"]:
        tv = saved[prefix + "ROI_%.0f/table_values.csv" % lab]
        out["values_%d" % int(lab)] = np.array([float(v) for v in tv[:, 1]])
    np.savez_compressed(os.path.join(HERE, "golden_roi.npz"), **out)
    print("wrote golden_roi.npz", out["ROI_labels"], out["table_MWF"])


# --------------------------------------------------------------------------- large "tail" fixtures
_TAIL = {}


def _tail_worker(args):
    """One chunk of voxels through the reference's own functions (runs in a forked worker)."""
    lo, hi = args
    import intravoxel_algorithms.algorithms as alg
    import intravoxel_algorithms.bayesian_interpolation as bay
    D, M, pens, lam_grid, methods = _TAIL["D"], _TAIL["M"], _TAIL["pens"], _TAIL["lam_grid"], _TAIL["methods"]
    npc = D.shape[1]
    res = {}
    for meth, pen, nmax in methods:
        L = pens[pen]
        a, b = lo, min(hi, nmax)
        n = max(0, b - a)
        f = np.zeros((n, npc)); lam = np.zeros(n); aux = np.zeros(n)
        for i in range(n):
            v = a + i
            with np.errstate(all="ignore"):
                if meth == "NNLS":
                    f[i], aux[i] = alg.nnls(D, M[v])
                elif meth == "X2":
                    f[i], lam[i], aux[i] = alg.nnls_x2(D, M[v], L, 1.02)
                elif meth == "L_curve":
                    lam[i] = alg.nnls_lcurve_wrapper(D, M[v], L, lam_grid)
                    f[i] = alg.nnls_tik(D, M[v], L, lam[i])
                elif meth == "GCV":
                    f[i], lam[i] = alg.nnls_gcv(D, M[v], L)
                elif meth == "BayesReg":
                    f[i], lam[i] = bay.BayesReg_nnls(D, M[v], L)
        res[(meth, pen)] = (a, f, lam, aux)
    return res


def gen_tail(tag, nte, npc, nvox, methods, seed, procs=7, chunk=32):
    """A few thousand voxels through the reference for the methods whose parity has a tail (X2: Brent ties;
    BayesReg/InvT2: flat evidence minimum; GCV: staircase objective), so that rates of 1e-4 are visible.
    Same recipe and dictionary (FA 150) as gen_shape; data normalised by the first echo as fitting_slice_T2 does."""
    import multiprocessing as mp
    from epg.epg import create_Dic_3D, epg_signal
    from motor.motor_recon_met2_real_data import create_Laplacian_matrix

    rng = np.random.default_rng(seed)
    T2s = t2_grid(npc)
    T1s = 1000.0 * np.ones_like(T2s)
    TR, te = 3000.0, 10.0
    D = np.ascontiguousarray(create_Dic_3D(npc, T2s, T1s, nte, te, np.array([150.0]), TR)[:, :, 0])
    data, _ = synth_voxels(rng, nvox, nte, epg_signal, fa_deg=150.0)
    M = data / data[:, :1]
    pens = penalties(npc, T2s, create_Laplacian_matrix)
    _TAIL.update(D=D, M=M, pens=pens, lam_grid=lambda_grid(), methods=methods)
    jobs = [(lo, min(nvox, lo + chunk)) for lo in range(0, nvox, chunk)]
    out = {"T2s": T2s, "T1s": T1s, "TR": TR, "tau": te, "nte": nte, "npc": npc, "D150": D, "data": data,
           "lambda_grid": lambda_grid()}
    acc = {}
    for meth, pen, nmax in methods:
        acc[(meth, pen)] = (np.zeros((nmax, npc)), np.zeros(nmax), np.zeros(nmax))
    with mp.get_context("fork").Pool(procs) as pool:
        for k, res in enumerate(pool.imap_unordered(_tail_worker, jobs)):
            for key, (a, f, lam, aux) in res.items():
                F, Lm, A = acc[key]
                F[a:a + f.shape[0]] = f; Lm[a:a + f.shape[0]] = lam; A[a:a + f.shape[0]] = aux
            if k % 16 == 0:
                print("  tail %s: chunk %d / %d" % (tag, k + 1, len(jobs)), flush=True)
    for (meth, pen), (F, Lm, A) in acc.items():
        base = "%s_%s" % (meth, pen)
        out[base + "_f"] = F
        if meth != "NNLS":
            out[base + "_lam"] = Lm
        if meth in ("X2", "NNLS"):
            out[base + "_aux"] = A            # k_est (X2) / residual norm (NNLS)
    np.savez_compressed(os.path.join(HERE, "golden_tail_%s.npz" % tag), **out)
    print("wrote golden_tail_%s.npz (%d arrays)" % (tag, len(out)))


def _x2_worker(args):
    """nnls_x2 of the reference on one chunk of voxels (forked worker)."""
    lo, hi = args
    import intravoxel_algorithms.algorithms as alg
    D, M, L = _TAIL["D"], _TAIL["M"], _TAIL["L"]
    n = hi - lo
    f = np.zeros((n, D.shape[1])); lam = np.zeros(n); kest = np.zeros(n)
    for i in range(n):
        with np.errstate(all="ignore"):
            f[i], lam[i], kest[i] = alg.nnls_x2(D, M[lo + i], L, 1.02)
    return lo, f, lam, kest


def _x2_reference(D, M, L, procs=7, chunk=64, tag=""):
    import multiprocessing as mp
    _TAIL.update(D=D, M=M, L=L)
    nvox = M.shape[0]
    F = np.zeros((nvox, D.shape[1])); Lm = np.zeros(nvox); Ke = np.zeros(nvox)
    jobs = [(lo, min(nvox, lo + chunk)) for lo in range(0, nvox, chunk)]
    with mp.get_context("fork").Pool(procs) as pool:
        for k, (lo, f, lam, kest) in enumerate(pool.imap_unordered(_x2_worker, jobs)):
            F[lo:lo + f.shape[0]] = f; Lm[lo:lo + f.shape[0]] = lam; Ke[lo:lo + f.shape[0]] = kest
            if k % 64 == 0:
                print("  %s: chunk %d / %d" % (tag, k + 1, len(jobs)), flush=True)
    return F, Lm, Ke


def gen_tail_x2(nvox, seed, procs=7, nte=32, npc=60, name="golden_tail_X2.npz"):
    """configs[1]'s method (X2/L2, 32 x 60, FA 150) on enough voxels through the reference's nnls_x2 (algorithms.py:211-233) that
    a 6e-5 tail is countable.  To keep the file small the signals are rounded to float32 BEFORE they go through the reference
    (so the stored float32 array IS the input, exactly) and the reference's spectra are stored as float32 (relative rounding
    6e-8, far inside the 1e-5 tolerance); lambda, k_est and the MWF of the float64 spectra stay float64."""
    from epg.epg import create_Dic_3D, epg_signal
    from motor.motor_recon_met2_real_data import create_Laplacian_matrix
    rng = np.random.default_rng(seed)
    T2s = t2_grid(npc); T1s = 1000.0 * np.ones_like(T2s)
    D = np.ascontiguousarray(create_Dic_3D(npc, T2s, T1s, nte, 10.0, np.array([150.0]), 3000.0)[:, :, 0])
    data, _ = synth_voxels(rng, nvox, nte, epg_signal, fa_deg=150.0)
    data = data.astype(np.float32).astype(np.float64)
    M = data / data[:, :1]
    L = penalties(npc, T2s, create_Laplacian_matrix)["L2"]
    F, Lm, Ke = _x2_reference(D, M, L, procs=procs, tag="tail X2/L2 %dx%d" % (nte, npc))
    mwf = F[:, T2s <= 40.0].sum(axis=1) / (F.sum(axis=1) + 1e-16)
    np.savez_compressed(os.path.join(HERE, name), T2s=T2s, T1s=T1s, TR=3000.0, tau=10.0, nte=nte, npc=npc, D150=D,
                        data=data.astype(np.float32), X2_L2_f=F.astype(np.float32), X2_L2_lam=Lm, X2_L2_aux=Ke, X2_L2_mwf=mwf)
    print("wrote %s (%d voxels)" % (name, nvox))


def gen_x2_failset(path):
    """The voxels of configs[1] where the HIP path and the oracle differ by more than 1e-5 (bench.py --dump-fail on the GPU box),
    through the reference's own nnls_x2: which of the two, if either, does the reference side with?"""
    from motor.motor_recon_met2_real_data import create_Laplacian_matrix
    from epg.epg import create_Dic_3D
    z = np.load(path)
    nte, npc = z["data"].shape[1], z["got"].shape[1]
    T2s = t2_grid(npc); T1s = 1000.0 * np.ones_like(T2s)
    D = np.ascontiguousarray(create_Dic_3D(npc, T2s, T1s, nte, 10.0, np.array([150.0]), 3000.0)[:, :, 0])
    L = penalties(npc, T2s, create_Laplacian_matrix)["L2"]
    data = z["data"]
    M = data / data[:, :1]
    F, Lm, Ke = _x2_reference(D, M, L, procs=min(7, max(1, data.shape[0])), chunk=4, tag="X2 fail set")
    out = {k: z[k] for k in z.files}
    out.update(T2s=T2s, D150=D, ref_f=F * data[:, :1], ref_lam=Lm, ref_kest=Ke)
    np.savez_compressed(os.path.join(HERE, "golden_x2_failset.npz"), **out)
    print("wrote golden_x2_failset.npz (%d voxels)" % data.shape[0])


def main():
    install_shims()
    which = sys.argv[1:] or ["S1", "S2", "motor", "nesma", "smooth", "roi"]
    if "S1" in which:
        gen_shape("S1", 32, 60, nvox=32, nvox_slow=32, seed=20260101, with_fa_full=True)
    if "S2" in which:
        gen_shape("S2", 48, 120, nvox=8, nvox_slow=4, seed=20260105, with_fa_full=False)
    if "motor" in which:
        gen_motor("x2_l2_bf", "X2", "L2", "brute-force", 20260111)
        gen_motor("lcurve_l1_spline", "L_curve", "L1", "spline", 20260112)
    if "nesma" in which:
        gen_nesma(20260113)
    if "smooth" in which:
        gen_smooth(20260114)
    if "roi" in which:
        gen_roi(20260115)
    # not in the default list: ~15 minutes on 7 processes
    if "tailS1" in which:
        gen_tail("S1", 32, 60, 4096, [("NNLS", "I", 4096), ("X2", "L2", 4096), ("X2", "I", 1024), ("L_curve", "L1", 4096),
                                      ("BayesReg", "InvT2", 4096), ("BayesReg", "I", 1024), ("GCV", "L2", 4096)], 20260121)
    if "tailX2" in which:        # ~8 minutes on 7 processes
        gen_tail_x2(65536, 20260123)
    if "tailX2S2" in which:      # the same at config 5's shape (48 x 120: the two-bins-per-lane kernels), 8 192 voxels; ~10 minutes on 7 processes
        gen_tail_x2(8192, 20260124, nte=48, npc=120, name="golden_tail_X2_S2.npz")
    if "x2fail" in which:        # needs gpurun_out/fail_x2l2.npz (bench.py --config 1 --dump-fail on the GPU box)
        gen_x2_failset(os.environ.get("MET2_FAILSET", os.path.join(HERE, "..", "..", "gpurun_out", "fail_x2l2.npz")))
    if "tailS2b" in which:       # BayesReg/I at 48 x 120 on 2 048 voxels (the method with rounding-level parity: pins the panel Cholesky); ~3 minutes
        gen_tail("S2b", 48, 120, 2048, [("BayesReg", "I", 2048)], 20260125)
    if "tailS2" in which:
        gen_tail("S2", 48, 120, 512, [("X2", "L2", 512), ("L_curve", "L1", 512), ("BayesReg", "InvT2", 512), ("GCV", "L2", 512)],
                 20260122)


if __name__ == "__main__":
    main()
