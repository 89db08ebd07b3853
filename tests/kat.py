"""Statistical known-answer test against the only numbers the reference commits for this path: the Monte-Carlo tables of
scripts_synthetic_data_evaluation/Paper_Comparison/Results/SNRs_50_150/All_methods_10000iters/ (10 000 random two-lobe
voxels, SNR 50-150, unseeded):

  table_errors.txt:3-12          column "1. MAE"   mean |MWF_est - MWF_true| per method
  table_regularization.txt:3-12  mean and standard deviation of the selected lambda per method

The recipe is the script's own (evaluate_all_methods_two_lobes_SNR50_150.py): parameters :156-170, signal :385 (EPG on a
1000-point T2 grid, Km = 1000), Rician noise :387-391, flip angle by compute_optimal_FA over the 91-grid :398, normalisation by
the first echo :401-402, the true MWF from the high-resolution pdf re-binned on the 60-point grid :404-428, methods :436-628
(X2 factor 1.02; L-curve grid 0 + logspace(1e-8, 100, 49) -- NOT the driver's 10, :214-215), MWF = sum(x[T2<=40]) / sum(x) :76-79.

The draws cannot be the reference's (no seed there), so agreement is statistical: with n voxels here and 10 000 there the
difference of two independent means has standard error sqrt(sd^2/n + sd^2/10000).  Data only -- no reference code.
"""
import math

import numpy as np

# (method label in the tables, reg_method, reg_matrix, MAE(MWF) table_errors.txt, mean lambda, STD lambda table_regularization.txt)
REF_ROWS = [
    ("1. NNLS", "NNLS", "I", 0.0679834, 0.0, 0.0),
    ("2. X2-I", "X2", "I", 0.0548569, 0.00257832, 0.00448045),
    ("3. X2-L1", "X2", "L1", 0.0557831, 0.0224889, 0.0656093),
    ("4. X2-L2", "X2", "L2", 0.0556009, 0.179334, 0.642022),
    ("5. Lcurve-I", "L_curve", "I", 0.0543839, 0.00543128, 0.00587748),
    ("6. Lcurve-L1", "L_curve", "L1", 0.0568595, 0.127913, 0.132735),
    ("7. Lcurve-L2", "L_curve", "L2", 0.0558122, 0.264592, 0.267592),
    ("8. GCV-I", "GCV", "I", 0.0581128, 0.000774211, 0.00273344),
    ("9. GCV-L1", "GCV", "L1", 0.0587813, 0.107992, 0.226097),
    ("10. GCV-L2", "GCV", "L2", 0.0598534, 0.865506, 1.3048),
]
N_REF = 10000
# standard deviation of |residual| per voxel: from RMSE and MAE of the same tables, sd^2 = RMSE^2 - MAE^2
REF_RMSE = {"1. NNLS": 0.088805, "2. X2-I": 0.0686559, "3. X2-L1": 0.0694512, "4. X2-L2": 0.0692368, "5. Lcurve-I": 0.0661569,
            "6. Lcurve-L1": 0.0680126, "7. Lcurve-L2": 0.0672535, "8. GCV-I": 0.0744071, "9. GCV-L1": 0.0717533, "10. GCV-L2": 0.0720697}


def kat_lambda_grid():
    """evaluate_all_methods_two_lobes_SNR50_150.py:212-215"""
    lam = np.zeros(50)
    lam[1:] = np.logspace(math.log10(1e-8), math.log10(100.0), num=49, endpoint=True, base=10.0)
    return lam


def t2_grid(npc=60):
    return np.logspace(math.log10(10.0), math.log10(2000.0), num=npc, endpoint=True, base=10.0)


def true_mwf(par, T2s, cut=40.0):
    """The script's `True_fM` (:404-428): the generating two-Gaussian pdf on linspace(1, 300, 1000), integrated over the bins
    of the 60-point grid (bin edges half-way between grid points), normalised, summed over T2 <= cut."""
    T2g, dT = np.linspace(1.0, 300.0, 1000, retstep=True)
    n = par["MWF"].shape[0]
    pdf = lambda mu, sg: np.exp(-0.5 * ((T2g[None, :] - mu[:, None]) / sg[:, None]) ** 2) / (sg[:, None] * math.sqrt(2.0 * math.pi))
    out = np.zeros(n)
    npc = T2s.shape[0]
    edges = T2s[:-1] + (T2s[1:] - T2s[:-1]) / 2.0
    # bin of every high-resolution point: 0 for T2g < edges[0], i for edges[i-1] <= T2g < edges[i], npc-1 above the last edge
    which = np.searchsorted(edges, T2g, side="right")
    for s in range(0, n, 2048):
        e = min(n, s + 2048)
        dist = par["MWF"][s:e, None] * pdf(par["T2m"][s:e], par["sm"][s:e]) + (1.0 - par["MWF"][s:e, None]) * pdf(par["T2ie"][s:e], par["sie"][s:e])
        dist = dist / dist.sum(axis=1, keepdims=True)
        d2 = np.zeros((e - s, npc))
        for b in range(npc):
            sel = which == b
            if sel.any():
                d2[:, b] = (dist[:, sel] * dT).sum(axis=1)
        d2 = d2 / d2.sum(axis=1, keepdims=True)
        out[s:e] = d2[:, T2s <= cut].sum(axis=1)
    return out


def tolerance(label, n_here, what):
    """4 standard errors of the difference of two independent means (n_here voxels here, 10 000 in the table)."""
    row = [r for r in REF_ROWS if r[0] == label][0]
    if what == "mae":
        sd = math.sqrt(max(REF_RMSE[label] ** 2 - row[3] ** 2, 0.0))
    else:
        sd = row[5]
    return 4.0 * sd * math.sqrt(1.0 / n_here + 1.0 / N_REF)
