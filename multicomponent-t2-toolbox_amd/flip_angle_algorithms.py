"""Drop-ins for flip_angle_algorithms/fa_estimation.py (brute-force path)."""
import numpy as np
import torch

from ._cache import plan_for


def _fa_batch(plan, data, mask):
    fa, km, _ = plan.fa_bruteforce(torch.as_tensor(np.ascontiguousarray(data, dtype=np.float64), device=plan.device),
                                   None if mask is None else torch.as_tensor(np.asarray(mask) > 0, device=plan.device))
    return fa.cpu().numpy(), km.cpu().numpy()


def compute_optimal_FA(M, Dic_3D, alpha_values):
    """fa_estimation.py:74-90 -> (index, alpha, km, SSE, f)"""
    M = np.ascontiguousarray(M, dtype=np.float64)
    if not np.isfinite(M).all():
        raise ValueError("array must not contain infs or NaNs")
    plan = plan_for(Dic_3D)
    idx, km = _fa_batch(plan, M[None, :], None)
    i = int(idx[0])
    # f and SSE at the selected flip angle: one plain NNLS with that kernel
    out = plan.fit("NNLS", torch.as_tensor(M[None, :], device=plan.device), fa_index=torch.tensor([float(i)], device=plan.device),
                   want_maps=False)
    f = out["fsol"][0].cpu().numpy()
    sse = float(np.sum((out["sig"][0].cpu().numpy() - M) ** 2))
    return i, alpha_values[i], float(np.sum(f)), sse, f


def fitting_slice_FA_brute_force(mask_1d, data_1d, nx, Dic_3D, alpha_values):
    """fa_estimation.py:92-111 -> (FA[nx], FA_index[nx], KM[nx], sum of spectra).  (The reference's own
    function raises NameError on Python 3 -- `xrange`, fa_estimation.py:99; this is its intended result.)"""
    plan = plan_for(Dic_3D)
    data = np.ascontiguousarray(data_1d, dtype=np.float64)
    idx, km = _fa_batch(plan, data, mask_1d)
    fitted = (np.asarray(mask_1d) > 0) & (data.sum(axis=1) > 0)
    FA = np.where(fitted, np.asarray(alpha_values)[idx.astype(int)], 0.0)
    out = plan.fit("NNLS", torch.as_tensor(data, device=plan.device), fa_index=torch.as_tensor(idx, device=plan.device),
                   mask=torch.as_tensor(fitted, device=plan.device), want_maps=False)
    # NB the FA step does not normalise and does not require M[0] > 0 (fa_estimation.py:100); voxels with
    # M[0] <= 0 but sum > 0 keep their FA index and contribute no spectrum here.
    fsum = out["fsol"].sum(dim=0).cpu().numpy()
    return FA, np.where(fitted, idx, 0.0), np.where(fitted, km, 0.0), fsum


def fitting_slice_FA_spline_method(Dic_3D_LR, Dic_3D, data_1d, mask_1d, alpha_values_spline, nx, alpha_values):
    """fa_estimation.py:35-70 -> (FA[nx], FA_index[nx], KM[nx], sum of spectra)"""
    plan_lr = plan_for(Dic_3D_LR)
    plan = plan_for(Dic_3D)
    data = np.ascontiguousarray(data_1d, dtype=np.float64)
    dd = torch.as_tensor(data, device=plan.device)
    mk = torch.as_tensor(np.asarray(mask_1d) > 0, device=plan.device)
    fa, km, _ = plan.fa_spline(plan_lr, alpha_values_spline, alpha_values, dd, mk)
    idx = fa.cpu().numpy()
    fitted = (np.asarray(mask_1d) > 0) & (data.sum(axis=1) > 0)
    FA = np.where(fitted, np.asarray(alpha_values)[idx.astype(int)], 0.0)
    out = plan.fit("NNLS", dd, fa_index=fa, mask=torch.as_tensor(fitted, device=plan.device), want_maps=False)
    return FA, np.where(fitted, idx, 0.0), np.where(fitted, km.cpu().numpy(), 0.0), out["fsol"].sum(dim=0).cpu().numpy()
