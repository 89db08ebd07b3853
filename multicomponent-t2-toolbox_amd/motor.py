"""Drop-ins for motor/motor_recon_met2_real_data.py: create_Laplacian_matrix, fitting_slice_T2, the NESMA filter,
recon_met2_arrays (the driver's steps 1-4 on in-memory arrays), motor_recon_met2 (the same with the on-disk
contract) and the ROI mode.  TV denoising (scikit-image), plots and the mean-spectrum PNG are not reproduced."""
import math

import numpy as np
import torch

from ._cache import plan_for
from ._lib import check, lib
from .plan import MAP_NAMES, Met2Plan


def create_Laplacian_matrix(Npc, order):
    """motor:86-111 -> dense [Npc, Npc]"""
    L = np.zeros((Npc, Npc))
    i = np.arange(Npc)
    if order == 2:
        L[i, i] = 2.0
        L[i[1:], i[1:] - 1] = -1.0
        L[i[:-1], i[:-1] + 1] = -1.0
        L[0, 0] = 1.0
        L[-1, -1] = 1.0
    elif order == 1:
        L[i, i] = 1.0
        L[i[1:], i[1:] - 1] = -1.0
    elif order == 0:
        L[i, i] = 1.0
    else:
        raise ValueError("order must be 0, 1 or 2")
    return L


def create_InvT2_matrix(T2s):
    """motor:263-269 (reg_matrix == 'InvT2')"""
    T2s = np.asarray(T2s, dtype=np.float64)
    d = T2s - np.concatenate(([T2s[0] - 1.0], T2s[:-1]))
    d[0] = d[1]
    return np.diag(1.0 / d)


def penalty_matrix(reg_matrix, Npc, T2s=None):
    """motor:254-273; unknown names raise (the reference calls sys.exit())."""
    if reg_matrix == "I":
        return create_Laplacian_matrix(Npc, 0)
    if reg_matrix == "L1":
        return create_Laplacian_matrix(Npc, 1)
    if reg_matrix == "L2":
        return create_Laplacian_matrix(Npc, 2)
    if reg_matrix == "InvT2":
        return create_InvT2_matrix(T2s)
    raise ValueError("Error: Wrong reg_matrix option!")


def fitting_slice_T2(mask_1d, data_1d, FA_index_1d, nx, Dic_3D, lambda_reg, T2dim, nEchoes, reg_method, Laplac, dist_x_prior=None):
    """motor:113-162 -> (f_sol[nx, T2dim], signal[nx, nEchoes], reg[nx])"""
    data = np.ascontiguousarray(data_1d, dtype=np.float64)
    if not np.isfinite(data[np.asarray(mask_1d) > 0]).all():
        raise ValueError("array must not contain infs or NaNs")     # algorithms.py:56
    plan = plan_for(Dic_3D, Laplac, lambda_reg)
    dev = plan.device
    out = plan.fit(reg_method, torch.as_tensor(data, device=dev), fa_index=torch.as_tensor(np.asarray(FA_index_1d, dtype=np.float64), device=dev),
                   mask=torch.as_tensor(np.asarray(mask_1d) > 0, device=dev), want_maps=False)
    return out["fsol"].cpu().numpy(), out["sig"].cpu().numpy(), out["reg"].cpu().numpy()


def nesma_filter(data, mask, device=0):
    """The NESMA filter of the driver (motor:305-333) on the device: `data` [nx,ny,nz,nt] as the driver holds it at
    that point (multiplied by the mask, negatives clipped), `mask` [nx,ny,nz]; voxels with mask == 1 become the
    mean of the similar voxels (relative L1 distance < 2.5 %) of their 12^3 window, all others zero.  Accepts numpy
    arrays (returns numpy) or CUDA tensors (returns a tensor on the same device)."""
    as_numpy = not torch.is_tensor(data)
    dev = torch.device("cuda", device) if as_numpy else data.device
    dd = torch.as_tensor(data, dtype=torch.float64, device=dev).contiguous()
    if dd.dim() != 4 or tuple(np.shape(mask)) != tuple(dd.shape[:3]):
        raise ValueError("data must be [nx,ny,nz,nt] and mask [nx,ny,nz]")
    mk = (torch.as_tensor(mask, device=dev) == 1).to(torch.uint8).contiguous()
    out = torch.empty_like(dd)
    nx, ny, nz, nt = dd.shape
    with torch.cuda.device(dev):
        check(lib().met2_nesma(dev.index or 0, nx, ny, nz, nt, dd.data_ptr(), mk.data_ptr(), out.data_ptr(),
                               torch.cuda.current_stream(dev).cuda_stream))
    return out.cpu().numpy() if as_numpy else out


def gaussian_smooth(data, sigma=2.0, truncate=4.0, device=0):
    """The Gaussian pre-smoothing of the FA step (motor:337-343): every echo volume of data [nx,ny,nz,nt] through the
    equivalent of scipy.ndimage.gaussian_filter(volume, sigma) (mode 'reflect', truncate 4), on the device, bit-identical
    to scipy.  numpy in -> numpy out, CUDA tensor in -> tensor out."""
    as_numpy = not torch.is_tensor(data)
    dev = torch.device("cuda", device) if as_numpy else data.device
    dd = torch.as_tensor(data, dtype=torch.float64, device=dev).contiguous()
    if dd.dim() != 4:
        raise ValueError("data must be [nx,ny,nz,nt]")
    radius = int(truncate * float(sigma) + 0.5)                      # scipy.ndimage.gaussian_filter1d
    x = np.arange(-radius, radius + 1)
    w = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    w = np.ascontiguousarray(w / w.sum(), dtype=np.float64)
    out = torch.empty_like(dd)
    work = torch.empty_like(dd)
    nx, ny, nz, nt = dd.shape
    import ctypes as C
    with torch.cuda.device(dev):
        check(lib().met2_smooth_separable(dev.index or 0, nx, ny, nz, nt, radius, w.ctypes.data_as(C.POINTER(C.c_double)), dd.data_ptr(),
                                          out.data_ptr(), work.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.current_stream(dev).synchronize()                 # `work` and the host weights stay alive until here
    return out.cpu().numpy() if as_numpy else out


def recon_met2_arrays(data, mask, TE_array, TR, reg_method="X2", reg_matrix="L2", FA_method="brute-force", myelin_T2=40.0,
                      fa_index=None, device=0, plan=None, denoise="None", prepared=False, FA_smooth="no"):
    """Steps 1-4 of motor_recon_met2 (motor:293-373, 427-472) on arrays: data [nx,ny,nz,nt] (or
    [nvox, nt]), mask [nx,ny,nz].  Mirrors the driver's preparation: data *= mask (motor:180-182),
    negative values clipped to 0 (motor:279), optional NESMA filter (motor:305-333, needs a 3-D volume),
    Npc = 60 (96 for T2SPARC, motor:207-213), T2 grid 10..2000 ms, T1 = 1000 ms, 91 flip angles for brute force.
    `prepared=True` says the caller already did that preparation (mask multiply, clip, denoise).
    FA_smooth='yes' (the CLI default, motor:337-343): the flip angles are estimated on the Gaussian-smoothed volume
    (sigma = 2 voxels, every echo), the spectra on the unsmoothed one; needs a 3-D volume.
    Returns a dict with the driver's ten outputs."""
    if FA_method not in ("brute-force", "spline"):
        raise ValueError("FA_method must be 'spline' or 'brute-force'")
    if denoise not in ("None", None, "none", "NESMA"):
        raise NotImplementedError("denoise=%r is not built: TV (motor:293-304) is scikit-image's estimate_sigma + "
                                  "denoise_tv_chambolle, a third-party dependency outside the path" % (denoise,))
    data = np.asarray(data, dtype=np.float64)
    vol_shape = data.shape[:-1]
    nt = data.shape[-1]
    mask = np.asarray(mask).reshape(vol_shape)
    dev = plan.device if plan is not None else torch.device("cuda", device)
    # the driver's preparation on the device (the volume goes up once; numpy would spend longer on these two passes
    # than the GPU on the whole fit)
    dd = torch.as_tensor(data, device=dev)
    mk = torch.as_tensor(mask, device=dev)
    if not prepared:
        dd = dd * mk.to(torch.float64).unsqueeze(-1)                # the mask VALUE multiplies (motor:180-182)
        dd = torch.where(dd < 0.0, torch.zeros((), dtype=torch.float64, device=dev), dd)      # motor:279
        if denoise == "NESMA":
            if len(vol_shape) != 3:
                raise ValueError("NESMA needs data [nx,ny,nz,nt]")
            dd = nesma_filter(dd, mk)
    dd_fa = None
    if FA_smooth == "yes" and fa_index is None:
        if len(vol_shape) != 3:
            raise ValueError("FA_smooth='yes' needs data [nx,ny,nz,nt]")
        dd_fa = gaussian_smooth(dd.reshape(vol_shape + (nt,)), 2.0).reshape(-1, nt)
    dd = dd.reshape(-1, nt).contiguous()
    if dd_fa is None:
        dd_fa = dd
    mm = (mk.reshape(-1) > 0)
    TE_array = np.asarray(TE_array, dtype=np.float64)
    tau = float(TE_array[1] - TE_array[0])
    Npc = 96 if reg_method == "T2SPARC" else 60
    T2s = np.logspace(math.log10(10.0), math.log10(2000.0), num=Npc, endpoint=True, base=10.0)
    T1s = 1000.0 * np.ones_like(T2s)
    spline = (FA_method == "spline") and fa_index is None
    alpha_values = np.linspace(90.0, 180.0, 91 * 3 if FA_method == "spline" else 91)      # motor:231-244
    alpha_values_spline = np.linspace(90.0, 180.0, 15)                                      # motor:237
    own = plan is None
    plan_lr = None
    if own:
        plan = Met2Plan(nt, Npc, alpha_values.shape[0], device=device, myelin_T2=myelin_T2)
        plan.build_dictionary_epg(T2s, T1s, tau, alpha_values, TR)
        plan.set_penalty("InvT2" if reg_method == "T2SPARC" else reg_matrix, T2s)   # run_real_data_script.py:91-93
    if spline:
        plan_lr = Met2Plan(nt, Npc, 15, device=device)
        plan_lr.build_dictionary_epg(T2s, T1s, tau, alpha_values_spline, TR)
        fa, km, _ = plan.fa_spline(plan_lr, alpha_values_spline, alpha_values, dd_fa, mm, want_km=False)
        plan_lr.close()
    elif fa_index is None:
        fa, km, _ = plan.fa_bruteforce(dd_fa, mm)
    else:
        fa = torch.as_tensor(np.asarray(fa_index, dtype=np.float64).reshape(-1), device=dev)
    out = plan.fit(reg_method, dd, fa_index=fa, mask=mm)
    res = {"fsol_4D": out["fsol"].cpu().numpy().reshape(vol_shape + (Npc,)),
           "Est_Signal": out["sig"].cpu().numpy().reshape(vol_shape + (nt,)),
           "reg_param": out["reg"].cpu().numpy().reshape(vol_shape),
           "FA_index": fa.cpu().numpy().reshape(vol_shape)}
    fitted_fa = (mm & (dd_fa.sum(dim=1) > 0)).cpu().numpy().reshape(vol_shape)      # gate of the FA step (fa_estimation.py:45)
    res["FA"] = np.where(fitted_fa, alpha_values[res["FA_index"].astype(int)], 0.0)
    maps = out["maps"].cpu().numpy()
    for i, name in enumerate(MAP_NAMES):
        res[name] = maps[i].reshape(vol_shape)
    res["T2s"] = T2s
    if own:
        plan.close()
    return res


def motor_recon_met2(TE_array, path_to_data, path_to_mask, path_to_save_data, TR, reg_method, reg_matrix, denoise, FA_method,
                     FA_smooth, myelin_T2, num_cores=-1, device=0):
    """Drop-in for motor_recon_met2 (motor:165-506) with the reference's on-disk contract: NIfTI in
    (data [nx,ny,nz,nt], mask [nx,ny,nz]), ten NIfTI volumes out (MWF, IEWF, FWF, T2_M, T2_IE, TWC, FA, fsol_4D,
    Est_Signal, reg_param .nii.gz at path_to_save_data, motor:475-503).  `num_cores` is accepted and ignored (one
    process drives the GPU).  denoise: 'None' or 'NESMA' (motor:305-333).  Not reproduced: TV denoising (motor:293-304,
    scikit-image) and the mean-spectrum PNG of motor:377-424."""
    from . import nifti
    if denoise not in ("None", None, "none", "NESMA"):
        raise NotImplementedError("denoise=%r is not built (TV, motor:293-304, is scikit-image code outside the path)" % (denoise,))
    img = nifti.load(path_to_data)
    data = img.get_fdata().astype(np.float64, copy=False)
    mask = nifti.load(path_to_mask).get_fdata().astype(np.int64)
    if data.ndim != 4 or mask.shape != data.shape[:3]:
        raise ValueError("data must be 4-D and mask must match its first three dimensions")
    res = recon_met2_arrays(data, mask, TE_array, TR, reg_method, reg_matrix, FA_method, myelin_T2, device=device, denoise=denoise,
                            FA_smooth=FA_smooth)
    for name in ("MWF", "IEWF", "FWF", "T2_M", "T2_IE", "TWC", "FA", "fsol_4D", "Est_Signal", "reg_param"):
        nifti.save(nifti.NiftiImage(res[name], img.affine), path_to_save_data + name + ".nii.gz")
    return res


def recon_met2_rois(data, rois, fa_index, Dic_3D, T2s, Laplac, factor=1.01, myelin_T2=40.0, device=0):
    """ROI-mode estimation (motor/motor_recon_met2_real_data_ROI.py:405-443): for every ROI label > 0 the mean signal and the
    mean EPG kernel over its voxels (each voxel contributes the dictionary slice of its own flip angle), one X2 fit
    (factor 1.01 there) per ROI, then the spectrum metrics.  data [..., nt], rois [...] integer labels, fa_index [...]
    (indices into Dic_3D's FA axis), Dic_3D [nt, nT2, nFA] (reference layout).
    Returns dict(labels, fsol [nROI, nT2] normalised to sum 1 like the reference, MWF, IEWF, FWF, T2_M, T2_IE, reg_opt, k_est)."""
    data = np.asarray(data, dtype=np.float64)
    nt = data.shape[-1]
    d2 = data.reshape(-1, nt)
    lab = np.asarray(rois).reshape(-1)
    fa = np.asarray(fa_index).reshape(-1).astype(np.int64)
    D3 = np.asarray(Dic_3D, dtype=np.float64)
    nT2, nFA = D3.shape[1], D3.shape[2]
    labels = np.array([v for v in np.unique(lab) if v > 0])
    if labels.size == 0:
        raise ValueError("no ROI label > 0")
    sig = np.zeros((labels.size, nt))
    W = np.zeros((labels.size, nFA))
    for i, v in enumerate(labels):
        sel = lab == v
        sig[i] = d2[sel].sum(axis=0) / sel.sum()
        W[i] = np.bincount(fa[sel], minlength=nFA) / sel.sum()
    kernels = np.einsum("rf,etf->etr", W, D3)                      # mean kernel per ROI, [nt, nT2, nROI]
    plan = Met2Plan(nt, nT2, labels.size, device=device, x2_factor=factor, myelin_T2=myelin_T2)
    try:
        plan.set_dictionary(np.ascontiguousarray(kernels)).set_t2_grid(T2s).set_penalty(np.asarray(Laplac, dtype=np.float64))
        out = plan.fit("X2", torch.as_tensor(sig, device=plan.device),
                       fa_index=torch.arange(labels.size, dtype=torch.float64, device=plan.device), want_lambda=True)
        f = out["fsol"].cpu().numpy()
        maps = out["maps"].cpu().numpy()
        res = {"labels": labels, "fsol": f / maps[5][:, None], "reg_opt": out["lam"].cpu().numpy(), "k_est": out["reg"].cpu().numpy()}
        for i, name in enumerate(MAP_NAMES[:5]):
            res[name] = maps[i]
        return res
    finally:
        plan.close()
