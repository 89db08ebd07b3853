"""Drop-ins for motor/motor_recon_met2_real_data.py: create_Laplacian_matrix, fitting_slice_T2, the NESMA filter,
recon_met2_arrays (the driver's steps 1-4 on in-memory arrays), motor_recon_met2 (the same with the on-disk
contract) and the ROI mode (recon_met2_rois, motor_recon_met2_ROIs).  Plots and the mean-spectrum PNG are not reproduced."""
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


def _prepare_volume(data, mask, dev, prepared, denoise):
    """The driver's preparation (motor:180-182, :279, :293-333) on the device.  The volume keeps the memory order it
    arrives in (nibabel arrays are Fortran-ordered; the solver reads either order in place)."""
    dd = torch.as_tensor(data, dtype=torch.float64, device=dev)
    mk = torch.as_tensor(mask, device=dev)
    if not prepared:
        dd = dd * mk.to(torch.float64).unsqueeze(-1)                # the mask VALUE multiplies (motor:180-182)
        dd = torch.where(dd < 0.0, torch.zeros((), dtype=torch.float64, device=dev), dd)      # motor:279
        if denoise == "NESMA":
            if dd.dim() != 4:
                raise ValueError("NESMA needs data [nx,ny,nz,nt]")
            dd = nesma_filter(dd, mk)
        elif denoise == "TV":
            if dd.dim() != 4:
                raise ValueError("TV denoising needs data [nx,ny,nz,nt]")
            from .tv import tv_denoise_volume
            dd = tv_denoise_volume(dd)
    return dd, mk


def _estimate_fa(plan, dd_fa, mm, FA_method, fa_index, T2s, T1s, tau, TR, alpha_values, device):
    """Driver step 2 (motor:349-373) -> float64 FA-index tensor shaped like the volume (dd_fa.shape[:-1]); the smoothed
    volume the angles are estimated on may be laid out differently from the one the spectra are fitted on, so the result
    is handed on by logical voxel position, not in a memory order."""
    from .plan import unflatten, voxel_layout
    _, nvox, _, _, vol, order = voxel_layout(dd_fa, plan.n_te)
    if fa_index is not None:
        return torch.as_tensor(np.asarray(fa_index, dtype=np.float64), device=dd_fa.device).reshape(vol)
    if FA_method == "spline":
        alpha_values_spline = np.linspace(90.0, 180.0, 15)                                      # motor:237
        plan_lr = Met2Plan(plan.n_te, plan.n_t2, 15, device=device)
        try:
            plan_lr.build_dictionary_epg(T2s, T1s, tau, alpha_values_spline, TR)
            fa, _, _ = plan.fa_spline(plan_lr, alpha_values_spline, alpha_values, dd_fa, mm, want_km=False)
        finally:
            plan_lr.close()
    else:
        fa, _, _ = plan.fa_bruteforce(dd_fa, mm)
    return unflatten(fa, vol, order)


def recon_met2_arrays(data, mask, TE_array, TR, reg_method="X2", reg_matrix="L2", FA_method="brute-force", myelin_T2=40.0,
                      fa_index=None, device=0, plan=None, denoise="None", prepared=False, FA_smooth="no", distributed=False,
                      return_prepared=False, devices=None):
    """Steps 1-4 of motor_recon_met2 (motor:293-373, 427-472) on arrays: data [nx,ny,nz,nt] (or
    [nvox, nt]), mask [nx,ny,nz].  Mirrors the driver's preparation: data *= mask (motor:180-182),
    negative values clipped to 0 (motor:279), optional NESMA / TV filter (motor:293-333, needs a 3-D volume),
    Npc = 60 (96 for T2SPARC, motor:207-213), T2 grid 10..2000 ms, T1 = 1000 ms, 91 flip angles for brute force.
    `prepared=True` says the caller already did that preparation (mask multiply, clip, denoise).
    FA_smooth='yes' (the CLI default, motor:337-343): the flip angles are estimated on the Gaussian-smoothed volume
    (sigma = 2 voxels, every echo), the spectra on the unsmoothed one; needs a 3-D volume.
    C- and Fortran-ordered volumes (nibabel's) are both read in place.
    distributed=True (under torch.distributed.run, one rank per GPU): every rank holds the volume, runs the FA step and the
    fit on its own interleaved 4096-voxel blocks of the voxel list and the outputs meet on rank 0 in ONE gather
    (dist.fit_sharded); ranks other than 0 return None.
    devices=[d0, d1, ...]: ONE process drives several GPUs through the C ABI's host entry (met2_fit_host: one plan and one host thread per
    device inside the call, blocks of voxels dealt round-robin, outputs copied by every device into the same host arrays -- no
    torch.distributed); the whole-volume filters (TV / NESMA / FA smoothing) run on devices[0] first.  Same outputs bit for bit.
    Returns a dict with the driver's ten outputs."""
    if FA_method not in ("brute-force", "spline"):
        raise ValueError("FA_method must be 'spline' or 'brute-force'")
    if denoise not in ("None", None, "none", "NESMA", "TV"):
        raise ValueError("denoise must be 'None', 'NESMA' or 'TV'")
    data = np.asarray(data, dtype=np.float64)
    vol_shape = data.shape[:-1]
    nt = data.shape[-1]
    mask = np.asarray(mask).reshape(vol_shape)
    if devices is None and plan is None and not distributed and data.ndim >= 2:
        devices = [device]                                       # the default: one device, through the C ABI's host entry (met2_fit_host: the block
                                                                 # pipeline inside the library; the torch pipeline of rounds 3-4 is retired to tests/tools)
    if devices is not None:
        if plan is not None or distributed:
            raise ValueError("devices=[...] builds its own plans and does not go with distributed=True")
        return _recon_multi_device(data, mask, TE_array, TR, reg_method, reg_matrix, FA_method, myelin_T2, fa_index, list(devices), prepared,
                                   denoise, FA_smooth, return_prepared)
    dev = plan.device if plan is not None else torch.device("cuda", device)
    # a caller's own plan, a distributed run, or a bare voxel list: the volume on the device in one piece
    dd, mk = _prepare_volume(data, mask, dev, prepared, denoise)
    dd_fa = dd
    if FA_smooth == "yes" and fa_index is None:
        if len(vol_shape) != 3:
            raise ValueError("FA_smooth='yes' needs data [nx,ny,nz,nt]")
        dd_fa = gaussian_smooth(dd, 2.0)
    mm = (mk > 0)
    TE_array = np.asarray(TE_array, dtype=np.float64)
    tau = float(TE_array[1] - TE_array[0])
    Npc = 96 if reg_method == "T2SPARC" else 60
    T2s = np.logspace(math.log10(10.0), math.log10(2000.0), num=Npc, endpoint=True, base=10.0)
    T1s = 1000.0 * np.ones_like(T2s)
    alpha_values = np.linspace(90.0, 180.0, 91 * 3 if FA_method == "spline" else 91)      # motor:231-244
    own = plan is None
    if own:
        plan = Met2Plan(nt, Npc, alpha_values.shape[0], device=device, myelin_T2=myelin_T2)
        plan.build_dictionary_epg(T2s, T1s, tau, alpha_values, TR)
        plan.set_penalty("InvT2" if reg_method == "T2SPARC" else reg_matrix, T2s)   # run_real_data_script.py:91-93
    try:
        if distributed:
            return _recon_sharded(plan, dd, dd_fa, mm, reg_method, FA_method, fa_index, T2s, T1s, tau, TR, alpha_values, device, vol_shape)
        fa_vol = _estimate_fa(plan, dd_fa, mm, FA_method, fa_index, T2s, T1s, tau, TR, alpha_values, device)
        out = plan.fit(reg_method, dd, fa_index=fa_vol, mask=mm)
        tot_fa = dd_fa.sum(dim=-1)
        res = {"fsol_4D": out["fsol"].cpu().numpy(), "Est_Signal": out["sig"].cpu().numpy(), "reg_param": out["reg"].cpu().numpy(),
               "FA_index": fa_vol.cpu().numpy()}
        fitted_fa = (mm & (tot_fa > 0)).cpu().numpy()      # gate of the FA step (fa_estimation.py:45)
        res["FA"] = np.where(fitted_fa, alpha_values[res["FA_index"].astype(int)], 0.0)
        maps = out["maps"].cpu().numpy()
        for i, name in enumerate(MAP_NAMES):
            res[name] = maps[i]
        res["T2s"] = T2s
        if return_prepared:
            res["data_prepared"] = dd.cpu().numpy()
        return res
    finally:
        if own:
            plan.close()


PIPELINE_CHUNK = 262144       # voxels per DMA block the driver asks met2_fit_host for (262 144: the library's own default; a test sets others)


def _match_layout(t, order):
    """the volume tensor t [..., nt] laid out in memory as `order` says ('C', or 'F': the reversed axes contiguous)"""
    rev = list(reversed(range(t.dim())))
    if order == "C":
        return t.contiguous()
    return t if t.permute(*rev).is_contiguous() else t.permute(*rev).contiguous().permute(*rev)


def _recon_multi_device(data, mask, TE_array, TR, reg_method, reg_matrix, FA_method, myelin_T2, fa_index, devices, prepared, denoise, FA_smooth,
                        return_prepared=False):
    """recon_met2_arrays(devices=[...]): one process, one plan per listed device, driven by met2_fit_host (host.fit_host).  The driver's
    preparation runs on the host for plain runs and, with a whole-volume filter (motor:293-343), on devices[0]; steps 2-4 then go through
    the host entry, which deals blocks of the voxel list to the devices."""
    from . import host as mhost
    if not devices:
        raise ValueError("devices is empty")
    vol_shape, nt = data.shape[:-1], data.shape[-1]
    filtered = denoise not in ("None", None, "none") or (FA_smooth == "yes" and fa_index is None)
    fa_vol = None
    mvals = None

    def to_pinned(t):                                            # device tensor -> numpy view of a pinned copy in the same memory order
        h = torch.empty_strided(tuple(t.shape), tuple(t.stride()), dtype=t.dtype, pin_memory=True)
        h.copy_(t)
        return h.numpy()

    keep_alive = None
    if filtered:
        dev0 = torch.device("cuda", devices[0])
        dd, _ = _prepare_volume(data, mask, dev0, prepared, denoise)
        one_device = len(set(devices)) == 1                      # the filtered volume stays where it is: met2_fit_host copies its blocks device to device
        place = (lambda t: t) if one_device else to_pinned
        if FA_smooth == "yes" and fa_index is None:
            if len(vol_shape) != 3:
                raise ValueError("FA_smooth='yes' needs data [nx,ny,nz,nt]")
            fa_vol = place(_match_layout(gaussian_smooth(dd, 2.0), "C" if dd.is_contiguous() else "F"))     # laid out like the volume itself
        if not (dd.is_contiguous() or dd.permute(*reversed(range(dd.dim()))).is_contiguous()):
            dd = dd.contiguous()
        vol = place(dd)                                          # keeps the memory order the volume came in
        keep_alive = dd
        torch.cuda.current_stream(dev0).synchronize()            # the library's streams do not wait for torch's
    else:
        vol = data                                               # prepared inside the library, per block, on the device (mask_values)
        if not prepared:
            mvals = np.asarray(mask, dtype=np.float64)
    if not torch.is_tensor(vol) and not (vol.flags.c_contiguous or vol.flags.f_contiguous):
        vol = np.ascontiguousarray(vol)
    order = "C" if (vol.is_contiguous() if torch.is_tensor(vol) else vol.flags.c_contiguous) else "F"
    TE_array = np.asarray(TE_array, dtype=np.float64)
    tau = float(TE_array[1] - TE_array[0])
    Npc = 96 if reg_method == "T2SPARC" else 60
    T2s = np.logspace(math.log10(10.0), math.log10(2000.0), num=Npc, endpoint=True, base=10.0)
    T1s = 1000.0 * np.ones_like(T2s)
    alpha_values = np.linspace(90.0, 180.0, 91 * 3 if FA_method == "spline" else 91)      # motor:231-244
    plans, coarse = [], []
    try:
        for d in devices:
            p = Met2Plan(nt, Npc, alpha_values.shape[0], device=d, myelin_T2=myelin_T2)
            plans.append(p)
            p.build_dictionary_epg(T2s, T1s, tau, alpha_values, TR)
            p.set_penalty("InvT2" if reg_method == "T2SPARC" else reg_matrix, T2s)      # run_real_data_script.py:91-93
        mode = False
        if fa_index is None:
            mode = "brute-force"
            if FA_method == "spline":
                alpha_lr = np.linspace(90.0, 180.0, 15)                                    # motor:237
                for d in devices:
                    q = Met2Plan(nt, Npc, 15, device=d)
                    coarse.append(q)
                    q.build_dictionary_epg(T2s, T1s, tau, alpha_lr, TR)
                mhost.attach_fa_spline(plans, coarse, alpha_lr, alpha_values)
                mode = "spline"
        nvox = int(np.prod(vol_shape))
        pin = lambda shape, dt=torch.float64: torch.empty(shape, dtype=dt, pin_memory=True).numpy()     # torch's caching host allocator: reused across calls
        bufs = {"fsol": pin((nvox, Npc)), "sig": pin((nvox, nt)), "reg": pin((nvox,)), "maps": pin((6, nvox)), "status": pin((nvox,), torch.int32),
                "fa_index": pin((nvox,)), "fa_gate": pin((nvox,))}
        out = mhost.fit_host(plans, reg_method, vol, fa_index=fa_index, mask=mask > 0, estimate_fa=mode, fa_data=fa_vol, mask_values=mvals, want_gate=True,
                             chunk=PIPELINE_CHUNK if PIPELINE_CHUNK != 262144 else 0, out=bufs)     # (0: the library's block size; a test that sets PIPELINE_CHUNK gets its chunks)
    finally:
        for p in plans + coarse:
            p.close()
    rv = tuple(reversed(vol_shape))
    nd = len(vol_shape)

    def unfold(a, lead=0):                                        # flat in the volume's memory order -> the volume's logical shape
        if order == "C":
            return a.reshape(a.shape[:lead] + vol_shape + a.shape[lead + 1:])
        u = a.reshape(a.shape[:lead] + rv + a.shape[lead + 1:])
        return u.transpose(list(range(lead)) + [lead + nd - 1 - i for i in range(nd)] + list(range(lead + nd, u.ndim)))

    res = {"fsol_4D": unfold(out["fsol"]), "Est_Signal": unfold(out["sig"]), "reg_param": unfold(out["reg"]), "FA_index": unfold(out["fa_index"])}
    fitted_fa = unfold(out["fa_gate"]) > 0                        # gate of the FA step (fa_estimation.py:45), formed on the device per block
    res["FA"] = np.where(fitted_fa, alpha_values[res["FA_index"].astype(int)], 0.0)
    maps = unfold(out["maps"], lead=1)
    for i, name in enumerate(MAP_NAMES):
        res[name] = maps[i]
    res["T2s"] = T2s
    if return_prepared:
        if mvals is not None:                                     # (plain runs prepare per block inside the library)
            vol = np.maximum(vol * mvals[..., None], 0.0)
        res["data_prepared"] = vol.cpu().numpy() if torch.is_tensor(vol) else vol
    return res


def _recon_sharded(plan, dd, dd_fa, mm, reg_method, FA_method, fa_index, T2s, T1s, tau, TR, alpha_values, device, vol_shape):
    """The multi-GPU leg of recon_met2_arrays: this rank's interleaved blocks through FA estimation + fit, one gather."""
    from . import dist as mdist
    from .plan import unflatten_back
    nt, Npc = plan.n_te, plan.n_t2
    flat = unflatten_back(dd, "C")                          # [nvox, nt] views in C voxel order (rows are gathered per shard)
    flat_fa = unflatten_back(dd_fa, "C")
    mflat = mm.reshape(-1)
    faflat = None if fa_index is None else torch.as_tensor(np.asarray(fa_index, dtype=np.float64).reshape(-1), device=dd.device)

    def fit_fn(idx):
        d = flat[idx].contiguous()
        dfa = d if dd_fa is dd else flat_fa[idx].contiguous()
        m = mflat[idx]
        fa = _estimate_fa(plan, dfa, m, FA_method, None if faflat is None else faflat[idx].cpu().numpy(), T2s, T1s, tau, TR, alpha_values, device)
        out = plan.fit(reg_method, d, fa_index=fa, mask=m)
        out["fa"] = fa
        out["fa_gate"] = (m & (dfa.sum(dim=1) > 0)).to(torch.float64)
        return out

    nvox = flat.shape[0]
    _, full = mdist.fit_sharded(fit_fn, nvox, gather=("fsol", "sig", "reg", "maps", "fa", "fa_gate"))
    if full is None:
        return None
    res = {"fsol_4D": full["fsol"].cpu().numpy().reshape(vol_shape + (Npc,)), "Est_Signal": full["sig"].cpu().numpy().reshape(vol_shape + (nt,)),
           "reg_param": full["reg"].cpu().numpy().reshape(vol_shape), "FA_index": full["fa"].cpu().numpy().reshape(vol_shape)}
    res["FA"] = np.where(full["fa_gate"].cpu().numpy().reshape(vol_shape) > 0, alpha_values[res["FA_index"].astype(int)], 0.0)
    maps = full["maps"].cpu().numpy()
    for i, name in enumerate(MAP_NAMES):
        res[name] = maps[i].reshape(vol_shape)
    res["T2s"] = T2s
    return res


def motor_recon_met2(TE_array, path_to_data, path_to_mask, path_to_save_data, TR, reg_method, reg_matrix, denoise, FA_method,
                     FA_smooth, myelin_T2, num_cores=-1, device=0, devices=None):
    """Drop-in for motor_recon_met2 (motor:165-506) with the reference's on-disk contract: NIfTI in
    (data [nx,ny,nz,nt], mask [nx,ny,nz]), ten NIfTI volumes out (MWF, IEWF, FWF, T2_M, T2_IE, TWC, FA, fsol_4D,
    Est_Signal, reg_param .nii.gz at path_to_save_data, motor:475-503).  `num_cores` is accepted and ignored (one
    process drives the GPU; devices=[0, 1, ...]: that one process drives all the listed GPUs through met2_fit_host).  denoise: 'None',
    'NESMA' (motor:305-333) or 'TV' (motor:293-304).  Not reproduced: the mean-spectrum PNG of motor:377-424."""
    from . import nifti
    img = nifti.load(path_to_data)
    data = img.get_fdata().astype(np.float64, copy=False)           # Fortran-ordered, like nibabel's: read in place by the solver
    mask = nifti.load(path_to_mask).get_fdata().astype(np.int64)
    if data.ndim != 4 or mask.shape != data.shape[:3]:
        raise ValueError("data must be 4-D and mask must match its first three dimensions")
    res = recon_met2_arrays(data, mask, TE_array, TR, reg_method, reg_matrix, FA_method, myelin_T2, device=device, denoise=denoise,
                            FA_smooth=FA_smooth, return_prepared=(denoise == "TV"), devices=devices)
    if denoise == "TV":                                             # motor:302-303
        nifti.save(nifti.NiftiImage(res.pop("data_prepared"), img.affine), path_to_save_data + "Data_denoised.nii.gz")
    for name in ("MWF", "IEWF", "FWF", "T2_M", "T2_IE", "TWC", "FA", "fsol_4D", "Est_Signal", "reg_param"):
        nifti.save(nifti.NiftiImage(res[name], img.affine), path_to_save_data + name + ".nii.gz")
    return res


def recon_met2_rois(data, rois, fa_index, Dic_3D, T2s, Laplac, factor=1.01, myelin_T2=40.0, device=0, plan=None):
    """ROI-mode estimation (motor/motor_recon_met2_real_data_ROI.py:405-443): for every ROI label > 0 the mean signal and the
    mean EPG kernel over its voxels (each voxel contributes the dictionary slice of its own flip angle), one X2 fit
    (factor 1.01 there) per ROI, then the spectrum metrics.  data [..., nt], rois [...] integer labels, fa_index [...]
    (indices into the FA axis of Dic_3D [nt, nT2, nFA], reference layout; or pass `plan`, a Met2Plan that already holds the
    dictionary).  The reduction runs on the device (met2_roi_reduce, deterministic).  numpy arrays or CUDA tensors.
    Returns dict(labels, count, fsol [nROI, nT2] normalised to sum 1 like the reference, MWF, IEWF, FWF, T2_M, T2_IE, TWC,
    reg_opt, k_est, mean_signal)."""
    from .plan import voxel_layout, _ptr
    src = plan if plan is not None else plan_for(Dic_3D, device=device)
    dev = src.device
    dd = torch.as_tensor(data, dtype=torch.float64, device=dev)
    nt = dd.shape[-1]
    dd, nvox, vs, es, vol, order = voxel_layout(dd, nt)
    lab = src._per_voxel(torch.as_tensor(rois, device=dev).to(torch.int64), nvox, torch.int64, "rois", order)
    fa = src._per_voxel(torch.as_tensor(fa_index, device=dev), nvox, torch.float64, "fa_index", order)
    labels = torch.unique(lab)
    labels = labels[labels > 0]
    nroi = int(labels.numel())
    if nroi == 0:
        raise ValueError("no ROI label > 0")
    pos = torch.searchsorted(labels, lab).clamp(max=nroi - 1)
    ridx = torch.where(labels[pos] == lab, pos, torch.full_like(pos, -1)).to(torch.int32).contiguous()
    dst = Met2Plan(nt, src.n_t2, nroi, device=dev.index or 0, x2_factor=factor, myelin_T2=myelin_T2)
    try:
        dst.set_t2_grid(T2s).set_penalty(np.asarray(Laplac, dtype=np.float64))
        sig = torch.empty((nroi, nt), dtype=torch.float64, device=dev)
        cnt = torch.empty((nroi,), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            check(lib().met2_roi_reduce(src._h, dst._h, nvox, _ptr(dd), vs, es, _ptr(ridx), _ptr(fa), _ptr(sig), _ptr(cnt), dst._stream()))
        out = dst.fit("X2", sig, fa_index=torch.arange(nroi, dtype=torch.float64, device=dev), want_lambda=True)
        f = out["fsol"].cpu().numpy()
        maps = out["maps"].cpu().numpy()
        res = {"labels": labels.cpu().numpy(), "count": cnt.cpu().numpy(), "fsol": f / maps[5][:, None], "reg_opt": out["lam"].cpu().numpy(),
               "k_est": out["reg"].cpu().numpy(), "mean_signal": sig.cpu().numpy()}
        for i, name in enumerate(MAP_NAMES):
            res[name] = maps[i]
        return res
    finally:
        dst.close()


def motor_recon_met2_ROIs(TE_array, path_to_data, path_to_mask, path_to_ROIs, path_to_save_data, TR, reg_matrix, denoise, FA_method,
                          FA_smooth, myelin_T2, num_cores=-1, device=0):
    """Drop-in for motor_recon_met2_ROIs (motor/motor_recon_met2_real_data_ROI.py:152-498): NIfTI data, mask and ROI labels in;
    flip angles per voxel (step 2), then one X2 fit (factor 1.01, :417) per ROI on the ROI's mean signal and mean kernel.
    Writes the reference's tables: table_MWF.csv, table_Spectra.csv, ROI_labels.csv at path_to_save_data and
    ROI_<label>/table_values.csv per ROI (:476-498; the PNG plots and the tabulate text table are not reproduced).
    Labels are taken from the ROI volume before the mask is applied, as the reference does (:175-178); a label that lies
    entirely outside the mask has no voxels and the reference's nnls_x2 raises ValueError on its nan kernel -- so does this."""
    import os
    from . import nifti
    img = nifti.load(path_to_data)
    data = img.get_fdata().astype(np.float64, copy=False)
    mask = nifti.load(path_to_mask).get_fdata().astype(np.int64)
    rois = nifti.load(path_to_ROIs).get_fdata().astype(np.int64)
    if data.ndim != 4 or mask.shape != data.shape[:3] or rois.shape != mask.shape:
        raise ValueError("data must be 4-D; mask and ROIs must match its first three dimensions")
    if FA_method not in ("brute-force", "spline"):
        raise ValueError("FA_method must be 'spline' or 'brute-force'")
    nt = data.shape[-1]
    labels_all = np.unique(rois)
    labels_all = labels_all[labels_all != 0]
    rois = rois * mask                                              # :191
    dev = torch.device("cuda", device)
    dd, mk = _prepare_volume(data, mask, dev, False, denoise)
    dd_fa = gaussian_smooth(dd, 2.0) if FA_smooth == "yes" else dd
    TE_array = np.asarray(TE_array, dtype=np.float64)
    tau = float(TE_array[1] - TE_array[0])
    Npc = 60
    T2s = np.logspace(math.log10(10.0), math.log10(2000.0), num=Npc, endpoint=True, base=10.0)
    T1s = 1000.0 * np.ones_like(T2s)
    alpha_values = np.linspace(90.0, 180.0, 91 * 3 if FA_method == "spline" else 91)
    Laplac = penalty_matrix(reg_matrix, Npc, T2s)
    plan = Met2Plan(nt, Npc, alpha_values.shape[0], device=device, myelin_T2=myelin_T2)
    try:
        plan.build_dictionary_epg(T2s, T1s, tau, alpha_values, TR)
        fa_vol = _estimate_fa(plan, dd_fa, mk > 0, FA_method, None, T2s, T1s, tau, TR, alpha_values, device)
        present = np.intersect1d(labels_all, np.unique(rois))
        if present.size != labels_all.size:
            raise ValueError("array must not contain infs or NaNs")  # a label without voxels inside the mask: 0/0 kernel (see docstring)
        res = recon_met2_rois(dd, torch.as_tensor(rois, device=dev), fa_vol, None, T2s, Laplac, factor=1.01, myelin_T2=myelin_T2,
                              device=device, plan=plan)
    finally:
        plan.close()
    np.savetxt(path_to_save_data + "table_MWF.csv", res["MWF"], delimiter=",", fmt="%s")
    np.savetxt(path_to_save_data + "table_Spectra.csv", res["fsol"], delimiter=",", fmt="%s")
    np.savetxt(path_to_save_data + "ROI_labels.csv", res["labels"], delimiter=",", fmt="%s")
    for i, lab in enumerate(res["labels"]):
        d = path_to_save_data + "ROI_%.0f/" % float(lab)
        os.makedirs(d, exist_ok=True)
        table = [["1. MWF       ", res["MWF"][i]], ["2. IEWF      ", res["IEWF"][i]], ["3. FWF       ", res["FWF"][i]],
                 ["4. T2M       ", res["T2_M"][i]], ["5. T2IE      ", res["T2_IE"][i]], ["6. TWC       ", res["TWC"][i]]]
        np.savetxt(d + "table_values.csv", np.array(table, dtype=object), delimiter=",", fmt="%s")
    return res
