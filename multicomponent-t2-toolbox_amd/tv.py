"""denoise='TV' of the driver (motor/motor_recon_met2_real_data.py:293-304): per echo volume
    sigma_est = mean(estimate_sigma(vol));  vol <- denoise_tv_chambolle(vol, weight = 2 sigma_est, eps = 2e-4, max_num_iter = 200)
on the device through `met2_tv_chambolle` (csrc/met2_tv.hip): every echo volume in the same launches, one fused stencil kernel per
Chambolle iteration, the stopping rule applied on the device.  There is no CPU or torch fallback.

PARITY UNPINNED: both functions live in scikit-image (PyWavelets underneath), neither of which is in this image; the kernels restate the
published algorithms in scikit-image's operation order; tests/test_tv.py checks them against an independent numpy restatement."""
import ctypes as C

import numpy as np
import torch

from ._lib import Met2Error, check, lib


def _layout(data):
    """-> (tensor as it lies in memory, echo_major flag).  A Fortran-ordered [nx,ny,nz,nt] tensor (the order nibabel's arrays have) is
    read in place as [nt][nz][ny][nx]; anything else is made C-contiguous."""
    if data.dim() != 4:
        raise ValueError("data must be [nx,ny,nz,nt]")
    if not data.is_contiguous() and data.permute(3, 2, 1, 0).is_contiguous():
        return data, 1
    return data.contiguous(), 0


def tv_chambolle(data, weight=None, weight_factor=2.0, eps=2.0e-4, max_num_iter=200, poll_every=4, device=0, return_info=False):
    """Every echo volume of data [nx,ny,nz,nt] through denoise_tv_chambolle; `weight` = one weight per echo (array) or None:
    weight_factor x the echo's estimate_sigma.  numpy in -> numpy out, CUDA tensor in -> tensor out (same memory order).
    return_info: also (sigma [nt], iterations [nt]) as numpy arrays."""
    as_numpy = not torch.is_tensor(data)
    dev = torch.device("cuda", device) if as_numpy else data.device
    if dev.type != "cuda":
        raise Met2Error("TV denoising runs on the GPU only; there is no CPU fallback")
    dd, echo_major = _layout(torch.as_tensor(data, dtype=torch.float64, device=dev))
    nx, ny, nz, nt = dd.shape
    out = torch.empty_strided(dd.shape, dd.stride(), dtype=torch.float64, device=dev)
    if dd.numel() == 0:
        return (out.cpu().numpy() if as_numpy else out, np.zeros(nt), np.zeros(nt, dtype=np.int32)) if return_info else (out.cpu().numpy() if as_numpy else out)
    L = lib()
    nbytes = int(L.met2_tv_work_bytes(nx, ny, nz, nt, echo_major))
    work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    sig = torch.empty(nt, dtype=torch.float64, device=dev)
    its = torch.empty(nt, dtype=torch.int32, device=dev)
    wp = None
    if weight is not None:
        w = np.ascontiguousarray(np.broadcast_to(np.asarray(weight, dtype=np.float64), (nt,)))
        wp = w.ctypes.data_as(C.POINTER(C.c_double))
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        check(L.met2_tv_chambolle(dev.index or 0, nx, ny, nz, nt, dd.data_ptr(), echo_major, wp, float(weight_factor), float(eps),
                                  int(max_num_iter), int(poll_every), out.data_ptr(), sig.data_ptr(), its.data_ptr(), work.data_ptr(),
                                  nbytes, stream.cuda_stream))
        stream.synchronize()                                         # `work` and the host weights stay alive until here
    res = out.cpu().numpy() if as_numpy else out
    if return_info:
        return res, sig.cpu().numpy(), its.cpu().numpy()
    return res


def tv_denoise_volume(data, weight_factor=2.0, eps=2.0e-4, max_num_iter=200, return_info=False):
    """motor:293-304 on data [nx, ny, nz, nt]: estimate_sigma + denoise_tv_chambolle for every echo volume.  A nan or inf in the data
    raises like the reference's finite check."""
    res = tv_chambolle(data, None, weight_factor, eps, max_num_iter, return_info=True)
    if weight_factor != 0 and not np.isfinite(res[1]).all():
        raise ValueError("array must not contain infs or NaNs")
    return res if return_info else res[0]
