"""denoise='TV' of the driver (motor/motor_recon_met2_real_data.py:293-304): per echo volume
    sigma_est = mean(estimate_sigma(vol));  vol <- denoise_tv_chambolle(vol, weight = 2 sigma_est, eps = 2e-4, max_num_iter = 200)

PARITY UNPINNED.  Both functions live in scikit-image (with PyWavelets underneath), neither of which is in this image, so the
reference's TV branch can be neither run nor turned into fixtures here.  What follows restates the published algorithms:
  * estimate_sigma: Donoho & Johnstone's robust wavelet estimator -- median(|d|) / Phi^-1(0.75) over the non-zero
    coefficients d of the finest all-detail sub-band of a separable db2 transform (half-sample symmetric extension, dyadic
    down-sampling: d[o] = sum_j g[j] x_ext[2 o + 1 - j]);
  * denoise_tv_chambolle: Chambolle's projection algorithm (J. Math. Imaging Vis. 20, 2004) for min_u |u - f|^2 / 2 + w TV(u)
    in n dimensions with forward differences, step tau = 1 / (2 n), stopping when the energy changes by less than
    eps x its first value.
It is pre-processing outside the hot path: plain torch tensor ops on whatever device the volume lives on (the volume is
already in HBM when the driver reaches this step); tests check the algorithmic properties, not parity."""
import math

import torch

# Daubechies-2 decomposition high-pass filter (PyWavelets' pywt.Wavelet('db2').dec_hi)
_DB2_DEC_HI = (-0.48296291314469025, 0.836516303737469, -0.22414386804185735, -0.12940952255092145)
_PHI_INV_075 = 0.6744897501960817          # scipy.stats.norm.ppf(0.75)


def _dwt_detail_axis(x, axis):
    """One level of the db2 high-pass branch along `axis` (symmetric extension, output length (N + 3) // 2)."""
    n = x.shape[axis]
    nout = (n + len(_DB2_DEC_HI) - 1) // 2
    o = torch.arange(nout, device=x.device)
    out = None
    for j, g in enumerate(_DB2_DEC_HI):
        idx = 2 * o + 1 - j
        idx = torch.where(idx < 0, -idx - 1, idx)
        idx = torch.where(idx >= n, 2 * n - 1 - idx, idx).clamp(0, n - 1)
        term = g * x.index_select(axis, idx)
        out = term if out is None else out + term
    return out


def estimate_sigma(vol):
    """Robust noise standard deviation of an n-d array (skimage.restoration.estimate_sigma(vol, channel_axis=None))."""
    d = vol
    for ax in range(vol.dim()):
        d = _dwt_detail_axis(d, ax)
    d = d.reshape(-1)
    d = d[d != 0].abs()
    if d.numel() == 0:
        return 0.0
    s, _ = torch.sort(d)
    m = s.numel()
    med = s[m // 2] if m % 2 else 0.5 * (s[m // 2 - 1] + s[m // 2])
    return float(med) / _PHI_INV_075


def denoise_tv_chambolle(image, weight=0.1, eps=2.0e-4, max_num_iter=200):
    """Chambolle's projection algorithm on an n-d tensor (skimage.restoration.denoise_tv_chambolle(image, weight, eps, max_num_iter,
    channel_axis=None)); returns the denoised tensor."""
    ndim = image.dim()
    if weight <= 0.0:
        return image.clone()
    p = torch.zeros((ndim,) + tuple(image.shape), dtype=image.dtype, device=image.device)
    g = torch.zeros_like(p)
    d = torch.zeros_like(image)
    tau = 1.0 / (2.0 * ndim)
    out = image
    e_init = e_prev = 0.0
    for i in range(max_num_iter):
        if i > 0:
            d = -p.sum(dim=0)                                          # minus the divergence of p (backward differences)
            for ax in range(ndim):
                n = image.shape[ax]
                d.narrow(ax, 1, n - 1).add_(p[ax].narrow(ax, 0, n - 1))
            out = image + d
        energy = float((d * d).sum())
        for ax in range(ndim):                                         # forward differences of `out`
            n = image.shape[ax]
            g[ax].zero_()
            g[ax].narrow(ax, 0, n - 1).copy_(out.narrow(ax, 1, n - 1) - out.narrow(ax, 0, n - 1))
        norm = torch.sqrt((g * g).sum(dim=0))
        energy += weight * float(norm.sum())
        norm = norm * (tau / weight) + 1.0
        p -= tau * g
        p /= norm.unsqueeze(0)
        energy /= float(image.numel())
        if i == 0:
            e_init = e_prev = energy
        else:
            if abs(e_prev - energy) < eps * e_init:
                break
            e_prev = energy
    return out


def tv_denoise_volume(data, weight_factor=2.0, eps=2.0e-4, max_num_iter=200):
    """motor:293-304 on data [nx, ny, nz, nt]: every echo volume through estimate_sigma + denoise_tv_chambolle."""
    if data.dim() != 4:
        raise ValueError("data must be [nx,ny,nz,nt]")
    out = torch.empty_like(data)
    for t in range(data.shape[3]):
        vol = data[..., t].contiguous()
        sigma = estimate_sigma(vol)
        if not math.isfinite(sigma):
            raise ValueError("array must not contain infs or NaNs")
        out[..., t] = denoise_tv_chambolle(vol, weight_factor * sigma, eps, max_num_iter)
    return out
