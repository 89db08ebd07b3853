"""CPU checker for denoise='TV' (motor/motor_recon_met2_real_data.py:293-304).  TEST INFRASTRUCTURE: only tests/ and bench.py's
cpu_baseline leg may import this module; the product (multicomponent-t2-toolbox_amd/tv.py -> csrc/met2_tv.hip) never does.

PARITY UNPINNED.  The reference calls scikit-image's `estimate_sigma` and `denoise_tv_chambolle` (motor:299-301); scikit-image
and PyWavelets are third-party code outside /root/reference and are not installed in this image (requirements.txt pins
scikit-image only as 'scikit-image', no fixture of theirs is in the reference tree), so neither function can be run here and no
golden vector exists.  What follows restates, in plain numpy and in the operation order scikit-image documents for them,
  * estimate_sigma(image, channel_axis=None): sigma = median(|d|) / Phi^-1(0.75) over the NON-ZERO coefficients d of the
    finest all-detail ('ddd') sub-band of pywt.dwtn(image, 'db2') -- separable, axis 0 first, mode 'symmetric' (half-sample
    symmetric extension), d[o] = sum_j dec_hi[j] * x_ext[2 o + 1 - j], output length (n + 3) // 2 per axis
    (Donoho & Johnstone, Biometrika 81, 1994);
  * denoise_tv_chambolle(image, weight, eps, max_num_iter, channel_axis=None) for a float64 n-d image
    (Chambolle, J. Math. Imaging Vis. 20, 2004): the loop of `_denoise_tv_chambolle_nd`.
"""
import numpy as np

DB2_DEC_HI = (-0.48296291314469025, 0.836516303737469, -0.22414386804185735, -0.12940952255092145)
PHI_INV_075 = 0.6744897501960817          # scipy.stats.norm.ppf(0.75)


def _reflect(idx, n):
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * n
    idx = np.mod(idx, period)
    return np.where(idx < n, idx, period - 1 - idx)


def dwt_detail_axis(x, axis):
    """One level of the db2 high-pass branch along `axis`: symmetric extension, dyadic down-sampling."""
    n = x.shape[axis]
    nout = (n + len(DB2_DEC_HI) - 1) // 2
    o = np.arange(nout)
    out = None
    for j, g in enumerate(DB2_DEC_HI):
        term = g * np.take(x, _reflect(2 * o + 1 - j, n), axis=axis)
        out = term if out is None else out + term
    return out


def detail_coefficients(vol):
    d = np.asarray(vol, dtype=np.float64)
    for ax in range(d.ndim):
        d = dwt_detail_axis(d, ax)
    return d


def estimate_sigma(vol):
    d = detail_coefficients(vol).ravel()
    d = d[np.nonzero(d)]
    if d.size == 0:
        return 0.0
    return float(np.median(np.abs(d)) / PHI_INV_075)


def denoise_tv_chambolle(image, weight=0.1, eps=2.0e-4, max_num_iter=200, return_iters=False):
    image = np.asarray(image, dtype=np.float64)
    ndim = image.ndim
    p = np.zeros((ndim,) + image.shape, dtype=image.dtype)
    g = np.zeros_like(p)
    d = np.zeros_like(image)
    out = image
    i = 0
    n_done = 0
    while i < max_num_iter:
        if i > 0:
            d = -p.sum(0)                                            # minus the divergence of p
            slices_d = [slice(None)] * ndim
            slices_p = [slice(None)] * (ndim + 1)
            for ax in range(ndim):
                slices_d[ax] = slice(1, None)
                slices_p[ax + 1] = slice(0, -1)
                slices_p[0] = ax
                d[tuple(slices_d)] += p[tuple(slices_p)]
                slices_d[ax] = slice(None)
                slices_p[ax + 1] = slice(None)
            out = image + d
        else:
            out = image
        E = (d ** 2).sum()
        slices_g = [slice(None)] * (ndim + 1)
        for ax in range(ndim):                                       # forward differences of `out`
            slices_g[ax + 1] = slice(0, -1)
            slices_g[0] = ax
            g[tuple(slices_g)] = np.diff(out, axis=ax)
            slices_g[ax + 1] = slice(None)
        norm = np.sqrt((g ** 2).sum(axis=0))[np.newaxis, ...]
        E += weight * norm.sum()
        tau = 1.0 / (2.0 * ndim)
        norm *= tau / weight
        norm += 1.0
        p -= tau * g
        p /= norm
        E /= float(image.size)
        n_done = i + 1
        if i == 0:
            E_init = E
            E_previous = E
        else:
            if np.abs(E_previous - E) < eps * E_init:
                break
            E_previous = E
        i += 1
    return (out, n_done) if return_iters else out


def tv_denoise_volume(data, weight_factor=2.0, eps=2.0e-4, max_num_iter=200, return_info=False):
    """motor:293-304 on data [nx, ny, nz, nt]."""
    data = np.asarray(data, dtype=np.float64)
    out = np.empty_like(data)
    sig = np.zeros(data.shape[3]); its = np.zeros(data.shape[3], dtype=np.int32)
    for t in range(data.shape[3]):
        vol = np.ascontiguousarray(data[..., t])
        sig[t] = estimate_sigma(vol)
        w = weight_factor * sig[t]
        if not (w > 0.0) or not np.isfinite(w):                      # documented deviation of the product: such an echo is copied through
            out[..., t] = vol
            continue
        out[..., t], its[t] = denoise_tv_chambolle(vol, w, eps, max_num_iter, return_iters=True)
    return (out, sig, its) if return_info else out
