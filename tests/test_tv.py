"""denoise='TV' (motor:293-304).  PARITY UNPINNED -- scikit-image / PyWavelets are not in this image, so there is nothing of the
reference's to compare with.  CPU: the numpy restatement (oracle/tv_oracle.py) against the published algorithms' properties and
against a second, independently written transcription of Chambolle's iteration.  GPU: the HIP kernels (csrc/met2_tv.hip, through
the C ABI) against that restatement -- same iteration count per echo, values to 1e-12."""
import importlib

import numpy as np
import pytest
import torch

from oracle import tv_oracle

PKG = "multicomponent-t2-toolbox_amd"
gpu = pytest.mark.gpu


def _np_chambolle(image, weight, eps=2e-4, iters=200):
    """Chambolle (2004), eqs. (9)-(10) in n-d with forward differences: p <- (p - tau grad(f - div p)) / (1 + tau/w |grad|)."""
    nd = image.ndim
    p = np.zeros((nd,) + image.shape)
    tau = 1.0 / (2 * nd)
    out = image
    e0 = ep = None
    for i in range(iters):
        if i > 0:
            div = np.zeros_like(image)
            for ax in range(nd):
                pa = p[ax]
                first = np.take(pa, [0], axis=ax)
                mid = np.diff(pa, axis=ax)
                div += np.concatenate([first, mid], axis=ax)
            d = -div
            out = image + d
        else:
            d = np.zeros_like(image)
        g = np.zeros_like(p)
        for ax in range(nd):
            sl = [slice(None)] * nd; sl[ax] = slice(0, -1)
            g[ax][tuple(sl)] = np.diff(out, axis=ax)
        nrm = np.sqrt((g ** 2).sum(axis=0))
        e = ((d ** 2).sum() + weight * nrm.sum()) / image.size
        p = (p - tau * g) / (1.0 + tau / weight * nrm)[None]
        if i == 0:
            e0 = ep = e
        elif abs(ep - e) < eps * e0:
            break
        else:
            ep = e
    return out


def _phantom(shape, nt, seed, noise=4.0):
    """Piecewise-smooth echo volumes: a decaying two-compartment box phantom plus Gaussian noise."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = shape
    x, y, z = np.meshgrid(np.linspace(-1, 1, nx), np.linspace(-1, 1, ny), np.linspace(-1, 1, nz), indexing="ij")
    inner = (np.abs(x) < 0.5) & (np.abs(y) < 0.6) & (np.abs(z) < 0.4)
    body = (x ** 2 + y ** 2 + z ** 2) < 0.9
    vol = np.zeros(shape + (nt,))
    for t in range(nt):
        vol[..., t] = body * (120.0 * np.exp(-t / 6.0) + 20.0 * x) + inner * 60.0 * np.exp(-t / 2.5)
    return np.abs(vol + noise * rng.standard_normal(vol.shape))


# ---- CPU: the restatement itself -------------------------------------------------------------------------------------------------
def test_oracle_chambolle_matches_second_transcription_and_reduces_tv():
    rng = np.random.default_rng(3)
    base = np.zeros((12, 10, 8)); base[3:9, 2:7, 2:6] = 100.0
    noisy = base + 5.0 * rng.standard_normal(base.shape)
    got = tv_oracle.denoise_tv_chambolle(noisy, weight=10.0)
    ref = _np_chambolle(noisy, 10.0)
    assert np.max(np.abs(got - ref)) < 1e-10
    tvn = lambda a: sum(np.abs(np.diff(a, axis=ax)).sum() for ax in range(a.ndim))
    assert tvn(got) < 0.5 * tvn(noisy)
    assert np.abs(got - base).mean() < np.abs(noisy - base).mean()          # closer to the clean image
    assert abs(got.mean() - noisy.mean()) < 1e-9                               # the projection preserves the mean
    const = np.full((6, 5, 4), 7.0)
    assert np.array_equal(tv_oracle.denoise_tv_chambolle(const, weight=3.0), const)


def test_oracle_estimate_sigma_recovers_the_noise_level():
    rng = np.random.default_rng(4)
    x, y, z = np.meshgrid(np.linspace(0, 1, 48), np.linspace(0, 1, 40), np.linspace(0, 1, 32), indexing="ij")
    smooth = 200.0 * np.exp(-((x - 0.5) ** 2 + (y - 0.4) ** 2 + (z - 0.5) ** 2) / 0.1)
    for sigma in (1.0, 7.5):
        est = tv_oracle.estimate_sigma(smooth + sigma * rng.standard_normal(smooth.shape))
        # the border coefficients of the symmetric extension see duplicated samples (variance 0.75 sigma^2 per border axis), which
        # pulls the median down by a few per cent on a volume this small -- PyWavelets' construction has the same property
        assert -0.08 < est / sigma - 1.0 < 0.02, (sigma, est)
    assert tv_oracle.estimate_sigma(np.zeros((8, 8, 8))) == 0.0
    # the db2 high-pass filter annihilates constants and linear ramps (two vanishing moments) away from the borders
    d = tv_oracle.dwt_detail_axis(np.arange(64, dtype=np.float64), 0)
    assert d.shape[0] == 33 and float(np.abs(d[2:-2]).max()) < 1e-12
    # the filter is the quadrature mirror of db2's low-pass: unit energy, zero mean (to the 13 digits PyWavelets' table carries)
    g = np.array(tv_oracle.DB2_DEC_HI)
    assert abs((g ** 2).sum() - 1.0) < 1e-11 and abs(g.sum()) < 1e-11


def test_product_tv_has_no_cpu_fallback():
    tv = importlib.import_module(PKG + "._lib")
    mod = importlib.import_module(PKG + ".tv")
    with pytest.raises(tv.Met2Error):
        mod.tv_chambolle(torch.zeros((4, 4, 4, 2), dtype=torch.float64))          # a CPU tensor: refused, not computed elsewhere
    assert "oracle" not in open(mod.__file__).read()                              # the module never touches the checker


# ---- GPU: the kernels against the restatement ---------------------------------------------------------------------------------
def _check(vol, got, sig, its, weight=None, tol=1e-12):
    want, wsig, wits = [], [], []
    for t in range(vol.shape[3]):
        v = np.ascontiguousarray(vol[..., t])
        s = tv_oracle.estimate_sigma(v)
        w = 2.0 * s if weight is None else float(np.broadcast_to(weight, (vol.shape[3],))[t])
        o, n = tv_oracle.denoise_tv_chambolle(v, w, return_iters=True)
        want.append(o); wsig.append(s); wits.append(n)
    want = np.stack(want, axis=3)
    assert np.array_equal(its, np.array(wits)), (its, wits)                        # the same iteration count for every echo
    if weight is None:
        assert np.max(np.abs(sig / np.array(wsig) - 1.0)) < 1e-14, (sig, wsig)
    scale = np.max(np.abs(want))
    assert np.max(np.abs(got - want)) <= tol * scale, np.max(np.abs(got - want)) / scale


@gpu
@pytest.mark.parametrize("shape,nt", [((20, 18, 24), 3), ((9, 37, 70), 2), ((35, 16, 64), 4), ((17, 33, 130), 1), ((3, 2, 5), 2), ((14, 11, 1), 2), ((1, 12, 9), 1),
                                      ((16, 1, 20), 2)])
def test_hip_tv_matches_the_numpy_restatement(shape, nt):
    """C-ordered volumes: one tile, tiles with halo rows along axis 1 (33, 37 > 15), halo lanes along axis 2 (70, 130 > 63),
    several segments along axis 0 (35 > 16), a volume smaller than any tile, volumes with a singleton axis (a 2-D slice kept as a 3-D array:
    scikit-image still runs its three-axis iteration with tau = 1/6 on it)."""
    tv = importlib.import_module(PKG + ".tv")
    vol = _phantom(shape, nt, seed=sum(shape) + nt)
    got, sig, its = tv.tv_denoise_volume(vol, return_info=True)
    assert got.shape == vol.shape and (its > 1).all()
    _check(vol, got, sig, its)


@gpu
def test_hip_tv_fortran_ordered_volume_in_place_layout():
    """The array nibabel hands the driver is Fortran-ordered: read as [nt][nz][ny][nx] without a copy; numpy's axis order of
    every ordered sum is kept, so the result is the C-ordered one to rounding (and the oracle's to 1e-12)."""
    tv = importlib.import_module(PKG + ".tv")
    vol = _phantom((70, 21, 19), 3, seed=11)
    dev = torch.device("cuda", 0)
    tf = torch.as_tensor(np.asfortranarray(vol), device=dev)                      # strides (1, nx, nx ny, nx ny nz)
    assert not tf.is_contiguous()
    got, sig, its = tv.tv_chambolle(tf, return_info=True)
    assert got.stride() == tf.stride()
    _check(vol, got.cpu().numpy(), sig, its)
    gc, sc, ic = tv.tv_chambolle(torch.as_tensor(vol, device=dev), return_info=True)
    assert np.array_equal(ic, its) and np.array_equal(sc, sig)
    assert torch.equal(gc, got)                                                   # same operations in the same order: same bits


@gpu
def test_hip_tv_weights_polling_copy_through_and_errors():
    tv = importlib.import_module(PKG + ".tv")
    vol = _phantom((24, 20, 16), 4, seed=5)
    w = np.array([3.0, 11.0, 0.0, 6.5])                                            # echo 2: weight 0 -> copied through
    got, sig, its = tv.tv_chambolle(vol, weight=w, return_info=True)
    assert its[2] == 0 and np.array_equal(got[..., 2], vol[..., 2])
    keep = [0, 1, 3]
    _check(vol[..., keep], got[..., keep], sig[keep], its[keep], weight=w[keep])
    assert np.max(np.abs(sig / np.array([tv_oracle.estimate_sigma(np.ascontiguousarray(vol[..., t])) for t in range(4)]) - 1.0)) < 1e-14
    # the host's polling interval changes when it stops enqueueing, never the result
    a = tv.tv_chambolle(vol, weight=w, poll_every=0)
    b = tv.tv_chambolle(vol, weight=w, poll_every=1)
    assert np.array_equal(a, got) and np.array_equal(b, got)
    # max_num_iter is honoured per echo
    c, _, itc = tv.tv_chambolle(vol, weight=w, max_num_iter=5, return_info=True)
    assert np.array_equal(itc, np.minimum(its, 5)) and (itc[keep] == 5).all()
    o5 = tv_oracle.denoise_tv_chambolle(np.ascontiguousarray(vol[..., 1]), 11.0, max_num_iter=5)
    assert np.max(np.abs(c[..., 1] - o5)) < 1e-12 * np.max(np.abs(o5))
    # an all-zero volume has sigma = 0: copied through (scikit-image would return nan); a constant one comes back unchanged
    zero = np.zeros((8, 8, 8, 2))
    out, sg, it = tv.tv_denoise_volume(zero, return_info=True)
    assert np.array_equal(out, zero) and (sg == 0).all() and (it == 0).all()
    const = np.full((8, 8, 8, 2), 5.0)
    assert np.array_equal(tv.tv_denoise_volume(const), const)
    bad = vol.copy(); bad[3, 4, 5, 1] = np.nan
    with pytest.raises(ValueError):
        tv.tv_denoise_volume(bad)
    # result written over the input
    dev = torch.device("cuda", 0)
    td = torch.as_tensor(vol, device=dev)
    L = importlib.import_module(PKG + "._lib")
    nb = int(L.lib().met2_tv_work_bytes(24, 20, 16, 4, 0))
    work = torch.empty(nb, dtype=torch.uint8, device=dev)
    L.check(L.lib().met2_tv_chambolle(0, 24, 20, 16, 4, td.data_ptr(), 0, None, 2.0, 2e-4, 200, 0, td.data_ptr(), None, None, work.data_ptr(), nb,
                                      torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
    assert np.array_equal(td.cpu().numpy(), tv.tv_denoise_volume(vol))
    with pytest.raises(L.Met2Error):
        L.check(L.lib().met2_tv_chambolle(0, 24, 20, 16, 4, td.data_ptr(), 0, None, 2.0, 2e-4, 200, 0, td.data_ptr(), None, None, work.data_ptr(), nb - 1, None))


@gpu
def test_tv_driver_step_runs_on_the_device():
    """recon_met2_arrays(denoise='TV') = mask multiply, clip, TV, fit: the prepared volume it returns is the kernel's result."""
    motor = importlib.import_module(PKG + ".motor")
    tv = importlib.import_module(PKG + ".tv")
    rng = np.random.default_rng(5)
    vol = _phantom((10, 9, 8), 32, seed=9, noise=2.0)
    mask = (rng.uniform(size=(10, 9, 8)) > 0.1).astype(np.int64)
    te = 10.0 * np.arange(1, 33)
    res = motor.recon_met2_arrays(vol, mask, te, 3000.0, "X2", "L2", "spline", 40.0, denoise="TV", return_prepared=True)
    prep = vol * mask[..., None]
    want = tv.tv_denoise_volume(prep)
    assert np.max(np.abs(np.asarray(res["data_prepared"]) - want)) == 0.0
    for t in range(0, 32, 7):
        assert np.std(want[..., t]) < np.std(prep[..., t])
    assert np.isfinite(res["MWF"]).all()
