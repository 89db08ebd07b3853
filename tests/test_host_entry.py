"""met2_fit_host (ABI 5, csrc/met2_host.hip): the host-to-host, one-or-several-devices entry of the C ABI -- numpy arrays in and out,
one host thread per plan inside the call.  One GPU is what the test box has: several plans share device 0, which exercises everything
but the choice of device (block dealing, one thread per plan, three streams each, staging, the error path)."""
import ctypes as C
import importlib

import numpy as np
import pytest

PKG = "multicomponent-t2-toolbox_amd"
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module(PKG)


def _plans(pkg, count, nte=32, nt2=60, nfa=5, penalty="L2"):
    synth = importlib.import_module(PKG + ".synth")
    T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2)
    alphas = np.linspace(120.0, 180.0, nfa) if nfa > 1 else np.array([150.0])
    plans = []
    for _ in range(count):
        p = pkg.Met2Plan(nte, nt2, nfa)
        p.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty(penalty, T2s)
        plans.append(p)
    return plans, alphas, T2s, T1s


def _reference_fit(plan, method, data_np, fa=None, mask=None):
    """one met2_fit over the whole list through device pointers"""
    import torch
    d = torch.as_tensor(np.ascontiguousarray(data_np), device="cuda")
    out = plan.fit(method, d, fa_index=None if fa is None else torch.as_tensor(fa, device="cuda"),
                   mask=None if mask is None else torch.as_tensor(mask, device="cuda"), want_lambda=True)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def _same(res, ref, keys=("fsol", "sig", "reg", "lam", "maps", "status")):
    for k in keys:
        a, b = res[k], ref[k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert np.array_equal(a, b, equal_nan=True), "%s differs: max |d| %.3e" % (k, np.nanmax(np.abs(a.astype(np.float64) - b)))


@gpu
@pytest.mark.parametrize("method", ["X2", "L_curve", "T2SPARC"])
def test_one_plan_pageable_arrays_equal_the_device_entry(pkg, method):
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    (plan,), alphas, _, _ = _plans(pkg, 1)
    nvox = 10_000 + 37
    data, fa, _ = synth.make_voxels(nvox, nte=32, seed=5, fa_values=alphas, device="cpu")
    data, fa = data.numpy(), fa.numpy()
    mask = (np.arange(nvox) % 7 != 3).astype(np.uint8)
    ref = _reference_fit(plan, method, data, fa, mask)
    res = host.fit_host(plan, method, data, fa_index=fa, mask=mask, chunk=3000, want_lambda=True)     # four blocks, a ragged last one
    _same(res, ref)
    assert np.array_equal(res["fa_index"], fa)
    assert res["plan_ms"].shape == (1,) and res["plan_ms"][0] > 0
    # a second call reuses the plan's block buffers and the caller's arrays
    again = host.fit_host(plan, method, data, fa_index=fa, mask=mask, chunk=3000, want_lambda=True, out=res)
    assert again["fsol"] is res["fsol"]
    _same(again, ref)
    # default block size, no mask, no flip angles (flip angle 0), optional outputs off
    ref0 = _reference_fit(plan, method, data)
    r0 = host.fit_host(plan, method, data, want_sig=False, want_maps=False, want_status=False)
    assert r0["sig"] is None and r0["maps"] is None and r0["status"] is None and r0["lam"] is None
    _same(r0, ref0, keys=("fsol", "reg"))
    assert not r0["fa_index"].any()


@gpu
@pytest.mark.parametrize("count,chunk", [(2, 1024), (3, 4096), (5, 700)])
def test_several_plans_share_the_device_bit_equal(pkg, count, chunk):
    """block b -> plan b mod n_plans, one host thread per plan: same bits as one fit, and every plan worked"""
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    plans, alphas, _, _ = _plans(pkg, count)
    nvox = 20_011
    data, fa, _ = synth.make_voxels(nvox, nte=32, seed=11, fa_values=alphas, device="cpu")
    data, fa = data.numpy(), fa.numpy()
    ref = _reference_fit(plans[0], "X2", data, fa)
    res = host.fit_host(plans, "X2", data, fa_index=fa, chunk=chunk, want_lambda=True)
    _same(res, ref)
    assert (res["plan_ms"] > 0).all()
    # fewer blocks than plans: the idle plans return at once
    few = host.fit_host(plans, "X2", data[:600], fa_index=fa[:600], chunk=600, want_lambda=True)
    _same(few, {k: (v[:, :600] if k == "maps" else v[:600]) for k, v in ref.items()})


@gpu
def test_layouts_fortran_volume_strided_rows_and_general_strides(pkg):
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    plans, alphas, _, _ = _plans(pkg, 2)
    nx, ny, nz, nte = 12, 11, 9, 32
    nvox = nx * ny * nz
    data, fa, _ = synth.make_voxels(nvox, nte=nte, seed=3, fa_values=alphas, device="cpu")
    data, fa = data.numpy(), fa.numpy()
    ref = _reference_fit(plans[0], "X2", data, fa)
    # the Fortran-ordered volume nibabel delivers: voxel v = x + nx (y + ny z), echo-major in memory
    vol_f = np.asfortranarray(data.reshape(nz, ny, nx, nte).transpose(2, 1, 0, 3))
    assert vol_f.flags.f_contiguous and not vol_f.flags.c_contiguous
    r = host.fit_host(plans, "X2", vol_f, fa_index=fa, chunk=500, want_lambda=True)
    _same(r, ref)
    # rows cut out of a wider table (echo_stride 1, voxel_stride > n_te)
    wide = np.zeros((nvox, nte + 5)); wide[:, :nte] = data
    r = host.fit_host(plans, "X2", wide[:, :nte], fa_index=fa, chunk=500, want_lambda=True)
    _same(r, ref)
    # neither stride is 1: every second echo column of a [nvox, 2 n_te] table, every second row
    big = np.zeros((2 * nvox, 2 * nte)); big[::2, ::2] = data
    r = host.fit_host(plans, "X2", big[::2, ::2], fa_index=fa, chunk=500, want_lambda=True)
    _same(r, ref)


@gpu
def test_pinned_arrays_are_used_in_place(pkg):
    """host arrays in pinned memory (torch's pinned allocator = hipHostMalloc) take the direct path: no staging slab, same bits"""
    import torch
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    plans, alphas, _, _ = _plans(pkg, 2)
    nvox = 9_001
    data, fa, _ = synth.make_voxels(nvox, nte=32, seed=8, fa_values=alphas, device="cpu")
    ref = _reference_fit(plans[0], "X2", data.numpy(), fa.numpy())
    pin = lambda shape, dt=torch.float64: torch.empty(shape, dtype=dt, pin_memory=True)
    d_p = pin(data.shape).copy_(data); fa_p = pin(fa.shape).copy_(fa)
    out = {"fsol": pin((nvox, 60)).numpy(), "sig": pin((nvox, 32)).numpy(), "reg": pin((nvox,)).numpy(), "lam": pin((nvox,)).numpy(),
           "maps": pin((6, nvox)).numpy(), "status": pin((nvox,), torch.int32).numpy(), "fa_index": pin((nvox,)).numpy()}
    r = host.fit_host(plans, "X2", d_p.numpy(), fa_index=fa_p.numpy(), chunk=2048, want_lambda=True, out=out)
    assert r["fsol"] is out["fsol"] and r["maps"] is out["maps"]
    _same(r, ref)
    # pinned echo-major input (2-D copy) with pageable outputs
    d_t = pin((32, nvox)).copy_(data.t())
    r = host.fit_host(plans, "X2", d_t.numpy().T, fa_index=fa.numpy(), chunk=2048, want_lambda=True)
    _same(r, ref)


@gpu
def test_brute_force_fa_per_block_matches_the_whole_list(pkg):
    import torch
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    plans, alphas, _, _ = _plans(pkg, 2, nfa=16)
    nvox = 6_000
    data, _, _ = synth.make_voxels(nvox, nte=32, seed=21, fa_values=alphas, device="cpu")
    data = data.numpy()
    mask = (np.arange(nvox) % 5 != 0).astype(np.uint8)
    d = torch.as_tensor(data, device="cuda"); m = torch.as_tensor(mask, device="cuda")
    fa_ref, _, _ = plans[0].fa_bruteforce(d, m)
    ref = _reference_fit(plans[0], "X2", data, fa_ref.cpu().numpy(), mask)
    r = host.fit_host(plans, "X2", data, mask=mask, estimate_fa=True, chunk=1000, want_lambda=True)
    assert np.array_equal(r["fa_index"], fa_ref.cpu().numpy())
    _same(r, ref)


@gpu
def test_two_bins_per_lane_and_gcv_through_the_host_entry(pkg):
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    plans, alphas, _, _ = _plans(pkg, 2, nte=48, nt2=120, nfa=3)
    nvox = 3_000
    data, fa, _ = synth.make_voxels(nvox, nte=48, seed=2, fa_values=alphas, device="cpu")
    data, fa = data.numpy(), fa.numpy()
    for method in ("GCV", "BayesReg"):
        ref = _reference_fit(plans[0], method, data, fa)
        r = host.fit_host(plans, method, data, fa_index=fa, chunk=512, want_lambda=True)
        _same(r, ref)


@gpu
def test_errors_leave_nothing_in_flight(pkg):
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    lib = importlib.import_module(PKG + "._lib")
    plans, alphas, _, _ = _plans(pkg, 2)
    nvox = 5_000
    data, fa, _ = synth.make_voxels(nvox, nte=32, seed=4, fa_values=alphas, device="cpu")
    data, fa = data.numpy(), fa.numpy()
    bad = fa.copy(); bad[3_333] = 99.0                              # IndexError in the reference (motor:127-128)
    with pytest.raises(lib.Met2Error, match="plan [01].*FA index"):
        host.fit_host(plans, "X2", data, fa_index=bad, chunk=1000)
    ref = _reference_fit(plans[0], "X2", data, fa)                  # the plans are usable afterwards, on any stream
    _same(host.fit_host(plans, "X2", data, fa_index=fa, chunk=1000, want_lambda=True), ref)
    with pytest.raises(lib.Met2Error, match="twice"):
        host.fit_host([plans[0], plans[0]], "X2", data)
    other, _, _, _ = _plans(pkg, 1, nfa=3)
    with pytest.raises(lib.Met2Error, match="different shapes"):
        host.fit_host([plans[0], other[0]], "X2", data)
    with pytest.raises(lib.Met2Error, match="estimate_fa together"):
        host.fit_host(plans, "X2", data, fa_index=fa, estimate_fa=True)
    empty = host.fit_host(plans, "X2", data[:0])
    assert empty["fsol"].shape == (0, 60) and not empty["plan_ms"].any()


@gpu
def test_host_entry_through_raw_ctypes_against_the_oracle(pkg, oracle):
    """what INTEGRATION.md section 4 shows: ctypes + numpy only (no torch, no device pointer on the caller's side), checked against the oracle"""
    synth = importlib.import_module(PKG + ".synth")
    build = importlib.import_module(PKG + "._build")
    L = C.CDLL(build.LIB)
    L.met2_last_error.restype = C.c_char_p
    nte, nt2, nvox = 32, 60, 600
    T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2); alphas = np.array([150.0])
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    plans = (C.c_void_p * 2)()
    for i in range(2):
        h = C.c_void_p()
        assert L.met2_plan_create(C.byref(h), nte, nt2, 1, None) == 0, L.met2_last_error()
        assert L.met2_plan_build_dictionary_epg(h, dp(T2s), dp(T1s), C.c_double(10.0), dp(alphas), C.c_double(3000.0), None) == 0
        assert L.met2_plan_set_penalty(h, 2, dp(T2s)) == 0
        plans[i] = h
    data, _, _ = synth.make_voxels(nvox, nte=nte, seed=7, device="cpu")
    data = data.numpy()
    fsol = np.empty((nvox, nt2)); reg = np.empty(nvox); maps = np.empty((6, nvox)); ms = np.zeros(2)
    L.met2_fit_host.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32] + \
        [C.c_void_p] * 8 + [C.c_int64, C.c_void_p]
    rc = L.met2_fit_host(plans, 2, 2, nvox, data.ctypes.data, None, nte, 1, None, None, None, 0, fsol.ctypes.data, None, reg.ctypes.data, None,
                         maps.ctypes.data, None, None, None, 128, ms.ctypes.data)
    assert rc == 0, L.met2_last_error()
    for i in range(2):
        assert L.met2_plan_destroy(C.c_void_p(plans[i])) == 0
    D = oracle.dictionary_fa_major(nt2, T2s, T1s, nte, 10.0, alphas, 3000.0)
    Lp = oracle.penalty(nt2, "L2")
    fs, _, rg, _ = oracle.fit_batch("X2", D, Lp, data, np.zeros(nvox), np.ones(nvox))
    err = np.max(np.abs(fsol - fs), axis=1) / np.max(np.abs(fs), axis=1)
    assert (err < 1e-5).mean() > 0.995, (err > 1e-5).sum()          # tolerance of north_star: 1e-5 relative; Brent ties are ~1e-4 of voxels
    ok = err < 1e-5
    mwf = oracle.metrics(fs, T2s, np.ones(nvox))["MWF"]
    assert np.max(np.abs(maps[0][ok] - mwf[ok])) < 1e-5
    assert (ms > 0).all()


@gpu
@pytest.mark.parametrize("count", [1, 2])
def test_spline_fa_and_a_separate_fa_volume(pkg, count):
    """estimate_fa = 2 (fa_estimation.py:35-70 per block, coarse plans attached) and fa_data (the volume the FA step sees, motor:337-343):
    the flip angles of Met2Plan.fa_spline over the whole list, the fits of met2_fit with them"""
    import torch
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    lib = importlib.import_module(PKG + "._lib")
    nte, nt2 = 32, 60
    alpha_hr = np.linspace(90.0, 180.0, 91)
    alpha_lr = np.linspace(90.0, 180.0, 15)                          # motor:237
    T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2)
    plans, coarse = [], []
    for _ in range(count):
        p = pkg.Met2Plan(nte, nt2, 91); p.build_dictionary_epg(T2s, T1s, 10.0, alpha_hr, 3000.0).set_penalty("L2", T2s)
        q = pkg.Met2Plan(nte, nt2, 15); q.build_dictionary_epg(T2s, T1s, 10.0, alpha_lr, 3000.0)
        plans.append(p); coarse.append(q)
    nvox = 5_000
    data, _, _ = synth.make_voxels(nvox, nte=nte, seed=33, fa_values=alpha_hr[30:], device="cpu")
    data = data.numpy()
    rng = np.random.default_rng(1)
    smooth = np.abs(data * (1.0 + 0.01 * rng.standard_normal(data.shape)))          # stands for the Gaussian-smoothed volume
    mask = (np.arange(nvox) % 6 != 1).astype(np.uint8)
    with pytest.raises(lib.Met2Error, match="attach_fa_spline"):
        host.fit_host(plans, "X2", data, estimate_fa="spline")
    host.attach_fa_spline(plans, coarse, alpha_lr, alpha_hr)
    d = torch.as_tensor(data, device="cuda"); sm = torch.as_tensor(smooth, device="cuda"); m = torch.as_tensor(mask, device="cuda")
    for fa_src, fa_np in ((d, None), (sm, smooth)):
        fa_ref, _, _ = plans[0].fa_spline(coarse[0], alpha_lr, alpha_hr, fa_src, m, want_km=False)
        ref = _reference_fit(plans[0], "X2", data, fa_ref.cpu().numpy(), mask)
        r = host.fit_host(plans, "X2", data, mask=mask, estimate_fa="spline", fa_data=fa_np, chunk=1200, want_lambda=True)
        assert np.array_equal(r["fa_index"], fa_ref.cpu().numpy())
        _same(r, ref)
    # brute force on the separate FA volume, Fortran-ordered pair
    vol = np.asfortranarray(data.reshape(10, 20, 25, nte)); vol_s = np.asfortranarray(smooth.reshape(10, 20, 25, nte))
    flat = lambda a: a.reshape(-1, nte, order="F")                                # voxels in memory order
    fa_ref, _, _ = plans[0].fa_bruteforce(torch.as_tensor(np.ascontiguousarray(flat(vol_s)), device="cuda"))
    ref = _reference_fit(plans[0], "X2", flat(vol), fa_ref.cpu().numpy())
    r = host.fit_host(plans, "X2", vol, estimate_fa="brute-force", fa_data=vol_s, chunk=1200, want_lambda=True)
    assert np.array_equal(r["fa_index"], fa_ref.cpu().numpy())
    _same(r, ref)
    with pytest.raises(lib.Met2Error, match="fa_data"):
        host.fit_host(plans, "X2", data, fa_data=smooth)
    host.attach_fa_spline(plans, None, None, None)
    with pytest.raises(lib.Met2Error, match="attach_fa_spline"):
        host.fit_host(plans, "X2", data, estimate_fa="spline")
    for p in plans + coarse:
        p.close()


@gpu
@pytest.mark.parametrize("order,denoise,smooth,fa_method,devices", [("C", "None", "no", "spline", [0, 0]), ("F", "None", "no", "brute-force", [0, 0, 0]),
                                                                    ("F", "TV", "yes", "spline", [0, 0]), ("C", "NESMA", "yes", "brute-force", [0]),
                                                                    ("C", "None", "no", "given", [0, 0])])
def test_driver_with_a_device_list_equals_the_default_driver(pkg, order, denoise, smooth, fa_method, devices):
    """recon_met2_arrays(devices=[...]): one process, one plan per listed device through met2_fit_host -- the ten outputs of the default
    (torch-pipelined, one device) driver bit for bit; mask values 0 / 2, negative samples, both memory orders"""
    motor = importlib.import_module(PKG + ".motor")
    synth = importlib.import_module(PKG + ".synth")
    dims = (14, 12, 13)
    nvox = int(np.prod(dims))
    alphas = np.linspace(90.0, 180.0, 91)
    data, fa, _ = synth.make_voxels(nvox, nte=32, seed=91, fa_values=alphas, device="cuda")
    vol = data.cpu().numpy().reshape(dims + (32,))
    rng = np.random.default_rng(5)
    vol[rng.integers(0, 14, 20), rng.integers(0, 12, 20), rng.integers(0, 13, 20), rng.integers(0, 32, 20)] *= -1.0
    mask = np.ones(dims, dtype=np.int64); mask[::5, ::3, :] = 0; mask[1, 1, 1] = 2
    if order == "F":
        vol = np.asfortranarray(vol)
    TE = 10.0 * np.arange(1, 33)
    fa_given = fa.cpu().numpy().reshape(dims) if fa_method == "given" else None
    method = "brute-force" if fa_method == "given" else fa_method
    from tools import torch_pipeline as tp                             # the torch pipeline of rounds 3-4 (retired from the product): the comparator
    keep = motor.PIPELINE_CHUNK
    try:
        motor.PIPELINE_CHUNK = 700
        ref = tp.recon_met2_arrays(vol, mask, TE, 3000.0, "X2", "L2", method, 40.0, denoise=denoise, FA_smooth=smooth, fa_index=fa_given, chunk=700)
        got = motor.recon_met2_arrays(vol, mask, TE, 3000.0, "X2", "L2", method, 40.0, denoise=denoise, FA_smooth=smooth, fa_index=fa_given, devices=devices)
        dflt = motor.recon_met2_arrays(vol, mask, TE, 3000.0, "X2", "L2", method, 40.0, denoise=denoise, FA_smooth=smooth, fa_index=fa_given)
    finally:
        motor.PIPELINE_CHUNK = keep
    assert np.array_equal(dflt["fsol_4D"], ref["fsol_4D"], equal_nan=True) and np.array_equal(dflt["FA"], ref["FA"])
    for k in ("fsol_4D", "Est_Signal", "reg_param", "FA_index", "FA", "MWF", "IEWF", "FWF", "T2_M", "T2_IE", "TWC"):
        assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k], equal_nan=True), k
    if denoise == "TV":                                             # the file-level driver saves the denoised volume (motor:302-303)
        a = tp.recon_met2_arrays(vol, mask, TE, 3000.0, "X2", "L2", method, 40.0, denoise=denoise, FA_smooth=smooth, return_prepared=True)
        b = motor.recon_met2_arrays(vol, mask, TE, 3000.0, "X2", "L2", method, 40.0, denoise=denoise, FA_smooth=smooth, return_prepared=True, devices=devices)
        assert np.array_equal(a["data_prepared"], b["data_prepared"]) and np.array_equal(a["MWF"], b["MWF"], equal_nan=True)


@gpu
def test_mask_values_and_the_fa_gate(pkg):
    """mask_values: data *= mask, clip at 0 on the device per block (motor:180-182, :279); fa_gate: mask and a positive echo sum of what the
    FA step sees (fa_estimation.py:45)"""
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    plans, alphas, _, _ = _plans(pkg, 2, nfa=16)
    nvox = 7_000
    data, _, _ = synth.make_voxels(nvox, nte=32, seed=17, fa_values=alphas, device="cpu")
    data = data.numpy()
    rng = np.random.default_rng(3)
    data[rng.integers(0, nvox, 50), rng.integers(0, 32, 50)] *= -1.0
    mv = rng.choice([0.0, 1.0, 1.0, 1.0, 2.0], nvox)
    mask = (mv > 0).astype(np.uint8)
    prepared = np.maximum(data * mv[:, None], 0.0)
    for layout in ("C", "F"):
        raw = data if layout == "C" else np.asfortranarray(data)
        ref = host.fit_host(plans, "X2", prepared, mask=mask, estimate_fa=True, chunk=1500, want_lambda=True, want_gate=True)
        got = host.fit_host(plans, "X2", raw, mask=mask, mask_values=mv, estimate_fa=True, chunk=1500, want_lambda=True, want_gate=True)
        _same(got, ref, keys=("fsol", "sig", "reg", "lam", "maps", "status", "fa_index", "fa_gate"))
        assert np.array_equal(got["fa_gate"], ((prepared.sum(axis=1) > 0) & (mask > 0)).astype(np.float64))
        assert np.array_equal(raw, data)                                # the caller's array is never written


@gpu
def test_concurrent_calls_pool_reuse_and_trim(pkg):
    """two host threads in met2_fit_host at once (each with its own plan); a destroyed plan's block buffers serve the next plan on the device;
    met2_host_trim frees what waits; a different shape adopts pooled buffers and regrows them"""
    import threading
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    lib = importlib.import_module(PKG + "._lib")
    plans, alphas, _, _ = _plans(pkg, 2)
    nvox = 12_000
    data, fa, _ = synth.make_voxels(nvox, nte=32, seed=41, fa_values=alphas, device="cpu")
    data, fa = data.numpy(), fa.numpy()
    ref = _reference_fit(plans[0], "X2", data, fa)
    res = [None, None]

    def work(i):
        for _ in range(3):
            res[i] = host.fit_host(plans[i], "X2", data, fa_index=fa, chunk=2000 + 500 * i, want_lambda=True)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    _same(res[0], ref); _same(res[1], ref)
    for p in plans:
        p.close()                                                    # their buffers wait for the next plans on device 0
    again, _, _, _ = _plans(pkg, 2)
    _same(host.fit_host(again, "X2", data, fa_index=fa, chunk=1500, want_lambda=True), ref)
    # another shape on the same device: NNLS at 48 x 120 adopts the pooled buffers and regrows them
    for p in again:
        p.close()
    big, alphas2, _, _ = _plans(pkg, 1, nte=48, nt2=120, nfa=3)
    d2, fa2, _ = synth.make_voxels(3_000, nte=48, seed=5, fa_values=alphas2, device="cpu")
    _same(host.fit_host(big, "T2SPARC", d2.numpy(), fa_index=fa2.numpy(), chunk=700, want_lambda=True), _reference_fit(big[0], "T2SPARC", d2.numpy(), fa2.numpy()))
    big[0].close()
    assert lib.lib().met2_host_trim() == 0 and lib.lib().met2_host_trim() == 0
    fresh, _, _, _ = _plans(pkg, 1)
    _same(host.fit_host(fresh, "NNLS", data, fa_index=fa, want_lambda=True), _reference_fit(fresh[0], "NNLS", data, fa))


@gpu
def test_full_size_volume_through_the_host_entry(pkg):
    """configs[2]'s size (200 x 200 x 128 = 5.12 M voxels, L-curve / L1): byte offsets beyond 2^31 in every big array, twenty blocks over two plans,
    pageable arrays -- equal to one met2_fit over the list on every output, every status word FITTED"""
    import torch
    host = importlib.import_module(PKG + ".host")
    synth = importlib.import_module(PKG + ".synth")
    plans, alphas, _, _ = _plans(pkg, 2, nfa=1, penalty="L1")
    nvox = 200 * 200 * 128
    data, _, _ = synth.make_voxels(nvox, nte=32, seed=20260103, device="cuda")
    out = plans[0].fit("L_curve", data, want_lambda=True)
    torch.cuda.synchronize()
    ref = {k: v.cpu().numpy() for k, v in out.items()}
    del out
    d = data.cpu().numpy()
    del data
    res = host.fit_host(plans, "L_curve", d, want_lambda=True)
    _same(res, ref)
    assert (res["status"] == 1).all()
    assert res["plan_ms"].min() > 0
