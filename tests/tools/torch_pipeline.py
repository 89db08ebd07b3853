"""The torch three-stream host pipeline of rounds 3-4 (motor.fit_host_pipeline, motor._recon_pipelined), retired from the product in round 5:
the product has ONE host pipeline, the C one (csrc/met2_host.hip: met2_fit_host).  Kept here as the tests' bit-equality comparator -- it drives
the same device entries (Met2Plan.fit(sync=False) / fa_bruteforce / fa_spline / finish) from Python."""
import importlib
import math

import numpy as np
import torch

PKG = "multicomponent-t2-toolbox_amd"
motor = importlib.import_module(PKG + ".motor")
Met2Plan = importlib.import_module(PKG).Met2Plan
MAP_NAMES = motor.MAP_NAMES
_match_layout = motor._match_layout

_COPY_POOL = None


def _host_copy(dst, src):
    """dst <- src, two CPU tensors of one shape (dst pinned): numpy's memcpy on four plain threads.  NOT Tensor.copy_: torch spreads a large CPU
    copy over its intra-op pool -- one thread per core of the machine -- whose workers spin after every parallel region; inside a container with a CPU
    quota (this GPU box: 16 of 256 cores) that burns the quota within the scheduler period and the whole process is throttled until the next one:
    measured, the driver's host thread stood 60-100 ms at a time in whichever call it was in (a 17 MB memcpy, an event wait)."""
    global _COPY_POOL
    d, s = dst.numpy(), src.numpy()
    if d.nbytes < (4 << 20) or d.shape[0] < 4:
        np.copyto(d, s)
        return dst
    if _COPY_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _COPY_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix="met2-stage")
    n0 = d.shape[0]
    cuts = [n0 * i // 4 for i in range(5)]
    for f in [_COPY_POOL.submit(np.copyto, d[a:b], s[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]:
        f.result()
    return dst


def fit_host_pipeline(plan, reg_method, host_data, fa_method=None, fa_index=None, mask=None, chunk=262144, want_lambda=False, out=None,
                      echo_major=False, mask_values=None, device_data=None, device_fa_data=None):
    """Driver steps 2-4 (motor:349-373, 427-472) for a voxel list that lives in HOST memory, as the reference's driver holds it
    (motor:167-182): chunks of `chunk` voxels go H2D on one stream, through [FA estimation and] the fit on a second, and the
    outputs D2H on a third, double-buffered, so that the copies of chunks c+1 and c-1 run under the fit of chunk c.  The fits are
    only enqueued (met2_fit_enqueue_strided); one met2_plan_finish at the end waits and reports.
      host_data [nvox, n_te] float64 (or [n_te, nvox] with echo_major=True: the flattened Fortran-ordered volume nibabel hands the
                driver): a pinned torch CPU tensor is copied from in place; pageable memory (numpy arrays, ordinary tensors) is
                staged through two pinned chunk buffers
      fa_method None (fa_index given, or flip angle 0 for all) | 'brute-force' | a callable (chunk [n, n_te] view on the device, mask
                chunk or None) -> float64 FA-index tensor [n] (the spline method of recon_met2_arrays)
      fa_index, mask: host arrays [nvox] or None; mask gates (mask != 0)
      mask_values  host array [nvox]: the driver's preparation on the device -- every echo is multiplied by it and negative values are
                clipped to 0 (motor:180-182, :279) before anything else sees the chunk
      device_data  instead of host_data (pass None for it): the prepared voxel list already ON THE DEVICE ([nvox, n_te], or [n_te, nvox] with
                echo_major=True) -- what is left of the driver after a whole-volume filter (TV, NESMA, FA smoothing).  Nothing is uploaded but
                the per-voxel arrays; the chunks are cut out of it in place and only the output side of the pipeline overlaps with the fits.
                device_fa_data: the same list as the FA step shall see it (the Gaussian-smoothed volume, motor:337-343); default: device_data
      out       a dict this function returned earlier for the same shapes: its pinned buffers are written again (pinning 0.8 GB of
                host memory costs tens of ms; torch's caching host allocator does the same for buffers that were freed)
    Returns pinned CPU tensors: fsol [nvox, n_t2], sig [nvox, n_te], reg [nvox], maps [6, nvox], status [nvox] (int32), fa_index
    [nvox], fa_gate [nvox] (1 where the FA step's gate holds, fa_estimation.py:45: mask and a positive echo sum) and lam when asked;
    chunking changes nothing in them (every voxel is solved on its own)."""
    dev = plan.device
    nte, nt2 = plan.n_te, plan.n_t2
    want_shape = "[n_te=%d, nvox]" % nte if echo_major else "[nvox, n_te=%d]" % nte
    on_device = device_data is not None
    if on_device:
        if host_data is not None or mask_values is not None:
            raise ValueError("device_data stands in for host_data and is already prepared (no mask_values)")
        src = device_data
        for t in (src,) if device_fa_data is None else (src, device_fa_data):
            if not (torch.is_tensor(t) and t.device == dev and t.dtype == torch.float64 and t.dim() == 2 and t.shape[0 if echo_major else 1] == nte
                    and t.is_contiguous() and t.shape == src.shape):
                raise ValueError("device_data / device_fa_data must be contiguous float64 %s tensors on %s" % (want_shape, dev))
    else:
        if device_fa_data is not None:
            raise ValueError("device_fa_data goes with device_data")
        src = host_data if torch.is_tensor(host_data) else torch.from_numpy(host_data)
        ok_shape = src.dim() == 2 and src.shape[0 if echo_major else 1] == nte
        if src.is_cuda or src.dtype != torch.float64 or not ok_shape or not src.is_contiguous():
            raise ValueError("host_data must be a contiguous float64 %s array in host memory" % want_shape)
    nvox = int(src.shape[1 if echo_major else 0])
    chunk = max(1, min(int(chunk), max(nvox, 1)))
    nch = (nvox + chunk - 1) // chunk
    pin = lambda shape, dt=torch.float64: torch.empty(shape, dtype=dt, pin_memory=True)
    shapes = {"fsol": ((nvox, nt2), torch.float64), "sig": ((nvox, nte), torch.float64), "reg": ((nvox,), torch.float64),
              "maps": ((6, nvox), torch.float64), "status": ((nvox,), torch.int32), "fa_index": ((nvox,), torch.float64),
              "fa_gate": ((nvox,), torch.float64)}
    if want_lambda:
        shapes["lam"] = ((nvox,), torch.float64)
    res = {}
    for name, (shp, dt) in shapes.items():
        t = None if out is None else out.get(name)
        ok = t is not None and tuple(t.shape) == shp and t.dtype == dt and t.is_pinned() and t.is_contiguous()
        res[name] = t if ok else pin(shp, dt)
    if nvox == 0:
        return res
    # the per-voxel host arrays go through PINNED copies: an H2D copy from pageable memory is not asynchronous -- it waits on the host for
    # everything its stream waits for, here the fit of chunk c - 2 (measured: up to 56 ms of the host standing in a 2 MB copy)
    def as1d(a, dt):
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a).reshape(-1).astype(dt, copy=False)))
        return _host_copy(pin(t.shape, t.dtype), t)
    fa_h = None if fa_index is None else as1d(fa_index, np.float64)
    mk_h = None if mask is None else as1d(np.asarray(mask).reshape(-1) != 0, np.uint8)
    mv_h = None if mask_values is None else as1d(mask_values, np.float64)
    in_shape = (nte, chunk) if echo_major else (chunk, nte)
    stage = None if (on_device or src.is_pinned()) else [pin(in_shape), pin(in_shape)]
    dv = lambda shape, dt=torch.float64: torch.empty(shape, dtype=dt, device=dev)
    cut = (lambda t, n: t[:, :n]) if echo_major else (lambda t, n: t[:n])         # the first n voxels of a chunk buffer
    with torch.cuda.device(dev):
        s_in, s_fit, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        d_in = [None, None] if on_device else [dv(in_shape), dv(in_shape)]
        d_fa = [dv((chunk,)), dv((chunk,))]
        d_gate = [dv((chunk,)), dv((chunk,))]
        d_mk = [dv((chunk,), torch.uint8), dv((chunk,), torch.uint8)] if mk_h is not None else [None, None]
        d_mv = [dv((chunk,)), dv((chunk,))] if mv_h is not None else [None, None]
        d_out = [{"fsol": dv((chunk, nt2)), "sig": dv((chunk, nte)), "reg": dv((chunk,)), "lam": dv((chunk,)), "maps": dv((6, chunk)),
                  "status": dv((chunk,), torch.int32)} for _ in range(2)]
        ones_te = torch.ones(nte, dtype=torch.float64, device=dev)
        ev_in = [torch.cuda.Event(), torch.cuda.Event()]
        ev_fit = [torch.cuda.Event(), torch.cuda.Event()]
        ev_out = [torch.cuda.Event(), torch.cuda.Event()]
        s_in.wait_stream(torch.cuda.current_stream(dev))

        def upload(c):
            k = c & 1
            lo, hi = c * chunk, min(nvox, (c + 1) * chunk)
            n = hi - lo
            h = None if on_device else (src[:, lo:hi] if echo_major else src[lo:hi])
            if stage is not None:
                if c >= 2:
                    ev_in[k].synchronize()          # the H2D of chunk c - 2 has left this staging buffer (it finished before that chunk's fit began)
                _host_copy(cut(stage[k], n), h)     # pageable -> pinned (host memcpy, one segment per echo when echo-major)
                h = cut(stage[k], n)
            with torch.cuda.stream(s_in):
                if c >= 2:                          # the device slot is free once chunk c - 2 has been fitted and its outputs (fa_index) copied out:
                    s_in.wait_event(ev_fit[k])      # waited for on the GPU, not by the host -- the host goes on staging while the GPU fits
                    s_in.wait_event(ev_out[k])
                if not on_device:
                    cut(d_in[k], n).copy_(h, non_blocking=True)
                if fa_h is not None:
                    d_fa[k][:n].copy_(fa_h[lo:hi], non_blocking=True)
                if mk_h is not None:
                    d_mk[k][:n].copy_(mk_h[lo:hi], non_blocking=True)
                if mv_h is not None:
                    d_mv[k][:n].copy_(mv_h[lo:hi], non_blocking=True)
                ev_in[k].record(s_in)

        try:
            upload(0)
            for c in range(nch):
                k = c & 1
                lo, hi = c * chunk, min(nvox, (c + 1) * chunk)
                n = hi - lo
                with torch.cuda.stream(s_fit):
                    s_fit.wait_event(ev_in[k])
                    if c >= 2:
                        s_fit.wait_event(ev_out[k])     # the outputs of chunk c - 2 have left this slot
                    o = d_out[k]
                    raw = (src[:, lo:hi] if echo_major else src[lo:hi]) if on_device else cut(d_in[k], n)
                    if mv_h is not None:                # motor:180-182, :279 on the device, in place
                        raw.mul_(d_mv[k][:n].unsqueeze(0) if echo_major else d_mv[k][:n].unsqueeze(1))
                        raw.clamp_(min=0.0)
                    dd = raw.t() if echo_major else raw # [n, n_te] either way (echo-major: a strided view, read in place)
                    dd_fa = dd                          # what the FA step sees (fa_estimation.py:45 gates on ITS echo sum)
                    if device_fa_data is not None:
                        dd_fa = device_fa_data[:, lo:hi].t() if echo_major else device_fa_data[lo:hi]
                    mk = None if mk_h is None else d_mk[k][:n]
                    gate = torch.mv(dd_fa, ones_te) > 0  # the echo sum as a matrix-vector product: torch's row reduction of a [n, 32] array took 1.6 ms per chunk, this 0.1
                    d_gate[k][:n].copy_(gate if mk is None else (gate & (mk != 0)))
                    if fa_method == "brute-force":
                        fa, _, _ = plan.fa_bruteforce(dd_fa, mk)
                        d_fa[k][:n].copy_(fa)
                    elif callable(fa_method):
                        d_fa[k][:n].copy_(fa_method(dd_fa, mk))
                    elif fa_h is None:
                        d_fa[k][:n].zero_()
                    # the chunk's maps are [6, n]: a contiguous [6 * n] prefix of the slot's buffer viewed as [6, n]
                    maps_v = o["maps"].reshape(-1)[: 6 * n].view(6, n)
                    plan.fit(reg_method, dd, fa_index=d_fa[k][:n], mask=mk, sync=False,
                             out={"fsol": o["fsol"][:n], "sig": o["sig"][:n], "reg": o["reg"][:n], "lam": o["lam"][:n], "maps": maps_v, "status": o["status"][:n]},
                             want_lambda=True)
                    ev_fit[k].record(s_fit)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(ev_fit[k])
                    res["fsol"][lo:hi].copy_(o["fsol"][:n], non_blocking=True)
                    res["sig"][lo:hi].copy_(o["sig"][:n], non_blocking=True)
                    res["reg"][lo:hi].copy_(o["reg"][:n], non_blocking=True)
                    res["status"][lo:hi].copy_(o["status"][:n], non_blocking=True)
                    res["fa_index"][lo:hi].copy_(d_fa[k][:n], non_blocking=True)
                    res["fa_gate"][lo:hi].copy_(d_gate[k][:n], non_blocking=True)
                    if want_lambda:
                        res["lam"][lo:hi].copy_(o["lam"][:n], non_blocking=True)
                    for i in range(6):
                        res["maps"][i, lo:hi].copy_(maps_v[i], non_blocking=True)
                    ev_out[k].record(s_out)
                if c + 1 < nch:
                    upload(c + 1)                       # staged and enqueued while the GPU works on chunk c
            with torch.cuda.stream(s_fit):
                plan.finish()
            s_out.synchronize()
            torch.cuda.current_stream(dev).wait_stream(s_fit)
        except BaseException:
            # the device buffers of this frame go back to the caching allocator as it unwinds: nothing may still be in flight on them
            # (copies on s_out, fits enqueued before a later plan.fit raised), and the plan must not keep a pending error word
            for st_ in (s_in, s_fit, s_out):
                st_.synchronize()
            try:
                with torch.cuda.stream(s_fit):
                    plan.finish()
            except Exception:
                pass
            raise
    return res


def _recon_pipelined(data, mask, TE_array, TR, reg_method, reg_matrix, FA_method, myelin_T2, fa_index, device, plan, prepared, on_device=None, chunk=262144):
    """recon_met2_arrays without denoising or FA smoothing (nothing needs the whole volume at once): the host volume streams through
    fit_host_pipeline in its own memory order -- C-ordered [.., nt] voxel-major, Fortran-ordered (nibabel's) echo-major -- and the
    outputs land in pinned host buffers that are returned as numpy views, reshaped to the volume.  Same numbers as the one-shot
    path (`test_driver_pipeline_equals_one_shot`).
    on_device = (dd, dd_fa or None): the prepared (and filtered) volume as a tensor on the device, and the smoothed one the FA step shall
    see -- the part of a denoised / FA-smoothed run that comes after its whole-volume filters; `data` then only gives the shape."""
    _plan = importlib.import_module(PKG + ".plan")
    unflatten, voxel_layout = _plan.unflatten, _plan.voxel_layout
    vol_shape = data.shape[:-1]
    nt = data.shape[-1]
    nvox = int(np.prod(vol_shape))
    order = "C" if data.flags.c_contiguous else "F"
    dev_src = dev_fa = None
    if on_device is not None:
        dd, dd_fa = on_device
        dd, _, _, _, _, order = voxel_layout(dd, nt)
        rev = list(reversed(range(dd.dim())))
        flat2 = (lambda t: t.reshape(nvox, nt)) if order == "C" else (lambda t: t.permute(*rev).reshape(nt, nvox))
        dev_src = flat2(dd)
        dev_fa = None if dd_fa is None else flat2(_match_layout(dd_fa, order))
    TE_array = np.asarray(TE_array, dtype=np.float64)
    tau = float(TE_array[1] - TE_array[0])
    Npc = 96 if reg_method == "T2SPARC" else 60
    T2s = np.logspace(math.log10(10.0), math.log10(2000.0), num=Npc, endpoint=True, base=10.0)
    T1s = 1000.0 * np.ones_like(T2s)
    alpha_values = np.linspace(90.0, 180.0, 91 * 3 if FA_method == "spline" else 91)      # motor:231-244
    flat = (lambda a: np.asarray(a).reshape(-1, order=order))                               # per-voxel arrays in the data's voxel order
    src = None
    if on_device is None:
        src = data.reshape(nvox, nt) if order == "C" else data.reshape(nvox, nt, order="F").T   # views: [nvox, nt] or echo-major [nt, nvox]
    mvals = None if prepared else flat(mask).astype(np.float64)
    own = plan is None
    plan_lr = None
    if own:
        plan = Met2Plan(nt, Npc, alpha_values.shape[0], device=device, myelin_T2=myelin_T2)
        plan.build_dictionary_epg(T2s, T1s, tau, alpha_values, TR)
        plan.set_penalty("InvT2" if reg_method == "T2SPARC" else reg_matrix, T2s)   # run_real_data_script.py:91-93
    try:
        fa_m = None
        if fa_index is None:
            if FA_method == "spline":
                alpha_lr = np.linspace(90.0, 180.0, 15)                                     # motor:237
                plan_lr = Met2Plan(plan.n_te, plan.n_t2, 15, device=plan.device.index or 0)
                plan_lr.build_dictionary_epg(T2s, T1s, tau, alpha_lr, TR)
                fa_m = lambda dd, mk: plan.fa_spline(plan_lr, alpha_lr, alpha_values, dd, mk, want_km=False)[0]
            else:
                fa_m = "brute-force"
        out = fit_host_pipeline(plan, reg_method, src, fa_method=fa_m, fa_index=None if fa_index is None else flat(fa_index),
                                mask=flat(mask) > 0, chunk=chunk, echo_major=(order == "F"), mask_values=mvals,
                                device_data=dev_src, device_fa_data=dev_fa)
        vol = lambda t, lead=0: unflatten(t, vol_shape, order, lead=lead).numpy()
        res = {"fsol_4D": vol(out["fsol"]), "Est_Signal": vol(out["sig"]), "reg_param": vol(out["reg"]), "FA_index": vol(out["fa_index"])}
        fitted_fa = vol(out["fa_gate"]) > 0                    # gate of the FA step (fa_estimation.py:45), formed on the prepared chunk
        res["FA"] = np.where(fitted_fa, alpha_values[res["FA_index"].astype(int)], 0.0)
        maps = vol(out["maps"], lead=1)
        for i, name in enumerate(MAP_NAMES):
            res[name] = maps[i]
        res["T2s"] = T2s
        return res
    finally:
        if plan_lr is not None:
            plan_lr.close()
        if own:
            plan.close()




def recon_met2_arrays(data, mask, TE_array, TR, reg_method="X2", reg_matrix="L2", FA_method="brute-force", myelin_T2=40.0, fa_index=None, device=0,
                      denoise="None", prepared=False, FA_smooth="no", return_prepared=False, chunk=262144):
    """The driver as rounds 3-4 ran it without the C ABI's host entry: a plain run streams through fit_host_pipeline; a denoised / FA-smoothed
    run filters on the device and then streams the device-resident voxel list; return_prepared=True takes the product's one-piece path (a
    caller's own plan)."""
    data = np.asarray(data, dtype=np.float64)
    vol_shape = data.shape[:-1]
    nt = data.shape[-1]
    mask = np.asarray(mask).reshape(vol_shape)
    if return_prepared:
        TE = np.asarray(TE_array, dtype=np.float64)
        Npc = 96 if reg_method == "T2SPARC" else 60
        T2s = np.logspace(math.log10(10.0), math.log10(2000.0), num=Npc, endpoint=True, base=10.0)
        alpha_values = np.linspace(90.0, 180.0, 91 * 3 if FA_method == "spline" else 91)
        plan = Met2Plan(nt, Npc, alpha_values.shape[0], device=device, myelin_T2=myelin_T2)
        try:
            plan.build_dictionary_epg(T2s, 1000.0 * np.ones_like(T2s), float(TE[1] - TE[0]), alpha_values, TR)
            plan.set_penalty("InvT2" if reg_method == "T2SPARC" else reg_matrix, T2s)
            return motor.recon_met2_arrays(data, mask, TE_array, TR, reg_method, reg_matrix, FA_method, myelin_T2, fa_index=fa_index, device=device, plan=plan,
                                           denoise=denoise, prepared=prepared, FA_smooth=FA_smooth, return_prepared=True)
        finally:
            plan.close()
    plain = denoise in ("None", None, "none") and not (FA_smooth == "yes" and fa_index is None)
    if plain:
        if not (data.flags.c_contiguous or data.flags.f_contiguous):
            data = np.ascontiguousarray(data)
        return _recon_pipelined(data, mask, TE_array, TR, reg_method, reg_matrix, FA_method, myelin_T2, fa_index, device, None, prepared, chunk=chunk)
    dev = torch.device("cuda", device)
    dd, mk = motor._prepare_volume(data, mask, dev, prepared, denoise)
    dd_fa = dd
    if FA_smooth == "yes" and fa_index is None:
        dd_fa = motor.gaussian_smooth(dd, 2.0)
    return _recon_pipelined(data, mask, TE_array, TR, reg_method, reg_matrix, FA_method, myelin_T2, fa_index, device, None, True,
                            on_device=(dd, dd_fa if dd_fa is not dd else None), chunk=chunk)
