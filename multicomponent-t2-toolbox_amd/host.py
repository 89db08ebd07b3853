"""fit_host: numpy in, numpy out, one or several GPUs from ONE process -- the Python face of met2_fit_host (include/met2_hip.h, ABI 5).

The reference's driver holds its volume in host memory (motor:167-182) and loops over image rows in one process (motor:427-441);
this is that loop handed to the library: blocks of voxels are dealt to the plans (one per device, each driven by its own host thread
inside the C call), copied in, fitted and copied out on three streams per device.  torch is not on this call path (ctypes + numpy; the package
imports it for Met2Plan, which owns the plan handles): the arrays are the caller's numpy arrays, staged through pinned block buffers inside the
library (arrays that already live in pinned memory are used in place; a float64 tensor on the device may stand where the volume does)."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .plan import METHODS

_vp = C.c_void_p


def _p(a):
    if a is None:
        return _vp(0)
    return _vp(a.data_ptr()) if hasattr(a, "data_ptr") else _vp(a.ctypes.data)


class _TensorView:
    """shape / strides / flags of a torch-like tensor (anything with data_ptr(), shape, stride(), element_size()) in numpy's terms, so that a
    float64 volume on a device can stand where the host volume does (data, fa_data) without this module importing torch"""

    def __init__(self, t):
        if "float64" not in str(t.dtype):
            raise ValueError("device volumes must be float64")
        self.t, self.shape, self.ndim, self.dtype = t, tuple(int(v) for v in t.shape), len(t.shape), np.dtype(np.float64)
        self.strides = tuple(int(v) * 8 for v in t.stride())
        c, f, acc = True, True, 8
        for n, st in zip(reversed(self.shape), reversed(self.strides)):
            c = c and (n == 1 or st == acc); acc *= n
        acc = 8
        for n, st in zip(self.shape, self.strides):
            f = f and (n == 1 or st == acc); acc *= n

        class _F:
            c_contiguous, f_contiguous = c, f
        self.flags = _F()

    def data_ptr(self):
        return self.t.data_ptr()


FA_MODES = {False: 0, None: 0, 0: 0, True: 1, 1: 1, "brute-force": 1, 2: 2, "spline": 2}


def attach_fa_spline(plans, plans_lr, alpha_lr, alpha_hr):
    """met2_plan_attach_fa_spline on every plan: plans_lr[i] holds the coarse-grid dictionary (motor:237-238) on plans[i]'s device; the coarse
    plans must stay alive while attached (plans_lr=None detaches)."""
    plans = list(plans) if isinstance(plans, (list, tuple)) else [plans]
    if plans_lr is None:
        for p in plans:
            check(lib().met2_plan_attach_fa_spline(p._h, None, 0, None, 0, None))
        return
    plans_lr = list(plans_lr) if isinstance(plans_lr, (list, tuple)) else [plans_lr]
    if len(plans_lr) != len(plans):
        raise ValueError("one coarse plan per plan")
    al = np.ascontiguousarray(alpha_lr, dtype=np.float64); ah = np.ascontiguousarray(alpha_hr, dtype=np.float64)
    dp = C.POINTER(C.c_double)
    for p, q in zip(plans, plans_lr):
        check(lib().met2_plan_attach_fa_spline(p._h, q._h, al.shape[0], al.ctypes.data_as(dp), ah.shape[0], ah.ctypes.data_as(dp)))


def fit_host(plans, method, data, fa_index=None, mask=None, estimate_fa=False, chunk=0, want_sig=True, want_maps=True, want_status=True,
             want_lambda=False, out=None, fa_data=None, mask_values=None, want_gate=False):
    """plans: a Met2Plan or a sequence of them (same shape, configured alike; one per device for a multi-GPU run).
    data: float64 numpy array [..., n_te] (or a float64 tensor on a device: anything with data_ptr() / shape / stride()) -- a voxel list or a volume, C-ordered or the Fortran-ordered array nibabel delivers (read as
    it lies); any other layout is gathered inside the library.  fa_index / mask: per voxel, in the memory order of `data`'s voxels.
    estimate_fa=True / 'brute-force': brute-force FA search over the plans' FA axis per block (fa_estimation.py:74-111); 'spline': the spline
    method on the coarse plans attached with attach_fa_spline (fa_estimation.py:35-70).  fa_data: the same voxels as the FA step shall see them
    (the smoothed volume of motor:337-343), same shape and memory layout as `data`.  mask_values: per voxel, the driver's preparation on
    the device (every echo times it, negatives clipped to 0: motor:180-182, :279).  want_gate: also return fa_gate [nvox] (1.0 where the FA
    step's gate holds, fa_estimation.py:45).
    Returns numpy arrays flat in the memory order of the voxels: fsol [nvox, n_t2], sig [nvox, n_te], reg [nvox], maps [6, nvox],
    status [nvox] int32, lam [nvox], fa_index [nvox] and plan_ms (wall ms of every plan's thread).  `out`: a dict returned earlier,
    whose arrays are written again."""
    plans = list(plans) if isinstance(plans, (list, tuple)) else [plans]
    if not plans:
        raise ValueError("no plan")
    if method not in METHODS:
        raise ValueError("unknown reg_method %r" % (method,))
    nte, nt2 = plans[0].n_te, plans[0].n_t2
    on_device = hasattr(data, "data_ptr")
    data = _TensorView(data) if on_device else np.asarray(data)
    if data.dtype != np.float64 or data.ndim < 2 or data.shape[-1] != nte:
        raise ValueError("data must be a float64 array [..., n_te=%d], got %s %s" % (nte, data.dtype, data.shape))
    nvox = int(np.prod(data.shape[:-1]))
    if data.flags.c_contiguous:
        vs, es = nte, 1
    elif data.flags.f_contiguous:
        vs, es = 1, nvox
    elif data.ndim == 2 and data.strides[0] > 0 and data.strides[1] > 0 and data.strides[0] % 8 == 0 and data.strides[1] % 8 == 0:
        vs, es = data.strides[0] // 8, data.strides[1] // 8
    elif on_device:
        raise ValueError("a device volume must be C- or Fortran-contiguous, or a 2-D list with positive strides")
    else:
        data = np.ascontiguousarray(data)
        vs, es = nte, 1
    order = "F" if (es != 1 and data.ndim > 2) else "C"
    if estimate_fa not in FA_MODES:
        raise ValueError("estimate_fa: False, True / 'brute-force' or 'spline'")
    if fa_data is not None and hasattr(fa_data, "data_ptr"):
        fa_data = _TensorView(fa_data)
        if fa_data.shape != data.shape or fa_data.strides != data.strides:
            raise ValueError("a device fa_data must have the shape and memory layout of data")
    elif fa_data is not None:
        fa_data = np.asarray(fa_data)
        if fa_data.dtype != np.float64 or fa_data.shape != data.shape or fa_data.strides != data.strides:
            fa_data = np.asarray(fa_data, dtype=np.float64)
            if fa_data.shape != data.shape:
                raise ValueError("fa_data must have the shape of data")
            if not (data.flags.c_contiguous or data.flags.f_contiguous):
                # a strided view (say big[:, :n_te]): np.empty_like would hand back a COMPACT array, and the one pair of strides the C entry takes
                # would then be wrong for fa_data -- both arrays go in compact instead
                if on_device:
                    raise ValueError("a device fa_data must have the memory layout of data")
                data = np.ascontiguousarray(data)
                vs, es = nte, 1
                order = "C"
            lay = np.empty_like(data)            # same memory layout as data (C- or Fortran-contiguous here)
            lay[...] = fa_data
            assert lay.strides == data.strides
            fa_data = lay

    def per_voxel(a, dt, what):
        if a is None:
            return None
        a = np.asarray(a)
        if a.size != nvox:
            raise ValueError("%s must have one entry per voxel (%d), got %s" % (what, nvox, a.shape))
        if a.ndim > 1:
            a = a.reshape(-1, order=order)
        if dt == np.uint8:
            a = (a != 0)
        return np.ascontiguousarray(a, dtype=dt)

    fa = per_voxel(fa_index, np.float64, "fa_index")
    mk = per_voxel(mask, np.uint8, "mask")
    mv = per_voxel(mask_values, np.float64, "mask_values")
    o = out or {}

    def buf(name, shape, dt=np.float64, want=True):
        if not want:
            return None
        a = o.get(name)
        if a is not None and a.shape == tuple(shape) and a.dtype == dt and a.flags.c_contiguous and a.flags.writeable:
            return a
        return np.empty(shape, dtype=dt)

    res = {"fsol": buf("fsol", (nvox, nt2)), "sig": buf("sig", (nvox, nte), want=want_sig), "reg": buf("reg", (nvox,)),
           "lam": buf("lam", (nvox,), want=want_lambda), "maps": buf("maps", (6, nvox), want=want_maps),
           "status": buf("status", (nvox,), np.int32, want=want_status), "fa_index": buf("fa_index", (nvox,)),
           "fa_gate": buf("fa_gate", (nvox,), want=want_gate)}
    handles = (_vp * len(plans))(*[p._h for p in plans])
    ms = np.zeros(len(plans))
    check(lib().met2_fit_host(handles, len(plans), METHODS[method], nvox, _p(data), _p(fa_data), vs, es, _p(mv), _p(fa), _p(mk), FA_MODES[estimate_fa],
                              _p(res["fsol"]), _p(res["sig"]), _p(res["reg"]), _p(res["lam"]), _p(res["maps"]), _p(res["status"]),
                              _p(res["fa_index"]), _p(res["fa_gate"]), int(chunk), ms.ctypes.data_as(C.POINTER(C.c_double))))
    res["plan_ms"] = ms
    return res
