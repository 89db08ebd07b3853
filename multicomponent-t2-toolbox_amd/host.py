"""fit_host: numpy in, numpy out, one or several GPUs from ONE process -- the Python face of met2_fit_host (include/met2_hip.h, ABI 5).

The reference's driver holds its volume in host memory (motor:167-182) and loops over image rows in one process (motor:427-441);
this is that loop handed to the library: blocks of voxels are dealt to the plans (one per device, each driven by its own host thread
inside the C call), copied in, fitted and copied out on three streams per device.  No torch here: the arrays are the caller's numpy
arrays, staged through pinned block buffers inside the library (arrays that already live in pinned memory are used in place)."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .plan import METHODS

_vp = C.c_void_p


def _p(a):
    return _vp(a.ctypes.data) if a is not None else _vp(0)


def fit_host(plans, method, data, fa_index=None, mask=None, estimate_fa=False, chunk=0, want_sig=True, want_maps=True, want_status=True,
             want_lambda=False, out=None):
    """plans: a Met2Plan or a sequence of them (same shape, configured alike; one per device for a multi-GPU run).
    data: float64 numpy array [..., n_te] -- a voxel list or a volume, C-ordered or the Fortran-ordered array nibabel delivers (read as
    it lies); any other layout is gathered inside the library.  fa_index / mask: per voxel, in the memory order of `data`'s voxels.
    estimate_fa=True: brute-force FA search over the plans' FA axis per block (fa_estimation.py:74-111).
    Returns numpy arrays flat in the memory order of the voxels: fsol [nvox, n_t2], sig [nvox, n_te], reg [nvox], maps [6, nvox],
    status [nvox] int32, lam [nvox], fa_index [nvox] and plan_ms (wall ms of every plan's thread).  `out`: a dict returned earlier,
    whose arrays are written again."""
    plans = list(plans) if isinstance(plans, (list, tuple)) else [plans]
    if not plans:
        raise ValueError("no plan")
    if method not in METHODS:
        raise ValueError("unknown reg_method %r" % (method,))
    nte, nt2 = plans[0].n_te, plans[0].n_t2
    data = np.asarray(data)
    if data.dtype != np.float64 or data.ndim < 2 or data.shape[-1] != nte:
        raise ValueError("data must be a float64 array [..., n_te=%d], got %s %s" % (nte, data.dtype, data.shape))
    nvox = int(np.prod(data.shape[:-1]))
    if data.flags.c_contiguous:
        vs, es = nte, 1
    elif data.flags.f_contiguous:
        vs, es = 1, nvox
    elif data.ndim == 2 and data.strides[0] > 0 and data.strides[1] > 0 and data.strides[0] % 8 == 0 and data.strides[1] % 8 == 0:
        vs, es = data.strides[0] // 8, data.strides[1] // 8
    else:
        data = np.ascontiguousarray(data)
        vs, es = nte, 1
    order = "F" if (es != 1 and data.ndim > 2) else "C"

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
           "status": buf("status", (nvox,), np.int32, want=want_status), "fa_index": buf("fa_index", (nvox,))}
    handles = (_vp * len(plans))(*[p._h for p in plans])
    ms = np.zeros(len(plans))
    check(lib().met2_fit_host(handles, len(plans), METHODS[method], nvox, _p(data), vs, es, _p(fa), _p(mk), 1 if estimate_fa else 0,
                              _p(res["fsol"]), _p(res["sig"]), _p(res["reg"]), _p(res["lam"]), _p(res["maps"]), _p(res["status"]),
                              _p(res["fa_index"]), int(chunk), ms.ctypes.data_as(C.POINTER(C.c_double))))
    res["plan_ms"] = ms
    return res
