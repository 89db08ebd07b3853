"""Plan cache for the per-function drop-ins (nnls_x2(D, M, L, ...) style calls hand over the
dictionary and penalty on every call; the device copies are reused when they are unchanged)."""
import hashlib
from collections import OrderedDict

import numpy as np

from .plan import Met2Plan

_PLANS = OrderedDict()        # least recently used first
_MAX = 8


def _key(D3, L, lam_grid, device=0):
    h = hashlib.blake2b(digest_size=16)
    h.update(b"dev%d" % int(device))
    for a in (D3, L, lam_grid):
        if a is None:
            h.update(b"-")
        else:
            a = np.ascontiguousarray(a, dtype=np.float64)
            h.update(str(a.shape).encode())
            h.update(a.tobytes())
    return h.hexdigest()


def plan_for(Dic_3D, Laplac=None, lambda_reg=None, T2s=None, device=0):
    """Dic_3D in the reference layout [nTE, nT2, nFA] (a 2-D kernel is taken as nFA = 1); the plan lives on GPU `device`."""
    D3 = np.asarray(Dic_3D, dtype=np.float64)
    if D3.ndim == 2:
        D3 = D3[:, :, None]
    k = _key(D3, Laplac, lambda_reg, device)
    p = _PLANS.get(k)
    if p is not None:
        _PLANS.move_to_end(k)
    else:
        if len(_PLANS) >= _MAX:
            # evicted plans are dropped, not closed: a caller may still hold one (fitting_slice_FA_spline_method keeps the
            # coarse plan while it asks for the fine one); Met2Plan.__del__ frees the device memory with the last reference
            _PLANS.popitem(last=False)
        p = Met2Plan(D3.shape[0], D3.shape[1], D3.shape[2], device=int(device))
        p.set_dictionary(np.ascontiguousarray(D3))
        if Laplac is not None:
            p.set_penalty(np.asarray(Laplac, dtype=np.float64))
        if lambda_reg is not None:
            p.set_lambda_grid(lambda_reg)
        _PLANS[k] = p
    if T2s is not None:
        p.set_t2_grid(T2s)
    return p


def clear():
    for p in _PLANS.values():
        p.close()
    _PLANS.clear()
