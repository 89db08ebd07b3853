"""Met2Plan: Python face of the C-ABI plan object.  torch is used only for device memory and
streams; every numeric step runs in libmet2_hip.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Options, check, lib

METHODS = {"NNLS": 0, "T2SPARC": 1, "X2": 2, "L_curve": 3, "GCV": 4, "BayesReg": 5}
PENALTIES = {"I": 0, "L1": 1, "L2": 2, "InvT2": 3}
MAP_NAMES = ("MWF", "IEWF", "FWF", "T2_M", "T2_IE", "TWC")

_dp = C.POINTER(C.c_double)


def _h(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def voxel_layout(data, n_te):
    """(tensor, nvox, voxel_stride, echo_stride, vol_shape) for the strided C entries: `data` is [..., n_te] (any number of
    leading voxel axes).  C-contiguous input -> strides (n_te, 1), voxels in C order.  Fortran-contiguous input (what nibabel
    hands the reference's driver, motor:167-173) -> strides (1, nvox), voxels in Fortran order: no copy, outputs come back in
    that voxel order (`unflatten` puts them into the volume's shape).  A 2-D voxel list with any positive strides (a slice of a larger
    list) is read in place as well.  Any other layout is made C-contiguous (one copy)."""
    if data.dim() < 2 or data.shape[-1] != n_te:
        raise ValueError("data must be [..., n_te=%d], got %s" % (n_te, tuple(data.shape)))
    vol = tuple(data.shape[:-1])
    nvox = 1
    for d in vol:
        nvox *= int(d)
    if data.is_contiguous():
        return data, nvox, n_te, 1, vol, "C"
    if data.permute(*reversed(range(data.dim()))).is_contiguous():       # (also an echo-major voxel list: a transposed [n_te, nvox] array)
        return data, nvox, 1, nvox, vol, "F"
    if data.dim() == 2 and data.stride(0) > 0 and data.stride(1) > 0:    # a run of voxels cut out of a larger list (either layout): the C entries take any strides
        return data, nvox, int(data.stride(0)), int(data.stride(1)), vol, "C"
    return data.contiguous(), nvox, n_te, 1, vol, "C"


def unflatten_back(t, order):
    """[vol..., k] output of fit() -> flat [nvox, k] in the data's voxel order (a view for order 'F' volumes too)."""
    if order == "C" or t.dim() <= 2:
        return t.reshape(-1, t.shape[-1])
    nd = t.dim() - 1
    return t.permute(*(list(reversed(range(nd))) + [nd])).reshape(-1, t.shape[-1])


def unflatten(t, vol, order, lead=0):
    """View a per-voxel output ([nvox, ...] or, with lead=1, [k, nvox]) with the voxel axis unfolded to `vol`."""
    if order == "C":
        return t.reshape(t.shape[:lead] + tuple(vol) + t.shape[lead + 1:])
    rv = tuple(reversed(vol))
    u = t.reshape(t.shape[:lead] + rv + t.shape[lead + 1:])
    nd = len(vol)
    perm = list(range(lead)) + [lead + nd - 1 - i for i in range(nd)] + list(range(lead + nd, u.dim()))
    return u.permute(*perm)


class Met2Plan:
    """Shared nTE x nT2 x nFA problem on one GPU: dictionary, Gram matrices, penalty, lambda grid."""

    def __init__(self, n_te, n_t2, n_fa, device=0, x2_factor=1.02, t2sparc_lambda=1.8, myelin_T2=40.0, brent_maxfun=0):
        if not torch.cuda.is_available():
            raise _lib.Met2Error("no GPU visible: the MI355X path has no CPU fallback")
        self.n_te, self.n_t2, self.n_fa = int(n_te), int(n_t2), int(n_fa)
        self.device = torch.device("cuda", int(device))
        opt = Options()
        lib().met2_default_options(C.byref(opt))
        opt.device = int(device)
        opt.x2_factor = x2_factor
        opt.t2sparc_lambda = t2sparc_lambda
        opt.t2_myelin_cut = myelin_T2
        opt.brent_maxfun = brent_maxfun
        self._h = C.c_void_p(0)
        check(lib().met2_plan_create(C.byref(self._h), self.n_te, self.n_t2, self.n_fa, C.byref(opt)))
        # the reference's L-curve grid, bit for bit as numpy builds it (motor:248-251); the C default is the same
        # grid through pow() and can differ in the last bit
        lam = np.zeros(50)
        lam[1:] = np.logspace(np.log10(1e-8), np.log10(10.0), num=49, endpoint=True, base=10.0)
        self._lam_grid = None
        self.set_lambda_grid(lam)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().met2_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- configuration
    def set_options(self, **kw):
        """x2_factor, t2sparc_lambda, brent_xtol, brent_maxfun, t2_myelin_cut, t2_ie_cut, and the lambda-search intervals x2_lo / x2_hi,
        gcv_lo / gcv_hi, bayes_lo / bayes_hi (scipy's fminbound bounds at algorithms.py:219, :280, bayesian_interpolation.py:101)"""
        opt = Options()
        check(lib().met2_plan_get_options(self._h, C.byref(opt)))
        for k, v in kw.items():
            if not hasattr(opt, k):
                raise AttributeError(k)
            setattr(opt, k, v)
        check(lib().met2_plan_set_options(self._h, C.byref(opt)))
        return self

    def get_options(self, *names):
        opt = Options()
        check(lib().met2_plan_get_options(self._h, C.byref(opt)))
        return {k: getattr(opt, k) for k in (names or ("x2_factor", "t2sparc_lambda", "brent_xtol", "brent_maxfun", "t2_myelin_cut", "t2_ie_cut",
                                                           "x2_lo", "x2_hi", "gcv_lo", "gcv_hi", "bayes_lo", "bayes_hi"))}

    def build_dictionary_epg(self, T2s, T1s, tau, alpha_values, TR):
        (_, p2), (_, p1), (_, pa) = _h(T2s), _h(T1s), _h(alpha_values)
        keep = (_h(T2s), _h(T1s), _h(alpha_values))
        check(lib().met2_plan_build_dictionary_epg(self._h, keep[0][1], keep[1][1], float(tau), keep[2][1], float(TR), self._stream()))
        return self

    def set_dictionary(self, Dic_3D):
        a, p = _h(Dic_3D)
        if a.shape != (self.n_te, self.n_t2, self.n_fa):
            raise ValueError("Dic_3D must be [n_te, n_t2, n_fa] = %s, got %s" % ((self.n_te, self.n_t2, self.n_fa), a.shape))
        check(lib().met2_plan_set_dictionary(self._h, p))
        return self

    def get_dictionary(self):
        out = np.zeros((self.n_te, self.n_t2, self.n_fa))
        check(lib().met2_plan_get_dictionary(self._h, out.ctypes.data_as(_dp)))
        return out

    def set_penalty(self, penalty, T2s=None):
        if isinstance(penalty, str):
            a, p = _h(T2s if T2s is not None else np.zeros(self.n_t2))
            check(lib().met2_plan_set_penalty(self._h, PENALTIES[penalty], p))
        else:
            a, p = _h(penalty)
            if a.shape != (self.n_t2, self.n_t2):
                raise ValueError("penalty matrix must be [n_t2, n_t2] = %s, got %s" % ((self.n_t2, self.n_t2), a.shape))
            check(lib().met2_plan_set_penalty_dense(self._h, p))
        return self

    def get_penalty(self):
        out = np.zeros((self.n_t2, self.n_t2))
        check(lib().met2_plan_get_penalty(self._h, out.ctypes.data_as(_dp)))
        return out

    def set_lambda_grid(self, lambda_reg):
        a, p = _h(lambda_reg)
        check(lib().met2_plan_set_lambda_grid(self._h, p, a.shape[0]))
        self._lam_grid = a.copy()
        return self

    def set_t2_grid(self, T2s):
        a, p = _h(T2s)
        check(lib().met2_plan_set_t2_grid(self._h, p))
        return self

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- hot path (device tensors in, device tensors out)
    def _check_data(self, data, what="data"):
        if not torch.is_tensor(data) or not data.is_cuda:
            raise ValueError("%s must be a CUDA tensor (the hot path has no host fallback)" % what)
        if data.device != self.device:
            raise ValueError("%s lives on %s but the plan on %s" % (what, data.device, self.device))
        if data.dtype != torch.float64:
            raise ValueError("%s must be float64, got %s" % (what, data.dtype))

    def _per_voxel(self, t, nvox, dtype, what, order="C", vol=None):
        """fa_index / mask arguments -> flat [nvox] tensor in the voxel order of the data layout."""
        if t is None:
            return None
        t = torch.as_tensor(t, device=self.device)
        if t.numel() != nvox:
            raise ValueError("%s has %d entries for %d voxels" % (what, t.numel(), nvox))
        if order == "F" and t.dim() > 1:
            t = t.permute(*reversed(range(t.dim())))
        t = t.reshape(-1)
        return (t != 0).to(torch.uint8).contiguous() if dtype == torch.uint8 else t.to(dtype).contiguous()

    def fit(self, method, data, fa_index=None, mask=None, want_sig=True, want_maps=True, want_status=True, want_lambda=False,
            out=None, sync=True):
        """data [..., n_te] float64 cuda tensor: a voxel list [nvox, n_te] or a volume [nx, ny, nz, n_te], C- or
        Fortran-contiguous (see voxel_layout; neither is copied).  fa_index / mask: one entry per voxel, flat in the data's
        voxel order or shaped like the volume.  Returns a dict of cuda tensors with the voxel axes of `data`.
        sync=False only enqueues the launches on the current stream (met2_fit_enqueue_strided): call finish() before the
        outputs are read on the host; an FA index outside the dictionary is then reported by finish()."""
        if method not in METHODS:
            raise ValueError("unknown reg_method %r" % (method,))
        self._check_data(data)
        data, nvox, vs, es, vol, order = voxel_layout(data, self.n_te)
        dev = self.device
        fa_index = self._per_voxel(fa_index, nvox, torch.float64, "fa_index", order)
        mask = self._per_voxel(mask, nvox, torch.uint8, "mask", order)
        o = out or {}

        def buf(name, shape, dtype=torch.float64):
            t = o.get(name)
            if t is None:
                return torch.empty(shape, dtype=dtype, device=dev)
            if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != dev or not t.is_contiguous():
                raise ValueError("out[%r] must be a contiguous %s tensor of shape %s on %s" % (name, dtype, tuple(shape), dev))
            return t

        fsol = buf("fsol", (nvox, self.n_t2))
        sig = buf("sig", (nvox, self.n_te)) if want_sig else None
        reg = buf("reg", (nvox,))
        lam = buf("lam", (nvox,)) if want_lambda else None
        maps = buf("maps", (6, nvox)) if want_maps else None
        status = buf("status", (nvox,), torch.int32) if want_status else None
        entry = lib().met2_fit_strided if sync else lib().met2_fit_enqueue_strided
        with torch.cuda.device(dev):
            check(entry(self._h, METHODS[method], nvox, _ptr(data), vs, es, _ptr(fa_index), _ptr(mask), _ptr(fsol),
                        _ptr(sig), _ptr(reg), _ptr(lam), _ptr(maps), _ptr(status), self._stream()))
        res = {"fsol": fsol, "sig": sig, "reg": reg, "lam": lam, "maps": maps, "status": status}
        if len(vol) > 1:
            res = {k: (None if t is None else unflatten(t, vol, order, lead=1 if k == "maps" else 0)) for k, t in res.items()}
        return res

    def finish(self):
        """Wait for the current stream and report what fits enqueued with sync=False deferred (met2_plan_finish)."""
        with torch.cuda.device(self.device):
            check(lib().met2_plan_finish(self._h, self._stream()))
        return self

    def objective_grid(self, method, data, lams, fa_index=None):
        """Values of `method`'s lambda-selection objective at `lams` for every voxel -> [nvox, len(lams)].
        The plan's own lambda grid (the L-curve grid) is put back afterwards."""
        lams = np.ascontiguousarray(lams, dtype=np.float64)
        if not 3 <= lams.shape[0] <= min(self.n_t2, 64):
            raise ValueError("objective grid needs 3..min(n_t2, 64) points")
        self._check_data(data)
        data = data.contiguous()
        nvox = data.shape[0]
        fsol = torch.empty((nvox, self.n_t2), dtype=torch.float64, device=data.device)
        reg = torch.empty((nvox,), dtype=torch.float64, device=data.device)
        fa = self._per_voxel(fa_index, nvox, torch.float64, "fa_index")
        saved = self._lam_grid
        self.set_lambda_grid(lams)
        try:
            with torch.cuda.device(data.device):
                check(lib().met2_fit(self._h, 10 + METHODS[method], nvox, _ptr(data), _ptr(fa), _ptr(None), _ptr(fsol), _ptr(None),
                                     _ptr(reg), _ptr(None), _ptr(None), _ptr(None), self._stream()))
                torch.cuda.current_stream(data.device).synchronize()
        finally:
            if saved is not None:
                self.set_lambda_grid(saved)
        return fsol[:, : lams.shape[0]]

    def fa_bruteforce(self, data, mask=None, want_resid=False):
        """fa_estimation.py:74-111 over the plan's flip angles.  data as in fit(); returns flat per-voxel tensors
        (fa_index, km, resid [nvox, n_fa] or None) in the data's voxel order."""
        self._check_data(data)
        data, nvox, vs, es, vol, order = voxel_layout(data, self.n_te)
        dev = self.device
        mask = self._per_voxel(mask, nvox, torch.uint8, "mask", order)
        fa = torch.empty((nvox,), dtype=torch.float64, device=dev)
        km = torch.empty((nvox,), dtype=torch.float64, device=dev)
        resid = torch.empty((nvox, self.n_fa), dtype=torch.float64, device=dev) if want_resid else None
        with torch.cuda.device(dev):
            check(lib().met2_fa_bruteforce_strided(self._h, nvox, _ptr(data), vs, es, _ptr(mask), _ptr(fa), _ptr(km), _ptr(resid), self._stream()))
        return fa, km, resid

    def fa_spline(self, plan_lr, alpha_lr, alpha_hr, data, mask=None, want_xmin=False, want_km=True):
        """Spline FA method (fa_estimation.py:35-70): `plan_lr` holds the coarse-grid dictionary (15 flip angles in the
        driver, motor:237-238), this plan the fine one (273).  Returns (fa_index into alpha_hr, km, xmin or None), flat in
        the data's voxel order."""
        self._check_data(data)
        if plan_lr.device != self.device or plan_lr.n_te != self.n_te:
            raise ValueError("the coarse plan must live on the same device and have the same n_te")
        data, nvox, vs, es, vol, order = voxel_layout(data, self.n_te)
        dev = self.device
        mk = self._per_voxel(mask, nvox, torch.uint8, "mask", order)
        _, _, resid = plan_lr.fa_bruteforce(data, None if mk is None else unflatten(mk, vol, order), want_resid=True)
        (al, pal), (ah, pah) = _h(alpha_lr), _h(alpha_hr)
        fa = torch.empty((nvox,), dtype=torch.float64, device=dev)
        xmin = torch.empty((nvox,), dtype=torch.float64, device=dev) if want_xmin else None
        with torch.cuda.device(dev):
            check(lib().met2_fa_spline_select_strided(dev.index or 0, nvox, al.shape[0], pal, _ptr(resid), ah.shape[0], pah, self.n_te,
                                                      _ptr(data), vs, es, _ptr(mk), _ptr(fa), _ptr(xmin), self._stream()))
        if not want_km:                       # the volume driver never uses it (its Ktotal comes from the final spectra, motor:455-468)
            return fa, None, xmin
        # km = sum of the plain-NNLS spectrum at the selected flip angle (fa_estimation.py:61-64)
        tot = data.sum(dim=-1)
        tot = tot.reshape(-1) if order == "C" else tot.permute(*reversed(range(tot.dim()))).reshape(-1)
        gate = (tot > 0) if mk is None else ((tot > 0) & (mk != 0))
        out = self.fit("NNLS", data, fa_index=fa, mask=gate, want_sig=False, want_maps=False, want_status=False)
        return fa, unflatten_back(out["fsol"], order).sum(dim=1), xmin

    def metrics(self, fsol, mask=None):
        fsol = fsol.contiguous()
        nvox = fsol.shape[0]
        if mask is not None:
            mask = (mask != 0).to(device=fsol.device, dtype=torch.uint8).contiguous()
        maps = torch.empty((6, nvox), dtype=torch.float64, device=fsol.device)
        with torch.cuda.device(fsol.device):
            check(lib().met2_metrics(self._h, nvox, _ptr(fsol), _ptr(mask), _ptr(maps), self._stream()))
        return maps

    def last_kernel_ms(self):
        ms = C.c_double(0.0)
        check(lib().met2_plan_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_second_pass_ms(self):
        ms = C.c_double(0.0)
        check(lib().met2_plan_last_second_pass_ms(self._h, C.byref(ms)))
        return ms.value

    def last_spill_count(self):
        """Voxels of the most recent fit whose passive set outgrew the wave's LDS region and went on in the spill-over slot."""
        n = C.c_int64(0)
        check(lib().met2_plan_last_spill_count(self._h, C.byref(n)))
        return n.value

    def gcv_form(self):
        """(low_rank, residual): whether GCV's trace is taken from the 17 x 17 form in the dictionary's low-rank basis on this plan."""
        lr, res = C.c_int32(0), C.c_double(0.0)
        check(lib().met2_plan_gcv_form(self._h, C.byref(lr), C.byref(res)))
        return bool(lr.value), res.value

    def launch_info(self, method="X2"):
        g, b, l = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(lib().met2_plan_launch_info(self._h, METHODS[method], C.byref(g), C.byref(b), C.byref(l)))
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value}
