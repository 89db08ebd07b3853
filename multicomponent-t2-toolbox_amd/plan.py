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
        """x2_factor, t2sparc_lambda, brent_xtol, brent_maxfun, t2_myelin_cut, t2_ie_cut"""
        opt = Options()
        check(lib().met2_plan_get_options(self._h, C.byref(opt)))
        for k, v in kw.items():
            if not hasattr(opt, k):
                raise AttributeError(k)
            setattr(opt, k, v)
        check(lib().met2_plan_set_options(self._h, C.byref(opt)))
        return self

    def build_dictionary_epg(self, T2s, T1s, tau, alpha_values, TR):
        (_, p2), (_, p1), (_, pa) = _h(T2s), _h(T1s), _h(alpha_values)
        keep = (_h(T2s), _h(T1s), _h(alpha_values))
        check(lib().met2_plan_build_dictionary_epg(self._h, keep[0][1], keep[1][1], float(tau), keep[2][1], float(TR), self._stream()))
        return self

    def set_dictionary(self, Dic_3D):
        a, p = _h(Dic_3D)
        assert a.shape == (self.n_te, self.n_t2, self.n_fa), a.shape
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
            assert a.shape == (self.n_t2, self.n_t2)
            check(lib().met2_plan_set_penalty_dense(self._h, p))
        return self

    def get_penalty(self):
        out = np.zeros((self.n_t2, self.n_t2))
        check(lib().met2_plan_get_penalty(self._h, out.ctypes.data_as(_dp)))
        return out

    def set_lambda_grid(self, lambda_reg):
        a, p = _h(lambda_reg)
        check(lib().met2_plan_set_lambda_grid(self._h, p, a.shape[0]))
        return self

    def set_t2_grid(self, T2s):
        a, p = _h(T2s)
        check(lib().met2_plan_set_t2_grid(self._h, p))
        return self

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- hot path (device tensors in, device tensors out)
    def fit(self, method, data, fa_index=None, mask=None, want_sig=True, want_maps=True, want_status=True, want_lambda=False,
            out=None):
        """data [nvox, n_te] float64 cuda tensor.  Returns dict of cuda tensors."""
        assert data.is_cuda and data.dtype == torch.float64 and data.dim() == 2 and data.shape[1] == self.n_te
        data = data.contiguous()
        nvox = data.shape[0]
        dev = data.device
        if fa_index is not None:
            fa_index = fa_index.to(device=dev, dtype=torch.float64).contiguous()
        if mask is not None:
            mask = (mask != 0).to(device=dev, dtype=torch.uint8).contiguous()
        o = out or {}
        fsol = o.get("fsol") if "fsol" in o else torch.empty((nvox, self.n_t2), dtype=torch.float64, device=dev)
        sig = (o.get("sig") if "sig" in o else torch.empty((nvox, self.n_te), dtype=torch.float64, device=dev)) if want_sig else None
        reg = o.get("reg") if "reg" in o else torch.empty((nvox,), dtype=torch.float64, device=dev)
        lam = (o.get("lam") if "lam" in o else torch.empty((nvox,), dtype=torch.float64, device=dev)) if want_lambda else None
        maps = (o.get("maps") if "maps" in o else torch.empty((6, nvox), dtype=torch.float64, device=dev)) if want_maps else None
        status = (o.get("status") if "status" in o else torch.empty((nvox,), dtype=torch.int32, device=dev)) if want_status else None
        with torch.cuda.device(dev):
            check(lib().met2_fit(self._h, METHODS[method], nvox, _ptr(data), _ptr(fa_index), _ptr(mask), _ptr(fsol), _ptr(sig),
                                 _ptr(reg), _ptr(lam), _ptr(maps), _ptr(status), self._stream()))
        return {"fsol": fsol, "sig": sig, "reg": reg, "lam": lam, "maps": maps, "status": status}

    def objective_grid(self, method, data, lams, fa_index=None):
        """Values of `method`'s lambda-selection objective at `lams` for every voxel -> [nvox, len(lams)]."""
        lams = np.ascontiguousarray(lams, dtype=np.float64)
        assert 3 <= lams.shape[0] <= self.n_t2
        self.set_lambda_grid(lams)
        data = data.contiguous()
        nvox = data.shape[0]
        fsol = torch.empty((nvox, self.n_t2), dtype=torch.float64, device=data.device)
        reg = torch.empty((nvox,), dtype=torch.float64, device=data.device)
        fa = None if fa_index is None else fa_index.to(device=data.device, dtype=torch.float64).contiguous()
        with torch.cuda.device(data.device):
            check(lib().met2_fit(self._h, 10 + METHODS[method], nvox, _ptr(data), _ptr(fa), _ptr(None), _ptr(fsol), _ptr(None),
                                 _ptr(reg), _ptr(None), _ptr(None), _ptr(None), self._stream()))
        return fsol[:, : lams.shape[0]]

    def fa_bruteforce(self, data, mask=None, want_resid=False):
        assert data.is_cuda and data.dtype == torch.float64 and data.shape[1] == self.n_te
        data = data.contiguous()
        nvox = data.shape[0]
        dev = data.device
        if mask is not None:
            mask = (mask != 0).to(device=dev, dtype=torch.uint8).contiguous()
        fa = torch.empty((nvox,), dtype=torch.float64, device=dev)
        km = torch.empty((nvox,), dtype=torch.float64, device=dev)
        resid = torch.empty((nvox, self.n_fa), dtype=torch.float64, device=dev) if want_resid else None
        with torch.cuda.device(dev):
            check(lib().met2_fa_bruteforce(self._h, nvox, _ptr(data), _ptr(mask), _ptr(fa), _ptr(km), _ptr(resid), self._stream()))
        return fa, km, resid

    def fa_spline(self, plan_lr, alpha_lr, alpha_hr, data, mask=None, want_xmin=False, want_km=True):
        """Spline FA method (fa_estimation.py:35-70): `plan_lr` holds the coarse-grid dictionary (15 flip angles in the
        driver, motor:237-238), this plan the fine one (273).  Returns (fa_index into alpha_hr, km, xmin or None)."""
        assert data.is_cuda and data.dtype == torch.float64 and data.shape[1] == self.n_te
        data = data.contiguous()
        nvox = data.shape[0]
        dev = data.device
        mk = None if mask is None else (mask != 0).to(device=dev, dtype=torch.uint8).contiguous()
        _, _, resid = plan_lr.fa_bruteforce(data, mk, want_resid=True)
        (al, pal), (ah, pah) = _h(alpha_lr), _h(alpha_hr)
        fa = torch.empty((nvox,), dtype=torch.float64, device=dev)
        xmin = torch.empty((nvox,), dtype=torch.float64, device=dev) if want_xmin else None
        with torch.cuda.device(dev):
            check(lib().met2_fa_spline_select(dev.index or 0, nvox, al.shape[0], pal, _ptr(resid), ah.shape[0], pah, self.n_te, _ptr(data),
                                              _ptr(mk), _ptr(fa), _ptr(xmin), self._stream()))
        if not want_km:                       # the volume driver never uses it (its Ktotal comes from the final spectra, motor:455-468)
            return fa, None, xmin
        # km = sum of the plain-NNLS spectrum at the selected flip angle (fa_estimation.py:61-64)
        gate = (data.sum(dim=1) > 0) if mk is None else ((data.sum(dim=1) > 0) & (mk != 0))
        out = self.fit("NNLS", data, fa_index=fa, mask=gate, want_sig=False, want_maps=False, want_status=False)
        return fa, out["fsol"].sum(dim=1), xmin

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

    def launch_info(self, method="X2"):
        g, b, l = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(lib().met2_plan_launch_info(self._h, METHODS[method], C.byref(g), C.byref(b), C.byref(l)))
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value}
