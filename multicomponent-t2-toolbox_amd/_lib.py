"""ctypes binding of include/met2_hip.h.  There is no CPU fallback: if the HIP library is
missing or no MI355X is visible, every product entry point raises."""
import ctypes as C
import os

from . import _build

_LIB = None

_dp = C.POINTER(C.c_double)


class Met2Error(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("x2_factor", C.c_double), ("t2sparc_lambda", C.c_double),
                ("brent_xtol", C.c_double), ("brent_maxfun", C.c_int32), ("reserved0", C.c_int32), ("t2_myelin_cut", C.c_double),
                ("t2_ie_cut", C.c_double), ("x2_lo", C.c_double), ("x2_hi", C.c_double), ("gcv_lo", C.c_double), ("gcv_hi", C.c_double),
                ("bayes_lo", C.c_double), ("bayes_hi", C.c_double)]


# every symbol include/met2_hip.h declares
SYMBOLS = ["met2_default_options", "met2_abi_version", "met2_device_count", "met2_last_error", "met2_plan_create",
           "met2_plan_destroy", "met2_plan_set_options", "met2_plan_get_options", "met2_plan_build_dictionary_epg", "met2_plan_set_dictionary", "met2_plan_get_dictionary",
           "met2_plan_set_penalty", "met2_plan_set_penalty_dense", "met2_plan_get_penalty", "met2_plan_set_lambda_grid",
           "met2_plan_set_t2_grid", "met2_fit", "met2_fit_strided", "met2_fit_enqueue_strided", "met2_plan_finish", "met2_fa_bruteforce", "met2_fa_bruteforce_strided", "met2_fa_spline_select",
           "met2_fa_spline_select_strided", "met2_roi_reduce", "met2_nesma", "met2_tv_work_bytes", "met2_tv_chambolle", "met2_tv_last_timing", "met2_smooth_separable", "met2_metrics", "met2_plan_last_kernel_ms", "met2_plan_last_second_pass_ms", "met2_plan_last_spill_count",
           "met2_plan_launch_info", "met2_plan_gcv_form", "met2_plan_get_shape", "met2_fit_host", "met2_plan_attach_fa_spline", "met2_host_trim"]


def lib():
    global _LIB
    if _LIB is None:
        path = _build.LIB
        if not os.path.exists(path):
            raise Met2Error("HIP extension %s is not built (run __graft_entry__.build()); there is no CPU fallback" % path)
        L = C.CDLL(path)
        L.met2_last_error.restype = C.c_char_p
        vp = C.c_void_p
        L.met2_plan_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32, C.POINTER(Options)]
        L.met2_plan_destroy.argtypes = [vp]
        L.met2_plan_set_options.argtypes = [vp, C.POINTER(Options)]
        L.met2_plan_get_options.argtypes = [vp, C.POINTER(Options)]
        L.met2_plan_build_dictionary_epg.argtypes = [vp, _dp, _dp, C.c_double, _dp, C.c_double, vp]
        L.met2_plan_set_dictionary.argtypes = [vp, _dp]
        L.met2_plan_get_dictionary.argtypes = [vp, _dp]
        L.met2_plan_set_penalty.argtypes = [vp, C.c_int32, _dp]
        L.met2_plan_set_penalty_dense.argtypes = [vp, _dp]
        L.met2_plan_get_penalty.argtypes = [vp, _dp]
        L.met2_plan_set_lambda_grid.argtypes = [vp, _dp, C.c_int32]
        L.met2_plan_set_t2_grid.argtypes = [vp, _dp]
        L.met2_fit.argtypes = [vp, C.c_int32, C.c_int64] + [vp] * 10
        L.met2_fit_strided.argtypes = [vp, C.c_int32, C.c_int64, vp, C.c_int64, C.c_int64] + [vp] * 9
        L.met2_fit_enqueue_strided.argtypes = [vp, C.c_int32, C.c_int64, vp, C.c_int64, C.c_int64] + [vp] * 9
        L.met2_plan_finish.argtypes = [vp, vp]
        L.met2_fa_bruteforce.argtypes = [vp, C.c_int64] + [vp] * 6
        L.met2_fa_bruteforce_strided.argtypes = [vp, C.c_int64, vp, C.c_int64, C.c_int64] + [vp] * 5
        L.met2_fa_spline_select_strided.argtypes = [C.c_int32, C.c_int64, C.c_int32, _dp, vp, C.c_int32, _dp, C.c_int32, vp, C.c_int64, C.c_int64,
                                                    vp, vp, vp, vp]
        L.met2_roi_reduce.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.c_int64, vp, vp, vp, vp, vp]
        L.met2_fa_spline_select.argtypes = [C.c_int32, C.c_int64, C.c_int32, _dp, vp, C.c_int32, _dp, C.c_int32, vp, vp, vp, vp, vp]
        L.met2_nesma.argtypes = [C.c_int32] * 5 + [vp] * 4
        L.met2_tv_work_bytes.argtypes = [C.c_int32] * 5
        L.met2_tv_work_bytes.restype = C.c_int64
        L.met2_tv_chambolle.argtypes = [C.c_int32] * 5 + [vp, C.c_int32, _dp, C.c_double, C.c_double, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_int64, vp]
        L.met2_tv_last_timing.argtypes = [_dp, C.POINTER(C.c_int32)]
        L.met2_smooth_separable.argtypes = [C.c_int32] * 6 + [_dp] + [vp] * 4
        L.met2_metrics.argtypes = [vp, C.c_int64, vp, vp, vp, vp]
        L.met2_plan_last_kernel_ms.argtypes = [vp, _dp]
        L.met2_plan_last_second_pass_ms.argtypes = [vp, _dp]
        L.met2_plan_last_spill_count.argtypes = [vp, C.POINTER(C.c_int64)]
        L.met2_plan_gcv_form.argtypes = [vp, C.POINTER(C.c_int32), _dp]
        L.met2_plan_launch_info.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.met2_plan_get_shape.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.met2_fit_host.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int64, vp, vp, C.c_int64, C.c_int64, vp, vp, vp, C.c_int32] + [vp] * 8 + [C.c_int64, _dp]
        L.met2_host_trim.argtypes = []
        L.met2_plan_attach_fa_spline.argtypes = [vp, vp, C.c_int32, _dp, C.c_int32, _dp]
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        msg = lib().met2_last_error()
        raise Met2Error("met2_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
