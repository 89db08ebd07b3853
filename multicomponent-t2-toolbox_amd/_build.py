"""Build libmet2_hip.so (hipcc, gfx950) in-tree so that it travels with the source tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmet2_hip.so")
# one object per translation unit (compiled side by side, linked into ONE library): the solver kernels, and the TV denoiser
# (-DMET2_SPLIT_TU, the shipped build: the fit kernels are instantiated in met2_fit_*.hip, one family of methods per file -- fit_kernel.hpp;
#  with extra MET2_BUILD_DEFINES -- development builds -- met2_hip.hip instantiates them all and those files compile to nothing)
SOURCES = ["met2_hip.hip", "met2_fit_x2_nb1.hip", "met2_fit_x2_nb2.hip", "met2_fit_x2_second.hip", "met2_fit_nnls_lcurve.hip", "met2_fit_gcv.hip", "met2_fit_bayes.hip", "met2_tv.hip", "met2_host.hip"]
HEADERS = ["abi_common.hpp", "wave_ops.hpp", "nnls_wave.hpp", "nnls_big.hpp", "objectives.hpp", "fit_kernel.hpp", os.path.join("..", "..", "include", "met2_hip.h")]
STAMP = LIB + ".flags"          # extra compile flags the library was built with (MET2_BUILD_DEFINES, e.g. -DMET2_CYCSTATS)
# Machine LICM off: it hoists the materialisation of fp64 literals (erf/log coefficients of the BayesReg objective, 20 register
# pairs) and per-lane address constants out of the voxel loop, runs out of registers and spills them to scratch -- BayesReg at
# 32 x 60 reloaded 110 KB per voxel from scratch.  Without it: 59 -> 2 spilled VGPRs there, configs[3] 2.77 -> 2.95 M voxels/s,
# configs[2] +2.7 %, configs[1] unchanged, X2 at 48 x 120 -2 %.
CODEGEN = ["-mllvm", "-disable-machine-licm"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built on this machine")
    return exe


def extra_flags():
    return os.environ.get("MET2_BUILD_DEFINES", "").split()


def stale():
    if not os.path.exists(LIB):
        return True
    built_with = open(STAMP).read().split() if os.path.exists(STAMP) else None
    if built_with != extra_flags():
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _object(src):
    return os.path.join(CSRC, os.path.splitext(src)[0] + ".o")


def _object_stale(src, flags):
    obj = _object(src)
    if not os.path.exists(obj) or not os.path.exists(obj + ".flags") or open(obj + ".flags").read().split() != flags:
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    flags = extra_flags()
    split = [] if flags else ["-DMET2_SPLIT_TU"]
    base = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + CODEGEN + split + flags
    jobs = []
    for src in SOURCES:
        if force or _object_stale(src, flags):
            cmd = base + ["-c", os.path.join(CSRC, src), "-o", _object(src)]
            if verbose:
                print(" ".join(cmd))
            jobs.append((src, cmd, subprocess.Popen(cmd)))
    for src, cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        with open(_object(src) + ".flags", "w") as f:
            f.write(" ".join(flags))
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_object(s) for s in SOURCES]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    with open(STAMP, "w") as f:
        f.write(" ".join(flags))
    return LIB
