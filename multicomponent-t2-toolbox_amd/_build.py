"""Build libmet2_hip.so (hipcc, gfx950) in-tree so that it travels with the source tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmet2_hip.so")
SOURCES = ["met2_hip.hip"]
HEADERS = ["wave_ops.hpp", "nnls_wave.hpp", "objectives.hpp", os.path.join("..", "..", "include", "met2_hip.h")]
STAMP = LIB + ".flags"          # extra compile flags the library was built with (MET2_BUILD_DEFINES, e.g. -DMET2_CYCSTATS)


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
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + extra_flags() + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(" ".join(extra_flags()))
    return LIB
