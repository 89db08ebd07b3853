// abi_common.hpp -- error plumbing shared by the translation units of libmet2_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace met2 {
// records `msg` as the calling thread's last error (met2_last_error) and returns `code`; defined in met2_hip.hip
__attribute__((visibility("hidden"))) int abi_fail(int code, const std::string &msg);
}  // namespace met2

static inline int fail(int code, const std::string &msg) { return met2::abi_fail(code, msg); }

// makes `dev` current for the duration of a C-ABI call and puts the caller's device back afterwards
struct DevGuard {
    int prev = -1;
    hipError_t err;
    explicit DevGuard(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) err = hipSetDevice(dev); else if (err == hipSuccess) prev = -1;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define USE_DEVICE(dev)                                                                                \
    DevGuard dev_guard_(dev);                                                                          \
    if (dev_guard_.err != hipSuccess) return fail(MET2_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(dev_guard_.err))

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(MET2_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
