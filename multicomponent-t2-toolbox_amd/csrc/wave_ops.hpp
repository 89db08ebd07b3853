// wave_ops.hpp -- wave64 cross-lane primitives for gfx950 (CDNA4).
// One wavefront = 64 lanes; every helper here must be called under full EXEC
// (wave-uniform control flow).
#pragma once
#include <hip/hip_runtime.h>

namespace met2 {

typedef unsigned long long u64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// The lane number again, but opaque to the optimiser.  Per-lane constants of an inlined routine (column bases, byte offsets,
// tile coordinates) are loop-invariant in the voxel loop: the compiler hoists them out of it, runs out of registers and spills
// them (BayesReg at 168 VGPRs: 32 such 64-bit values, reloaded from scratch in the inner loops -- 110 KB of fetches per voxel).
// Derived from this value they are recomputed in place: a few integer instructions per call.
__device__ __forceinline__ int lane_opaque(int lane) { asm volatile("" : "+v"(lane)); return lane; }

// broadcast lane `l` (wave-uniform) of v to all lanes (v_readlane_b32 x2 -> SGPR pair)
__device__ __forceinline__ double bcast(double v, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int bcast_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// per-lane gather from lane `src` (ds_bpermute_b32 x2)
__device__ __forceinline__ double gather(double v, int src)
{
    int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int gather_i(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// DPP controls: quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E, row_half_mirror=0x141, row_mirror=0x140
#define MET2_ROW_REDUCE(v, OP)                 \
    v = OP(v, dpp_mov<0xB1>(v));               \
    v = OP(v, dpp_mov<0x4E>(v));               \
    v = OP(v, dpp_mov<0x141>(v));              \
    v = OP(v, dpp_mov<0x140>(v));

__device__ __forceinline__ double op_add(double a, double b) { return a + b; }
__device__ __forceinline__ double op_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ double op_min(double a, double b) { return fmin(a, b); }

__device__ __forceinline__ double wave_sum(double v)
{
    MET2_ROW_REDUCE(v, op_add)
    return (bcast(v, 0) + bcast(v, 16)) + (bcast(v, 32) + bcast(v, 48));
}
// two sums at once (independent chains interleave)
__device__ __forceinline__ void wave_sum2(double &a, double &b)
{
    MET2_ROW_REDUCE(a, op_add)
    MET2_ROW_REDUCE(b, op_add)
    a = (bcast(a, 0) + bcast(a, 16)) + (bcast(a, 32) + bcast(a, 48));
    b = (bcast(b, 0) + bcast(b, 16)) + (bcast(b, 32) + bcast(b, 48));
}
__device__ __forceinline__ double wave_max(double v)
{
    MET2_ROW_REDUCE(v, op_max)
    return fmax(fmax(bcast(v, 0), bcast(v, 16)), fmax(bcast(v, 32), bcast(v, 48)));
}
__device__ __forceinline__ double wave_min(double v)
{
    MET2_ROW_REDUCE(v, op_min)
    return fmin(fmin(bcast(v, 0), bcast(v, 16)), fmin(bcast(v, 32), bcast(v, 48)));
}

__device__ __forceinline__ u64 ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int first_lane(u64 m) { return __builtin_ctzll(m); }

} // namespace met2
