// met2_tv.hip -- denoise='TV' of the driver (motor/motor_recon_met2_real_data.py:293-304) on gfx950:
//     for every echo volume:  sigma = mean(estimate_sigma(vol));  vol <- denoise_tv_chambolle(vol, weight = 2 sigma, eps = 2e-4, max_num_iter = 200)
// Both functions live in scikit-image (PyWavelets underneath), which is not part of the reference tree; what is restated here are
// the published algorithms they implement, operation for operation as scikit-image orders them:
//   estimate_sigma          Donoho & Johnstone's robust estimator: median(|d|) / Phi^-1(0.75) over the non-zero coefficients d of the
//                           finest all-detail sub-band of a separable db2 transform (half-sample symmetric extension, dyadic
//                           down-sampling, d[o] = sum_j g[j] x_ext[2 o + 1 - j], axis 0 first);
//   denoise_tv_chambolle    Chambolle's projection algorithm (J. Math. Imaging Vis. 20, 2004) in 3-D with forward differences,
//                           tau = 1 / 6, stopping when the energy changes by less than eps x its first value.
//
// All echo volumes go through every step TOGETHER (blockIdx.y = echo), in an echo-major working copy f[t][n0][n1][n2] (n2 contiguous):
//   tv_gather_kernel    [voxel][echo] -> [echo][voxel] through an LDS tile (not needed for a Fortran-ordered volume, which already is echo-major)
//   tv_detail_kernel    the 'ddd' coefficients, 64 taps per coefficient straight from f
//   tv_sigma_kernel     one workgroup per echo: exact median by bisection on the IEEE bit patterns (63 counting passes), weight = factor x sigma
//   tv_iter_kernel      ONE fused stencil kernel per Chambolle iteration: a workgroup owns a (n1, n2) tile and marches along axis 0;
//                       per plane it forms  out = f - div p  (the neighbours of p through L1/L2, the plane of `out` through LDS),
//                       the forward differences, |grad|, the projected update of p (ping-pong buffers) and its share of the energy.
//                       Compulsory HBM traffic: read p (3) + f (1), write p (3) = 56 bytes per (voxel, echo) and iteration.
//   tv_reduce_kernel    one workgroup per echo: adds the workgroups' energy partials in a fixed order, applies the stopping rule and
//                       sets the echo's device-side `done` flag -- later launches return at once for that echo; the host reads nothing
//                       per iteration (it may poll the flags every few iterations to stop launching).
//   tv_final_kernel     out = f - div p of the last iteration's input, written back in the caller's layout.
// Sums that numpy orders -- (p0 + p1) + p2, the three `d += p[ax]` steps, (g0^2 + g1^2) + g2^2 -- keep numpy's axis order whichever way
// the volume lies in memory; products and sums are rounded separately (fp contract off), divisions and the square root are IEEE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/met2_hip.h"
#include "abi_common.hpp"

namespace {

// Daubechies-2 decomposition high-pass filter (pywt.Wavelet('db2').dec_hi)
__constant__ double DB2_HI[4] = {-0.48296291314469025, 0.836516303737469, -0.22414386804185735, -0.12940952255092145};
#define TV_PHI_INV_075 0.6744897501960817          // scipy.stats.norm.ppf(0.75)

struct TvState {
    double e_init, e_prev, e_last, weight, sigma;
    int32_t done;        // 1: no further iteration for this echo
    int32_t iters;       // Chambolle iterations executed
    int32_t parity;      // the ping-pong buffer the LAST executed iteration read: the result is f - div p of that one
    int32_t copy;        // 1: weight <= 0 or not finite -> the echo is copied through
};

__device__ __forceinline__ int reflect_idx(int i, int n)            // half-sample symmetric extension ... b a | a b c ... | ... (any distance)
{
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period; if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

// ---- [voxel][echo] <-> [echo][voxel] -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tv_gather_kernel(int64_t vol, int nt, const double *__restrict__ data, double *__restrict__ F)
{
    extern __shared__ double tile[];                                 // [64][nt + 1]
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    const int nv = (int)std::min<int64_t>(64, vol - v0);
    const int total = nv * nt, ld = nt + 1;
    for (int i = threadIdx.x; i < total; i += 256) { const int v = i / nt, t = i - v * nt; tile[v * ld + t] = data[v0 * nt + i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * nt; i += 256) {
        const int t = i >> 6, v = i & 63;
        if (v < nv) F[(int64_t)t * vol + v0 + v] = tile[v * ld + t];
    }
}

struct TvScatterArgs {
    int64_t vol; int nt;
    const double *P0, *P1;                                           // the echo's result sits in plane 0 of the buffer its last iteration WROTE
    const TvState *state;
    double *out;
};
__global__ __launch_bounds__(256) void tv_scatter_kernel(TvScatterArgs A)
{
    extern __shared__ double tile[];
    const int nt = A.nt, ld = nt + 1;
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    const int nv = (int)std::min<int64_t>(64, A.vol - v0);
    for (int i = threadIdx.x; i < 64 * nt; i += 256) {
        const int t = i >> 6, v = i & 63;
        const double *src = (A.state[t].parity ? A.P0 : A.P1) + (int64_t)t * 3 * A.vol;
        if (v < nv) tile[v * ld + t] = src[v0 + v];
    }
    __syncthreads();
    const int total = nv * nt;
    for (int i = threadIdx.x; i < total; i += 256) { const int v = i / nt, t = i - v * nt; A.out[v0 * nt + i] = tile[v * ld + t]; }
}

// ---- estimate_sigma ---------------------------------------------------------------------------------------------------------
struct TvDetailArgs {
    int n0, n1, n2, c0, c1, c2;
    int64_t vol, nc;
    const double *F;                                                 // [nt][vol]
    double *coef;                                                    // [nt][nc]
};
// REV: the volume lies in memory as (z, y, x) -- numpy's axis 0 is the contiguous one; the separable passes run numpy's axis 0 first.
template <bool REV>
__global__ __launch_bounds__(256) void tv_detail_kernel(TvDetailArgs A)
{
#pragma clang fp contract(off)
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= A.nc) return;
    const int t = blockIdx.y;
    const int o2 = (int)(o % A.c2), o1 = (int)((o / A.c2) % A.c1), o0 = (int)(o / ((int64_t)A.c2 * A.c1));
    int i0[4], i1[4], i2[4];
    for (int j = 0; j < 4; ++j) {
        i0[j] = reflect_idx(2 * o0 + 1 - j, A.n0); i1[j] = reflect_idx(2 * o1 + 1 - j, A.n1); i2[j] = reflect_idx(2 * o2 + 1 - j, A.n2);
    }
    const double *f = A.F + (int64_t)t * A.vol;
    double s3 = 0.0;
    for (int a = 0; a < 4; ++a) {                                    // last pass (numpy axis 2)
        double s2 = 0.0;
        for (int b = 0; b < 4; ++b) {                                // middle pass (numpy axis 1 = memory axis 1 either way)
            double s1 = 0.0;
            for (int c = 0; c < 4; ++c) {                            // first pass (numpy axis 0)
                const int m0 = REV ? i0[a] : i0[c], m2 = REV ? i2[c] : i2[a];
                const double v = DB2_HI[c] * f[((int64_t)m0 * A.n1 + i1[b]) * A.n2 + m2];
                s1 = c ? s1 + v : v;
            }
            const double v = DB2_HI[b] * s1;
            s2 = b ? s2 + v : v;
        }
        const double v = DB2_HI[a] * s2;
        s3 = a ? s3 + v : v;
    }
    A.coef[(int64_t)t * A.nc + o] = s3;
}

__device__ __forceinline__ int wave_sum_i(int v)
{
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)               // butterfly: the same order on every run
{
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct TvSigmaArgs {
    int64_t nc;
    const double *coef;
    TvState *state;
    double factor;                                                   // weight = factor * sigma
    const double *weight_in;                                         // device [nt] or NULL: explicit weights (sigma still reported)
};
// one workgroup per echo: the median of the non-zero |d|, exactly, without sorting.  A positive double orders like its 63-bit pattern,
// so the k-th smallest key is found from the top bit down: (1) over the whole array, two bits per pass (three thresholds), until the
// keys that share the prefix found so far fit the LDS list -- on an echo volume of 128 x 128 x 64 that takes 7-8 passes; (2) those
// candidates are gathered into LDS and the remaining bits are decided there, one per pass; (3) for an even count one more pass over the
// array finds the upper middle value (the smallest key above the lower one, unless the lower one is repeated).
#define TV_SIGMA_CAP 6144
__device__ __forceinline__ unsigned long long tv_key(double v)
{
    return (v == 0.0) ? ~0ull : (unsigned long long)__double_as_longlong(fabs(v));
}
__global__ __launch_bounds__(1024) void tv_sigma_kernel(TvSigmaArgs A)
{
    __shared__ long long red[4][16];
    __shared__ unsigned long long bc[4];
    __shared__ unsigned long long cand[TV_SIGMA_CAP];
    __shared__ int ncand;
    const int t = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const double *c = A.coef + (int64_t)t * A.nc;
    // block-wide sums of up to four per-thread counters
    auto block_sum4 = [&](long long v0, long long v1, long long v2, long long v3, long long *out) {
        int a0 = wave_sum_i((int)v0), a1 = wave_sum_i((int)v1), a2 = wave_sum_i((int)v2), a3 = wave_sum_i((int)v3);
        __syncthreads();                                             // the previous use of red[] has been read
        if (lane == 0) { red[0][w] = a0; red[1][w] = a1; red[2][w] = a2; red[3][w] = a3; }
        __syncthreads();
        for (int j = 0; j < 4; ++j) { long long sum = 0; for (int i = 0; i < 16; ++i) sum += red[j][i]; out[j] = sum; }
    };
    long long r[4];
    {
        int cnt = 0, bad = 0;
#pragma unroll 4
        for (int64_t i = tid; i < A.nc; i += 1024) { const double v = c[i]; cnt += (v != 0.0); bad += !(fabs(v) <= 1.79769313486231570815e308); }
        block_sum4(cnt, bad, 0, 0, r);
    }
    const long long m = r[0], nbad = r[1];
    double sigma;
    if (nbad) sigma = __builtin_nan("");
    else if (m == 0) sigma = 0.0;
    else {
        const long long k1 = (m - 1) / 2;
        unsigned long long prefix = 0;
        long long below = 0, inb = m;                                // keys under the prefix; keys that share it
        int bit = 62;                                                // the highest undecided bit
        while (inb > TV_SIGMA_CAP && bit >= 1) {
            const unsigned long long q = 1ull << (bit - 1), c1 = prefix + q, c2 = prefix + 2 * q, c3 = prefix + 3 * q;
            int n1 = 0, n2 = 0, n3 = 0;
#pragma unroll 4
            for (int64_t i = tid; i < A.nc; i += 1024) { const unsigned long long key = tv_key(c[i]); n1 += key < c1; n2 += key < c2; n3 += key < c3; }
            block_sum4(n1, n2, n3, 0, r);
            const long long e = below + inb;
            if (r[2] <= k1)      { prefix = c3; below = r[2]; inb = e - r[2]; }
            else if (r[1] <= k1) { prefix = c2; below = r[1]; inb = r[2] - r[1]; }
            else if (r[0] <= k1) { prefix = c1; below = r[0]; inb = r[1] - r[0]; }
            else                 { inb = r[0] - below; }
            bit -= 2;
        }
        unsigned long long a_key;
        if (inb <= TV_SIGMA_CAP) {
            if (tid == 0) ncand = 0;
            __syncthreads();
            const unsigned long long span = (bit >= 0) ? ((1ull << bit) << 1) : 1ull;
            for (int64_t i = tid; i < A.nc; i += 1024) {
                const unsigned long long key = tv_key(c[i]);
                if (key - prefix < span) cand[atomicAdd(&ncand, 1)] = key;
            }
            __syncthreads();
            const int nc2 = ncand;
            for (; bit >= 0; --bit) {
                const unsigned long long c1 = prefix | (1ull << bit);
                int n1 = 0;
                for (int i = tid; i < nc2; i += 1024) n1 += cand[i] < c1;
                block_sum4(n1, 0, 0, 0, r);
                if (below + r[0] <= k1) prefix = c1;
            }
            a_key = prefix;
        } else {                                                     // bit == 0 left and still too many equal-prefix keys: decide it on the array
            for (; bit >= 0; --bit) {
                const unsigned long long c1 = prefix | (1ull << bit);
                int n1 = 0;
                for (int64_t i = tid; i < A.nc; i += 1024) n1 += tv_key(c[i]) < c1;
                block_sum4(n1, 0, 0, 0, r);
                if (r[0] <= k1) prefix = c1;
            }
            a_key = prefix;
        }
        const double a = __longlong_as_double((long long)a_key);
        double med = a;
        if ((m & 1) == 0) {                                          // np.median: the mean of the two middle values
            int nle = 0;
            unsigned long long mn = ~0ull;
#pragma unroll 4
            for (int64_t i = tid; i < A.nc; i += 1024) {
                const unsigned long long key = tv_key(c[i]);
                nle += key <= a_key;
                if (key > a_key && key < mn) mn = key;
            }
            for (int off = 32; off; off >>= 1) { const unsigned long long o = __shfl_xor(mn, off, 64); mn = o < mn ? o : mn; }
            block_sum4(nle, 0, 0, 0, r);
            __syncthreads();
            if (lane == 0) red[1][w] = (long long)mn;
            __syncthreads();
            unsigned long long gm = ~0ull;
            for (int i = 0; i < 16; ++i) { const unsigned long long o = (unsigned long long)red[1][i]; gm = o < gm ? o : gm; }
            const double b = (r[0] >= k1 + 2) ? a : __longlong_as_double((long long)gm);
            med = (a + b) / 2.0;
        }
        sigma = med / TV_PHI_INV_075;
    }
    if (tid == 0) {
        TvState s;
        s.e_init = s.e_prev = s.e_last = 0.0;
        s.sigma = sigma;
        s.weight = A.weight_in ? A.weight_in[t] : A.factor * sigma;
        s.copy = !(s.weight > 0.0) || !(s.weight <= 1.79769313486231570815e308);
        s.done = s.copy; s.iters = 0; s.parity = 0;
        A.state[t] = s;
    }
    (void)bc;
}

// explicit weights, no sigma estimate wanted
__global__ void tv_state_init_kernel(int nt, const double *weight_in, TvState *state)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    TvState s;
    s.e_init = s.e_prev = s.e_last = 0.0; s.sigma = __builtin_nan("");
    s.weight = weight_in[t];
    s.copy = !(s.weight > 0.0) || !(s.weight <= 1.79769313486231570815e308);
    s.done = s.copy; s.iters = 0; s.parity = 0;
    state[t] = s;
}

// ---- one Chambolle iteration ---------------------------------------------------------------------------------------------------
struct TvIterArgs {
    int n0, n1, n2;                                                  // memory order, n2 contiguous
    int xlen, nseg, nt1, nt2, step1, step2;                          // planes per segment; tiles per echo; rows / lanes a tile OWNS
    int ntiles, iter;
    int64_t vol;
    const double *F, *Pin;
    double *Pout;
    double *partial;                                                 // [nt][ntiles][2]
    const TvState *state;
};

struct TvPlane { double p0, p1, p2, f, p1m, p2m; };

template <bool REV>
__device__ __forceinline__ double tv_minus_div(const TvPlane &q, double p0m, bool h0, bool h1, bool h2)
{
#pragma clang fp contract(off)
    // numpy: d = -p.sum(0); d[1:] += p[0][:-1]; d[:, 1:] += p[1][:, :-1]; d[:, :, 1:] += p[2][:, :, :-1]
    double d;
    if (!REV) {
        d = -((q.p0 + q.p1) + q.p2);
        if (h0) d = d + p0m;
        if (h1) d = d + q.p1m;
        if (h2) d = d + q.p2m;
    } else {
        d = -((q.p2 + q.p1) + q.p0);
        if (h2) d = d + q.p2m;
        if (h1) d = d + q.p1m;
        if (h0) d = d + p0m;
    }
    return d;
}

#ifndef MET2_TV_NT
#define MET2_TV_NT 1           // the new p leaves with non-temporal stores (nothing in this launch reads it): 0.379 -> 0.366 ms per iteration
#endif
#ifndef MET2_TV_WPE
#define MET2_TV_WPE 1          // minimum waves per SIMD the kernel is compiled for: 1 = 72 VGPRs, 7 waves; 8 = 64 VGPRs with 6 spilled: 0.484 instead of 0.382 ms per iteration
#endif
__device__ __forceinline__ double bcast_lane(double v, int l)        // lane l's value in every lane (l a compile-time constant at the call sites)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// EDGE: the contiguous axis is longer than a wave (n2 > 64).  Tiles along it are 64 lanes wide WITHOUT overlap; what lane 63 needs from the
// next tile -- `out` at column z0 + 64 of its own row -- is formed in the wave itself: per plane, lanes 0..4 fetch that column's p0, p1, p2,
// f and p1 of the row below with ONE load, the values are broadcast and every lane forms the (wave-uniform) out.  Eight registers instead
// of a 64th lane that owns nothing: with overlapping 63-lane tiles a volume of 128 along this axis took three tiles per row, the last one
// with two live lanes, and its loads straddled the 512-byte segments (3.25 instead of 5.2 TB/s; a Fortran-ordered volume has nx here).
template <int OY, bool REV, bool EDGE>
__global__ __launch_bounds__(64 * OY, MET2_TV_WPE) void tv_iter_kernel(TvIterArgs A)
{
#pragma clang fp contract(off)
    const int t = blockIdx.y;
    if (A.state[t].done) return;
    // workgroups of one XCD (blockIdx.x mod 8) walk a contiguous eighth of the tile list: tiles that share a halo row meet in one L2
    const int per = gridDim.x >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (logical >= A.ntiles) return;
    const int t1 = logical % A.nt1, seg = (logical / A.nt1) % A.nseg, t2 = logical / (A.nt1 * A.nseg);
    const int wy = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n0 = A.n0, n1 = A.n1, n2 = A.n2;
    const int y = t1 * A.step1 + wy, z = t2 * A.step2 + lane;
    const int xs = seg * A.xlen, xe = min(xs + A.xlen, n0);
    const bool in = y < n1 && z < n2;
    const bool own = in && wy < A.step1 && lane < A.step2;
    const bool h1 = y >= 1, h2 = z >= 1;                             // a lower neighbour exists along axis 1 / 2
    const bool u1 = in && y + 1 < n1 && wy + 1 < OY, u2 = in && z + 1 < n2 && lane + 1 < 64;   // an upper neighbour, held by this workgroup
    const bool etile = EDGE && (z - lane + 64 < n2);                 // the tile has a neighbour along axis 2 (wave-uniform)
    const bool e2 = etile && in && lane == 63;                       // ... whose first column is this lane's upper neighbour
    const double weight = A.state[t].weight;
    const double tau = 1.0 / 6.0;
    const double r = tau / weight;
    const int64_t s0 = (int64_t)n1 * n2;
    const double *__restrict__ f = A.F + (int64_t)t * A.vol;
    const double *__restrict__ q0 = A.Pin + (int64_t)t * 3 * A.vol, *__restrict__ q1 = q0 + A.vol, *__restrict__ q2 = q1 + A.vol;
    double *__restrict__ w0 = A.Pout + (int64_t)t * 3 * A.vol, *__restrict__ w1 = w0 + A.vol, *__restrict__ w2 = w1 + A.vol;
    __shared__ double outp[2][OY][64];
    __shared__ double red[2][OY];

    const int64_t at = ((int64_t)xs * n1 + (in ? y : 0)) * n2 + (in ? z : 0);
    auto load = [&](int64_t i) {
        TvPlane q;
        q.p0 = q0[i]; q.p1 = q1[i]; q.p2 = q2[i]; q.f = f[i];      // (non-temporal loads of p0 and f, which no other thread reads: 0.367 against 0.363 ms, not kept)
        q.p1m = h1 ? q1[i - n2] : 0.0; q.p2m = h2 ? q2[i - 1] : 0.0;
        return q;
    };
    TvPlane cur = {0, 0, 0, 0, 0, 0}, nxt = cur, nn = cur;
    double out_c = 0.0, d_c = 0.0;
    if (in) {
        cur = load(at);
        const double p0m = xs >= 1 ? q0[at - s0] : 0.0;
        d_c = tv_minus_div<REV>(cur, p0m, xs >= 1, h1, h2);
        out_c = cur.f + d_c;
        if (xs + 1 < n0) nxt = load(at + s0);
    }
    // the neighbour tile's first column (EDGE): lane k < 6 fetches item k of {p0, p1, p2, f, p1 of the row below, p0 of the plane before}
    const double *hbase = (lane == 0 || lane == 5) ? q0 : ((lane == 1 || lane == 4) ? q1 : (lane == 2 ? q2 : f));
    const bool hrow = etile && y < n1;
    const int64_t hat = ((int64_t)xs * n1 + (hrow ? y : 0)) * n2 + (hrow ? z - lane + 64 : 0) - (lane == 4 ? n2 : 0) - (lane == 5 ? s0 : 0);
    const bool hact = hrow && (lane < 4 || (lane == 4 && h1) || (lane == 5 && xs >= 1));
    auto hload = [&](int k) { return (hact && (lane < 5 || k == 0)) ? hbase[hat + (int64_t)k * s0] : 0.0; };
    auto hout = [&](double hv, double p0prev, bool hasprev, double p2own) {      // out at the neighbour column from one plane's six numbers
        TvPlane q;
        q.p0 = bcast_lane(hv, 0); q.p1 = bcast_lane(hv, 1); q.p2 = bcast_lane(hv, 2); q.f = bcast_lane(hv, 3); q.p1m = bcast_lane(hv, 4);
        q.p2m = p2own;                                               // its lower neighbour along axis 2 is this wave's last column
        return q.f + tv_minus_div<REV>(q, p0prev, hasprev, h1, true);
    };
    double hnxt = 0.0, hnn = 0.0, hp0 = 0.0, oh_c = 0.0;             // next plane's six numbers, the plane after; p0 of the current plane; out of the current plane
    if (EDGE && etile) {
        const double h0v = hload(0);
        oh_c = hout(h0v, bcast_lane(h0v, 5), xs >= 1, bcast_lane(cur.p2, 63));
        hp0 = bcast_lane(h0v, 0);
        if (xs + 1 < n0) hnxt = hload(1);
    }
    outp[0][wy][lane] = out_c;
    __syncthreads();
    double acc_d = 0.0, acc_n = 0.0;
    int64_t i = at;
    for (int x = xs; x < xe; ++x, i += s0) {
        const int b = (x - xs) & 1;
        const bool more = x + 1 < n0;
        if (in && x + 2 < n0 && x + 1 < xe) nn = load(i + 2 * s0);   // two planes ahead: in flight under this plane's arithmetic
        if (EDGE && etile && x + 2 < n0 && x + 1 < xe) hnn = hload(x + 2 - xs);
        double out_n = 0.0, d_n = 0.0;
        if (more) { d_n = tv_minus_div<REV>(nxt, cur.p0, true, h1, h2); out_n = nxt.f + d_n; }
        double oh_n = 0.0;
        if (EDGE && etile && more) oh_n = hout(hnxt, hp0, true, bcast_lane(nxt.p2, 63));
        const double g0 = more ? out_n - out_c : 0.0;
        const double g1 = u1 ? outp[b][wy + 1][lane] - out_c : 0.0;
        const double g2 = (EDGE && e2) ? oh_c - out_c : (u2 ? outp[b][wy][lane + 1] - out_c : 0.0);
        outp[b ^ 1][wy][lane] = out_n;
        if (own) {
            const double nrm = REV ? sqrt((g2 * g2 + g1 * g1) + g0 * g0) : sqrt((g0 * g0 + g1 * g1) + g2 * g2);
            acc_d += d_c * d_c;
            acc_n += nrm;
            const double den = nrm * r + 1.0;
#if MET2_TV_NT
            __builtin_nontemporal_store((cur.p0 - tau * g0) / den, &w0[i]);
            __builtin_nontemporal_store((cur.p1 - tau * g1) / den, &w1[i]);
            __builtin_nontemporal_store((cur.p2 - tau * g2) / den, &w2[i]);
#else
            w0[i] = (cur.p0 - tau * g0) / den;
            w1[i] = (cur.p1 - tau * g1) / den;
            w2[i] = (cur.p2 - tau * g2) / den;
#endif
        }
        cur = nxt; nxt = nn; out_c = out_n; d_c = d_n;
        if (EDGE) { hp0 = bcast_lane(hnxt, 0); hnxt = hnn; oh_c = oh_n; }
        __syncthreads();
    }
    acc_d = wave_sum_d(acc_d); acc_n = wave_sum_d(acc_n);
    if (lane == 0) { red[0][wy] = acc_d; red[1][wy] = acc_n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, c = 0.0;
        for (int k = 0; k < OY; ++k) { a += red[0][k]; c += red[1][k]; }
        double *dst = A.partial + ((int64_t)t * A.ntiles + logical) * 2;
        dst[0] = a; dst[1] = c;
    }
}

struct TvReduceArgs {
    int ntiles, iter, max_iter;
    double eps, size;
    const double *partial;
    TvState *state;
};
__global__ __launch_bounds__(256) void tv_reduce_kernel(TvReduceArgs A)
{
#pragma clang fp contract(off)
    const int t = blockIdx.x;
    if (A.state[t].done) return;
    __shared__ double red[2][256];
    const double *p = A.partial + (int64_t)t * A.ntiles * 2;
    double a = 0.0, c = 0.0;
    for (int k = threadIdx.x; k < A.ntiles; k += 256) { a += p[2 * k]; c += p[2 * k + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = c;
    __syncthreads();
    for (int h = 128; h; h >>= 1) {
        if ((int)threadIdx.x < h) { red[0][threadIdx.x] += red[0][threadIdx.x + h]; red[1][threadIdx.x] += red[1][threadIdx.x + h]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        TvState s = A.state[t];
        double E = red[0][0];                                        // E = (d ** 2).sum(); E += weight * norm.sum(); E /= size
        E = E + s.weight * red[1][0];
        E = E / A.size;
        s.e_last = E;
        s.iters = A.iter + 1;
        s.parity = A.iter & 1;
        if (A.iter == 0) { s.e_init = E; s.e_prev = E; }
        else if (fabs(s.e_prev - E) < A.eps * s.e_init) s.done = 1;
        else s.e_prev = E;
        if (A.iter + 1 >= A.max_iter) s.done = 1;
        A.state[t] = s;
    }
}

// ---- the result: out = f - div p of the last iteration's input ------------------------------------------------------------------
struct TvFinalArgs {
    int n0, n1, n2;
    int64_t vol;
    const double *F, *P0, *P1;
    double *Q0, *Q1;                                                 // writable views of P0 / P1 (echo-major result, plane 0)
    const TvState *state;
    double *out; int64_t out_vs, out_es;                             // direct == 1: out[v * out_vs + t * out_es]
    int direct;
};
template <bool REV>
__global__ __launch_bounds__(256) void tv_final_kernel(TvFinalArgs A)
{
#pragma clang fp contract(off)
    const int t = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.vol) return;
    const TvState s = A.state[t];
    const double fv = A.F[(int64_t)t * A.vol + i];
    double o = fv;
    if (!s.copy && s.iters > 0) {
        const int z = (int)(i % A.n2), y = (int)((i / A.n2) % A.n1), x = (int)(i / ((int64_t)A.n2 * A.n1));
        const double *q0 = (s.parity ? A.P1 : A.P0) + (int64_t)t * 3 * A.vol, *q1 = q0 + A.vol, *q2 = q1 + A.vol;
        TvPlane q;
        q.p0 = q0[i]; q.p1 = q1[i]; q.p2 = q2[i]; q.f = fv;
        q.p1m = y >= 1 ? q1[i - A.n2] : 0.0; q.p2m = z >= 1 ? q2[i - 1] : 0.0;
        const double p0m = x >= 1 ? q0[i - (int64_t)A.n1 * A.n2] : 0.0;
        o = fv + tv_minus_div<REV>(q, p0m, x >= 1, y >= 1, z >= 1);
    }
    if (A.direct) A.out[i * A.out_vs + (int64_t)t * A.out_es] = o;
    else ((s.parity ? A.Q0 : A.Q1) + (int64_t)t * 3 * A.vol)[i] = o;   // the buffer the last iteration wrote is free
}

__global__ void tv_report_kernel(int nt, const TvState *state, double *sigma, int32_t *iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    if (sigma) sigma[t] = state[t].sigma;
    if (iters) iters[t] = state[t].iters;
}

struct TvLayout {
    int64_t vol, nc, elems;
    int c0, c1, c2, ntiles, nt1, nt2, nseg, step1, step2, xlen, oy;
    size_t off_F, off_P0, off_P1, off_coef, off_partial, off_state, off_weight, bytes;
};

int tv_env(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

void tv_layout(int n0, int n1, int n2, int nt, TvLayout &L)
{
    L.vol = (int64_t)n0 * n1 * n2; L.elems = L.vol * nt;
    L.c0 = (n0 + 3) / 2; L.c1 = (n1 + 3) / 2; L.c2 = (n2 + 3) / 2;
    L.nc = (int64_t)L.c0 * L.c1 * L.c2;
    L.oy = tv_env("MET2_TV_OY", 8);                                  // rows of a tile = waves of a workgroup; measured 4 / 8 / 16: 0.387 / 0.381 / 0.428 ms per iteration
    if (L.oy != 4 && L.oy != 8 && L.oy != 16) L.oy = 8;
    L.step1 = n1 <= L.oy ? L.oy : L.oy - 1;                          // a tile that spans the axis needs no halo row
    L.step2 = 64;                                                    // no overlap along the contiguous axis (tv_iter_kernel: EDGE)
    L.nt1 = (n1 + L.step1 - 1) / L.step1; L.nt2 = (n2 + L.step2 - 1) / L.step2;
    L.xlen = std::max(1, std::min(n0, tv_env("MET2_TV_XLEN", 16)));
    L.nseg = (n0 + L.xlen - 1) / L.xlen;
    L.ntiles = L.nt1 * L.nt2 * L.nseg;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t o = 0;
    L.off_F = o; o += up(sizeof(double) * (size_t)L.elems);
    L.off_P0 = o; o += up(sizeof(double) * 3 * (size_t)L.elems);
    L.off_P1 = o; o += up(sizeof(double) * 3 * (size_t)L.elems);
    L.off_coef = o; o += up(sizeof(double) * (size_t)L.nc * nt);
    L.off_partial = o; o += up(sizeof(double) * 2 * (size_t)L.ntiles * nt);
    L.off_state = o; o += up(sizeof(TvState) * (size_t)nt);
    L.off_weight = o; o += up(sizeof(double) * (size_t)nt);
    L.bytes = o;
}

// HIP events around the iteration launches of the calling thread's most recent met2_tv_chambolle (met2_tv_last_timing)
struct TvTiming {
    hipEvent_t e0 = nullptr, e1 = nullptr, pe[2] = {nullptr, nullptr};   // pe: the two poll copies in flight
    TvState *hpin = nullptr;                                         // pinned: [2][hpin_cap] copies of the echoes' states
    int hpin_cap = 0;
    int launches = 0, device = -1;
    bool valid = false;
};
thread_local TvTiming g_tv_timing;

template <bool REV>
void launch_iter(const TvIterArgs &A, int oy, int nt, hipStream_t s)
{
    const dim3 grid((unsigned)(((A.ntiles + 7) / 8) * 8), (unsigned)nt);
    const bool edge = A.n2 > 64;
    if (oy == 4) {
        if (edge) hipLaunchKernelGGL((tv_iter_kernel<4, REV, true>), grid, dim3(256), 0, s, A);
        else      hipLaunchKernelGGL((tv_iter_kernel<4, REV, false>), grid, dim3(256), 0, s, A);
    } else if (oy == 8) {
        if (edge) hipLaunchKernelGGL((tv_iter_kernel<8, REV, true>), grid, dim3(512), 0, s, A);
        else      hipLaunchKernelGGL((tv_iter_kernel<8, REV, false>), grid, dim3(512), 0, s, A);
    } else {
        if (edge) hipLaunchKernelGGL((tv_iter_kernel<16, REV, true>), grid, dim3(1024), 0, s, A);
        else      hipLaunchKernelGGL((tv_iter_kernel<16, REV, false>), grid, dim3(1024), 0, s, A);
    }
}

}  // namespace

extern "C" int64_t met2_tv_work_bytes(int32_t nx, int32_t ny, int32_t nz, int32_t n_te, int32_t echo_major)
{
    if (nx < 1 || ny < 1 || nz < 1 || n_te < 1) return 0;
    TvLayout L;
    tv_layout(echo_major ? nz : nx, ny, echo_major ? nx : nz, n_te, L);
    return (int64_t)L.bytes;
}

extern "C" int met2_tv_chambolle(int32_t device, int32_t nx, int32_t ny, int32_t nz, int32_t n_te, const double *data, int32_t echo_major,
                                 const double *weight, double weight_factor, double eps, int32_t max_num_iter, int32_t poll_every,
                                 double *out, double *sigma, int32_t *iters, void *work, int64_t work_bytes, void *stream)
{
    if (nx < 0 || ny < 0 || nz < 0 || n_te < 1) return fail(MET2_E_INVALID, "bad shape");
    if (n_te > 65535) return fail(MET2_E_UNSUPPORTED, "at most 65535 echo volumes");
    if ((int64_t)nx * ny * nz == 0) return MET2_OK;
    if (!data || !out) return fail(MET2_E_INVALID, "NULL argument");
    if (max_num_iter < 1) return fail(MET2_E_INVALID, "max_num_iter must be at least 1");
    if (!(eps >= 0.0)) return fail(MET2_E_INVALID, "eps must be non-negative");
    const bool rev = echo_major != 0;
    // memory order of one echo volume, slowest axis first: (x, y, z) of a C-ordered [nx][ny][nz][nt] array, (z, y, x) of a Fortran-ordered one
    const int n0 = rev ? nz : nx, n1 = ny, n2 = rev ? nx : nz;
    TvLayout L;
    tv_layout(n0, n1, n2, n_te, L);
    if ((L.vol + 255) / 256 > 0x7fffffffLL || (L.nc + 255) / 256 > 0x7fffffffLL) return fail(MET2_E_UNSUPPORTED, "volume too large for one launch");
    if (work && work_bytes < (int64_t)L.bytes) return fail(MET2_E_INVALID, "work buffer smaller than met2_tv_work_bytes()");
    USE_DEVICE(device);
    hipStream_t s = (hipStream_t)stream;
    char *W = (char *)work;
    if (!W) HIPCHK(hipMalloc((void **)&W, L.bytes));
    struct Free { char *p; bool own; ~Free() { if (own && p) (void)hipFree(p); } } guard{W, work == nullptr};
    double *F = (double *)(W + L.off_F), *P0 = (double *)(W + L.off_P0), *P1 = (double *)(W + L.off_P1);
    double *coef = (double *)(W + L.off_coef), *partial = (double *)(W + L.off_partial);
    TvState *state = (TvState *)(W + L.off_state);
    const size_t lds_t = sizeof(double) * 64 * ((size_t)n_te + 1);
    if (!rev && lds_t > 64 * 1024) return fail(MET2_E_UNSUPPORTED, "more than 127 echoes: pass the volume echo-major");

    // 1. echo-major working copy
    const double *Fsrc = F;
    if (rev) Fsrc = data;                                            // already [echo][z][y][x]
    else hipLaunchKernelGGL(tv_gather_kernel, dim3((unsigned)((L.vol + 63) / 64)), dim3(256), lds_t, s, L.vol, (int)n_te, data, F);
    // 2. noise level and weight per echo
    double *dweight = nullptr;
    if (weight) {
        dweight = (double *)(W + L.off_weight);
        HIPCHK(hipMemcpyAsync(dweight, weight, sizeof(double) * n_te, hipMemcpyHostToDevice, s));
    }
    if (!weight || sigma) {
        TvDetailArgs D;
        D.n0 = n0; D.n1 = n1; D.n2 = n2; D.c0 = L.c0; D.c1 = L.c1; D.c2 = L.c2; D.vol = L.vol; D.nc = L.nc; D.F = Fsrc; D.coef = coef;
        const dim3 g((unsigned)((L.nc + 255) / 256), (unsigned)n_te);
        if (rev) hipLaunchKernelGGL(tv_detail_kernel<true>, g, dim3(256), 0, s, D);
        else hipLaunchKernelGGL(tv_detail_kernel<false>, g, dim3(256), 0, s, D);
        TvSigmaArgs S;
        S.nc = L.nc; S.coef = coef; S.state = state; S.factor = weight_factor; S.weight_in = dweight;
        hipLaunchKernelGGL(tv_sigma_kernel, dim3((unsigned)n_te), dim3(1024), 0, s, S);
    } else {
        hipLaunchKernelGGL(tv_state_init_kernel, dim3((unsigned)((n_te + 63) / 64)), dim3(64), 0, s, (int)n_te, (const double *)dweight, state);
    }
    HIPCHK(hipGetLastError());
    // 3. Chambolle iterations, all echoes per launch
    HIPCHK(hipMemsetAsync(P0, 0, sizeof(double) * 3 * (size_t)L.elems, s));
    TvIterArgs I;
    I.n0 = n0; I.n1 = n1; I.n2 = n2; I.xlen = L.xlen; I.nseg = L.nseg; I.nt1 = L.nt1; I.nt2 = L.nt2; I.step1 = L.step1; I.step2 = L.step2;
    I.ntiles = L.ntiles; I.vol = L.vol; I.F = Fsrc; I.partial = partial; I.state = state;
    TvReduceArgs R;
    R.ntiles = L.ntiles; R.max_iter = max_num_iter; R.eps = eps; R.size = (double)L.vol; R.partial = partial; R.state = state;
    TvTiming &T = g_tv_timing;
    T.valid = false;
    if (!T.e0 || T.device != device) {
        if (T.e0) {
            (void)hipEventDestroy(T.e0); (void)hipEventDestroy(T.e1); (void)hipEventDestroy(T.pe[0]); (void)hipEventDestroy(T.pe[1]);
            T.e0 = T.e1 = T.pe[0] = T.pe[1] = nullptr;
        }
        if (T.hpin) { (void)hipHostFree(T.hpin); T.hpin = nullptr; T.hpin_cap = 0; }
        HIPCHK(hipEventCreate(&T.e0)); HIPCHK(hipEventCreate(&T.e1));
        HIPCHK(hipEventCreateWithFlags(&T.pe[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&T.pe[1], hipEventDisableTiming));
        T.device = device;
    }
    if (poll_every > 0 && T.hpin_cap < n_te) {
        if (T.hpin) { (void)hipHostFree(T.hpin); T.hpin = nullptr; T.hpin_cap = 0; }
        HIPCHK(hipHostMalloc((void **)&T.hpin, sizeof(TvState) * 2 * (size_t)n_te, hipHostMallocDefault));
        T.hpin_cap = n_te;
    }
    HIPCHK(hipEventRecord(T.e0, s));
    T.launches = 0;
    for (int it = 0; it < max_num_iter; ++it) {
        I.iter = it; I.Pin = (it & 1) ? P1 : P0; I.Pout = (it & 1) ? P0 : P1;
        if (rev) launch_iter<true>(I, L.oy, n_te, s); else launch_iter<false>(I, L.oy, n_te, s);
        R.iter = it;
        hipLaunchKernelGGL(tv_reduce_kernel, dim3((unsigned)n_te), dim3(256), 0, s, R);
        T.launches = it + 1;
        // The host stops launching once every echo is done.  It looks at the flags ONE batch late: the copy of poll k is read after batch
        // k + 1 has been enqueued, so the stream never runs dry while the host waits (a blocking read per poll left a 40 us bubble each
        // time; the price is up to poll_every more launches that return at once).
        if (poll_every > 0 && (it + 1) % poll_every == 0 && it + 1 < max_num_iter) {
            const int k = (it + 1) / poll_every - 1;
            HIPCHK(hipMemcpyAsync(T.hpin + (size_t)(k & 1) * T.hpin_cap, state, sizeof(TvState) * n_te, hipMemcpyDeviceToHost, s));
            HIPCHK(hipEventRecord(T.pe[k & 1], s));
            if (k >= 1) {
                HIPCHK(hipEventSynchronize(T.pe[(k - 1) & 1]));
                const TvState *h = T.hpin + (size_t)((k - 1) & 1) * T.hpin_cap;
                bool all = true;
                for (int t = 0; t < n_te; ++t) all = all && h[t].done;
                if (all) break;
            }
        }
    }
    HIPCHK(hipEventRecord(T.e1, s));
    T.valid = true;
    HIPCHK(hipGetLastError());
    // 4. the result in the caller's layout
    TvFinalArgs Z;
    Z.n0 = n0; Z.n1 = n1; Z.n2 = n2; Z.vol = L.vol; Z.F = Fsrc; Z.P0 = P0; Z.P1 = P1; Z.Q0 = P0; Z.Q1 = P1; Z.state = state;
    Z.out = out; Z.out_vs = 1; Z.out_es = L.vol; Z.direct = rev ? 1 : 0;
    const dim3 gf((unsigned)((L.vol + 255) / 256), (unsigned)n_te);
    if (rev) hipLaunchKernelGGL(tv_final_kernel<true>, gf, dim3(256), 0, s, Z);
    else {
        hipLaunchKernelGGL(tv_final_kernel<false>, gf, dim3(256), 0, s, Z);
        TvScatterArgs C;
        C.vol = L.vol; C.nt = n_te; C.P0 = P0; C.P1 = P1; C.state = state; C.out = out;
        hipLaunchKernelGGL(tv_scatter_kernel, dim3((unsigned)((L.vol + 63) / 64)), dim3(256), lds_t, s, C);
    }
    if (sigma || iters) hipLaunchKernelGGL(tv_report_kernel, dim3((unsigned)((n_te + 63) / 64)), dim3(64), 0, s, (int)n_te, (const TvState *)state, sigma, iters);
    HIPCHK(hipGetLastError());
    if (!work) HIPCHK(hipStreamSynchronize(s));
    return MET2_OK;
}

extern "C" int met2_tv_last_timing(double *iter_ms, int32_t *launches)
{
    TvTiming &T = g_tv_timing;
    if (!T.valid) return fail(MET2_E_STATE, "no met2_tv_chambolle call on this thread yet");
    USE_DEVICE(T.device);
    HIPCHK(hipEventSynchronize(T.e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, T.e0, T.e1));
    if (iter_ms) *iter_ms = ms;
    if (launches) *launches = T.launches;
    return MET2_OK;
}
