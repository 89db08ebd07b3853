// met2_hip.hip -- gfx950 kernels and the C ABI of include/met2_hip.h.
//
// Kernels
//   epg_dictionary_kernel   epg/epg.py:47-162      one wavefront per (T2, flip angle); the EPG
//                                                  coherence orders live on the lanes, the
//                                                  shift operator is a lane shuffle
//   gram_kernel             B_fa = D_fa^T D_fa, D_fa^T   (once per plan)
//   classify/scan/scatter   gates of motor:124,131 + counting sort of voxels by FA index, so that
//                           neighbouring entries of the work list read the same D / B rows from L2
//   fit_kernel<METHOD>      motor:113-162 + motor:443-472: one voxel per wavefront; every wave of the
//                           persistent workgroups pulls its own voxels from per-XCD queue cursors over
//                           the FA-sorted list; D, D^T, B and K rows are read through L1/L2
//   fa_kernel               fa_estimation.py:74-111: brute force over the plan's flip angles
//   fa_spline_kernel        fa_estimation.py:54-59
//   roi_reduce / roi_kernel motor_recon_met2_real_data_ROI.py:405-420: per-ROI mean signal and mean kernel
//   nesma_kernel, smooth_axis_kernel   motor:305-343
//   finalize_unfitted       zeros / all-zero-spectrum metrics for gated-out voxels
//   metrics_kernel          motor:443-472 standalone
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/met2_hip.h"
#include "abi_common.hpp"
#include "nnls_wave.hpp"
#include "objectives.hpp"
#include "fit_kernel.hpp"

using namespace met2;

#ifdef MET2_SPLIT_TU
// the fit kernels are instantiated in met2_fit_*.hip (fit_kernel.hpp)
extern template int launch_fit_nb<0, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<0, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<1, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<1, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<2, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<2, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<3, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<3, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<4, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<4, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<5, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<5, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<6, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<6, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<12, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<12, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<14, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<14, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<15, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<15, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<16, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<16, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<0, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<0, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<1, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<1, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<2, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<2, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<3, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<3, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<4, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<4, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<5, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<5, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<6, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
extern template int launch_fit_nb<6, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
#endif

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
namespace met2 { __attribute__((visibility("hidden"))) void host_release(met2_plan *plan); }
static thread_local std::string g_err;
int met2::abi_fail(int code, const std::string &msg) { g_err = msg; return code; }

// ------------------------------------------------------------------------------------------
// EPG dictionary  (epg/epg.py:64-153).  Lane k holds order k: Fp = F_k (lane 0: F_0),
// Fm = F_-k, Z = Z_k.  One inter-echo period = P T P, P = shift then relax over tau/2.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void epg_dictionary_kernel(int nte, int nt2, int nfa, const double *__restrict__ T2s,
                                                             const double *__restrict__ T1s, double tau,
                                                             const double *__restrict__ alpha_deg, double TR,
                                                             double *__restrict__ D /* [nfa][nte][nt2] */)
{
    const int lane = lane_id();
    const int gw = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (gw >= nfa * nt2) return;
    const int fa = gw / nt2, j = gw - fa * nt2;
    const double rad = M_PI / 180.0;
    const double alpha = alpha_deg[fa] * rad, aexc = alpha_deg[fa] / 2.0 * rad;
    const double th = tau / 2.0;
    const double R2 = 1.0 / T2s[j], R1 = 1.0 / T1s[j];
    const double E2 = exp(-th * R2), E1 = exp(-th * R1);
    const double ca2 = cos(alpha / 2.0), sa2 = sin(alpha / 2.0);
    const double c2 = ca2 * ca2, s2 = sa2 * sa2, sa = sin(alpha), ca = cos(alpha);
    const double scale = 1.0 - exp(-TR / T1s[j]);
    const int n = nte;
    double Fp = (lane == 0) ? sin(aexc) : 0.0;
    double Fm = (lane == 1) ? cos(aexc) : 0.0;
    double Z = 0.0;
    for (int e = 0; e < nte; ++e) {
        for (int half = 0; half < 2; ++half) {
            double up = gather(Fp, (lane + 63) & 63);     // F_{k-1}
            double dn = gather(Fm, (lane + 1) & 63);      // F_-(k+1)
            dn = (lane < n) ? dn : 0.0;                   // F_-n <- 0
            Fp = (lane == 0) ? dn : up;                   // F_0 <- F_-1 ; F_k <- F_{k-1}
            Fm = (lane == 0) ? 0.0 : dn;
            Fp *= E2; Fm *= E2; Z *= E1;
            if (lane > n) { Fp = 0.0; Fm = 0.0; Z = 0.0; }
            if (half == 0 && lane >= 1) {
                double a = Fp, b = Fm, z = Z;
                Fp = c2 * a + s2 * b + sa * z;
                Fm = s2 * a + c2 * b - sa * z;
                Z = -0.5 * sa * a + 0.5 * sa * b + ca * z;
            }
        }
        if (lane == 0) D[((size_t)fa * nte + e) * nt2 + j] = Fp * scale;
    }
}

__global__ __launch_bounds__(256) void gram_kernel(int nte, int nt2, const double *__restrict__ D, double *__restrict__ B,
                                                   double *__restrict__ Dt)
{
    const double *Df = D + (size_t)blockIdx.x * nte * nt2;
    double *Bf = B + (size_t)blockIdx.x * nt2 * nt2;
    double *Dtf = Dt + (size_t)blockIdx.x * nte * nt2;          // [t2][te]: the model signal D x reads columns of D
    for (int idx = threadIdx.x; idx < nt2 * nte; idx += blockDim.x) {
        int a = idx / nte, e = idx - a * nte;
        Dtf[idx] = Df[e * nt2 + a];
    }
    for (int idx = threadIdx.x; idx < nt2 * nt2; idx += blockDim.x) {
        int a = idx / nt2, b = idx - a * nt2;
        double t = 0.0;
        for (int e = 0; e < nte; ++e) t = fma(Df[e * nt2 + a], Df[e * nt2 + b], t);
        Bf[idx] = t;
    }
}

// reference layout [te][t2][fa]  <->  device layout [fa][te][t2]
__global__ void relayout_kernel(int nte, int nt2, int nfa, const double *__restrict__ src, double *__restrict__ dst, int to_device)
{
    size_t total = (size_t)nte * nt2 * nfa;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t f = i / ((size_t)nte * nt2), r = i - f * nte * nt2;     // i indexes [fa][te][t2]
        size_t ref = r * nfa + f;
        if (to_device) dst[i] = src[ref]; else dst[ref] = src[i];
    }
}

// ------------------------------------------------------------------------------------------
// GCV's low-rank basis (objectives.hpp, "Round 4"): per flip angle an orthonormal Q (m x 16) of the dominant column space of D by
// pivoted Gram-Schmidt (the column of largest remaining norm, orthogonalised twice against the vectors found so far), then
// A = Q^T D from the ORIGINAL dictionary, stored transposed ([bin][16]: a support bin's 16 coefficients are one 128-byte line).
// res[fa] = largest remaining column norm / largest column norm: what the basis leaves of D.  One wave per flip angle; the
// remainder R (m x n) lives in LDS, lane <-> column for the sweeps over columns, lane <-> echo for the basis vectors.
// ------------------------------------------------------------------------------------------
#define MET2_GCV_LR_TOL 1e-9
__global__ __launch_bounds__(64) void gcv_basis_kernel(int m, int n, const double *__restrict__ D, double *__restrict__ Aq, double *__restrict__ Qt,
                                                       double *__restrict__ res)
{
    extern __shared__ double basis_lds[];
    double *R = basis_lds, *Q = basis_lds + (size_t)m * n;             // R[m][n], Q[16][64]
    const int fa = blockIdx.x, lane = threadIdx.x;
    const double *Df = D + (size_t)fa * m * n;
    for (int i = lane; i < m * n; i += 64) R[i] = Df[i];
    for (int i = lane; i < MET2_GCV_LR_RANK * 64; i += 64) Q[i] = 0.0;
    __syncthreads();
    auto colnorm2 = [&](int j) { double sum = 0.0; for (int e = 0; e < m; ++e) { const double v = R[e * n + j]; sum = fma(v, v, sum); } return sum; };
    double nrm0 = 0.0;
    for (int s = 0; s < MET2_GCV_LR_RANK; ++s) {
        const double c0 = lane < n ? colnorm2(lane) : -1.0, c1 = lane + 64 < n ? colnorm2(lane + 64) : -1.0;
        const double mx = wave_max(fmax(c0, c1));
        if (s == 0) nrm0 = mx;
        if (!(mx > nrm0 * 1e-30)) break;                                // nothing left above rounding: the dictionary has rank s
        const u64 b0 = ballot(c0 == mx), b1 = ballot(c1 == mx);
        const int piv = b0 ? first_lane(b0) : 64 + first_lane(b1);
        double q = lane < m ? R[lane * n + piv] : 0.0;
        for (int pass = 0; pass < 2; ++pass)
            for (int t = 0; t < s; ++t) { const double qt = Q[t * 64 + lane]; const double d = wave_sum(q * qt); q = fma(-d, qt, q); }
        q = q / sqrt(wave_sum(q * q));
        Q[s * 64 + lane] = q;
        __syncthreads();
        for (int j = lane; j < n; j += 64) {                            // R -= q (q^T R)
            double a = 0.0;
            for (int e = 0; e < m; ++e) a = fma(Q[s * 64 + e], R[e * n + j], a);
            for (int e = 0; e < m; ++e) R[e * n + j] = fma(-a, Q[s * 64 + e], R[e * n + j]);
        }
        __syncthreads();
    }
    {
        const double c0 = lane < n ? colnorm2(lane) : -1.0, c1 = lane + 64 < n ? colnorm2(lane + 64) : -1.0;
        const double mx = wave_max(fmax(c0, c1));
        if (lane == 0) res[fa] = nrm0 > 0.0 ? sqrt(fmax(mx, 0.0) / nrm0) : 0.0;
    }
    for (int j = lane; j < n; j += 64)
        for (int s = 0; s < MET2_GCV_LR_RANK; ++s) {
            double a = 0.0;
            for (int e = 0; e < m; ++e) a = fma(Q[s * 64 + e], Df[e * n + j], a);
            Aq[((size_t)fa * n + j) * MET2_GCV_LR_RANK + s] = a;
        }
    // the basis itself, one vector per row ([fa][16][m]): the FA walk's lower bounds project the voxel's signal on it (fa_kernel)
    for (int s = 0; s < MET2_GCV_LR_RANK; ++s)
        if (lane < m) Qt[((size_t)fa * MET2_GCV_LR_RANK + s) * m + lane] = Q[s * 64 + lane];
}

// ------------------------------------------------------------------------------------------
// voxel classification and counting sort by FA index
// ------------------------------------------------------------------------------------------

// counters and cursors of a fit start at zero: ONE launch instead of two or three fills (a fit over a block of a host pipeline starts while the previous
// block's download bursts, and every tiny launch then waits its turn: three fills stood 1.8 ms in a 5 ms gap between two blocks' solver kernels).
// keep_err: the error word keeps what earlier ENQUEUED fits may have set.
__global__ __launch_bounds__(256) void reset_sort_kernel(SortBufs sb, int nfa, int keep_err)
{
    const int e = 4 * (nfa + 1) + 1;                                  // index of the error word behind hist | cursor | bucket_start | chunk_start | queue
    for (int i = threadIdx.x; i < e + 15; i += blockDim.x)
        if (i != e || !keep_err) sb.hist[i] = 0;
}

#define MET2_SORT_BLOCK 1024
// The histogram and the cursors take ONE global atomic per (workgroup, flip angle): a wave adds its counts to a table in LDS first.  With one atomic
// per wave a single-angle fit sent 3 600 atomics per 230 000 voxels to one address -- 0.33 ms of classify and 0.75 ms of scatter were that queue.
__global__ __launch_bounds__(MET2_SORT_BLOCK) void classify_kernel(int64_t nvox, int nte, int nfa, const double *__restrict__ data,
                                                       int64_t vs, int64_t es,      // element strides of data: voxel, echo
                                                       const double *__restrict__ fa_index, const uint8_t *__restrict__ mask,
                                                       int require_first_echo, SortBufs sb, int32_t *__restrict__ status)
{
    extern __shared__ int sort_lds[];                                 // [nfa] counts of this workgroup
    const int lane = lane_id();
    for (int i = threadIdx.x; i < nfa; i += blockDim.x) sort_lds[i] = 0;
    __syncthreads();
    int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int key = -1, st = 0;
    if (v < nvox) {
        const double *M = data + v * vs;
        double sum = 0.0; bool finite = true;
        for (int e = 0; e < nte; ++e) { double t = M[e * es]; sum += t; finite = finite && isfinite(t); }
        bool mk = mask ? (mask[v] != 0) : true;
        if (!finite) st = MET2_ST_NONFINITE;
        else if (mk && sum > 0.0 && (!require_first_echo || M[0] > 0.0)) {
            int fi = fa_index ? (int)fa_index[v] : 0;
            if (fi < 0 || fi >= nfa) atomicOr(sb.err, 1);
            else { key = fi; st = MET2_ST_FITTED; }
        }
        sb.key[v] = key;
        if (status) status[v] = st;
    }
    // wave-aggregated histogram, into the workgroup's table
    u64 todo = ballot(key >= 0);
    while (todo) {
        int leader = first_lane(todo);
        int k0 = bcast_i(key, leader);
        u64 same = ballot(key == k0) & todo;
        if (lane == leader) atomicAdd(&sort_lds[k0], __popcll(same));
        todo &= ~same;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nfa; i += blockDim.x) { const int c = sort_lds[i]; if (c) atomicAdd(&sb.hist[i], c); }
}

// bucket and chunk offsets from the histogram.  The histogram comes in and the offsets go out in parallel; the running sums are taken over LDS (one
// thread walking 273 counters in global memory stood 0.13 ms per launch, two launches per fit).
__global__ __launch_bounds__(256) void scan_kernel(int nfa, int chunk, SortBufs sb)
{
    extern __shared__ int sort_lds[];                                 // [nfa + 1] bucket starts, [nfa + 1] chunk starts
    int *bs = sort_lds, *cs = sort_lds + nfa + 1;
    for (int f = threadIdx.x; f < nfa; f += blockDim.x) { const int c = sb.hist[f]; bs[f] = c; cs[f] = (c + chunk - 1) / chunk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0, cacc = 0;
        for (int f = 0; f < nfa; ++f) { const int c = bs[f], cc = cs[f]; bs[f] = acc; cs[f] = cacc; acc += c; cacc += cc; }
        bs[nfa] = acc; cs[nfa] = cacc;
        sb.queue[0] = 0;
    }
    __syncthreads();
    for (int f = threadIdx.x; f <= nfa; f += blockDim.x) {
        sb.bucket_start[f] = bs[f]; sb.chunk_start[f] = cs[f];
        if (f < nfa) sb.cursor[f] = 0;
    }
}

__global__ __launch_bounds__(MET2_SORT_BLOCK) void scatter_kernel(int64_t nvox, int nfa, SortBufs sb)
{
    extern __shared__ int sort_lds[];                                 // [nfa] counts of this workgroup, then [nfa] its base in every bucket
    const int lane = lane_id();
    for (int i = threadIdx.x; i < nfa; i += blockDim.x) sort_lds[i] = 0;
    __syncthreads();
    int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int key = (v < nvox) ? sb.key[v] : -1;
    int off = 0;                                                      // this voxel's place among the workgroup's voxels of its flip angle
    u64 todo = ballot(key >= 0);
    while (todo) {
        int leader = first_lane(todo);
        int k0 = bcast_i(key, leader);
        u64 same = ballot(key == k0) & todo;
        int base = 0;
        if (lane == leader) base = atomicAdd(&sort_lds[k0], __popcll(same));
        base = bcast_i(base, leader);
        if (key == k0) off = base + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nfa; i += blockDim.x) { const int c = sort_lds[i]; if (c) sort_lds[nfa + i] = atomicAdd(&sb.cursor[i], c); }
    __syncthreads();
    if (key >= 0) sb.perm[sb.bucket_start[key] + sort_lds[nfa + key] + off] = (int)v;
}

// second pass of the capacity scheme: voxels whose passive set hit the fast path's kmax are queued again
__global__ __launch_bounds__(256) void requeue_overflow_kernel(int64_t nvox, const double *__restrict__ fa_index,
                                                               const int32_t *__restrict__ status, SortBufs sb)
{
    const int lane = lane_id();
    int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int key = -1;
    if (v < nvox) {
        if (status[v] & MET2_ST_KOVERFLOW) key = fa_index ? (int)fa_index[v] : 0;
        sb.key[v] = key;
    }
    u64 todo = ballot(key >= 0);
    while (todo) {
        int leader = first_lane(todo);
        int k0 = bcast_i(key, leader);
        u64 same = ballot(key == k0) & todo;
        if (lane == leader) atomicAdd(&sb.hist[k0], __popcll(same));
        todo &= ~same;
    }
}

// ------------------------------------------------------------------------------------------
// brute-force flip-angle estimation (flip_angle_algorithms/fa_estimation.py:74-111):
// for every flip angle of the dictionary one plain NNLS per voxel, argmin of the residual norm.
// Every wave keeps VPW voxels in registers and walks the FA axis on its own.
// ------------------------------------------------------------------------------------------
struct FaArgs {
    int n, m, nfa, kmax, waves, wave_doubles;
    const double *Dfa, *Bfa, *Dtfa;
    const double *Kd;         // [n][n], only multiplied by lambda = 0 here (the refactorisation reads its rows)
    const double *data;       // echo e of voxel v at data[v * vs + e * es]
    int64_t vs, es;
    const uint8_t *mask;
    double *fa_index, *km, *resid;
    int *queue;
    int64_t nvox;
    const double *Hq;         // [nvox of this pass][nfa * 16]: c_fa = Q_fa^T b in every flip angle's low-rank basis (fa_project_kernel on the stacked bases), or NULL: no pruning.
    double *Lb;               // [nvox of this pass][nfa]: the walk parks every angle's lower bound here (a voxel's row belongs to the one wave that walks it)
    const double *Aq;         // [nfa][n][16]: (Q_fa^T D_fa)^T -- with Hq, the walk forms h_fa = A_fa^T c_fa = D_fa^T (Q Q^T b) from it (round 5: no H then)
    const double *H;          // [nvox of this pass][nfa * n]: h = D_fa^T b of every flip angle (fa_project_kernel), or NULL: formed in the walk
    int64_t v0;               // first voxel of this pass (H row 0), voxels [v0, v0 + nvox_pass)
    int64_t v_end;            // one past the last voxel of this pass
};

// ------------------------------------------------------------------------------------------
// The batched contraction of the brute-force FA step on the matrix cores: every NNLS of fa_estimation.py:78-82 starts from
// h_fa = D_fa^T b, 91 dictionaries per voxel.  As ONE GEMM per pass of voxels,  H[v][fa * n + bin] = sum_e data[v][e] D^T[fa][bin][e]
// ([T x m] . [m x nfa n]) with v_mfma_f64_16x16x4: a wave keeps the echoes of 32 voxels in registers (two 16-voxel operand sets),
// walks the nfa * n rows of the stacked D^T in tiles of 16 -- each tile's operand is loaded once from L2 and used for both voxel
// sets -- and writes the 16 x 16 results voxel-major, so that the walk reads its h as one contiguous row per flip angle.
// The walk without it re-derives h per (voxel pair, flip angle) on the vector unit from D rows streamed through L2 (half of the
// 2.8 MB per voxel the FA walk pulls through L2 at 48 x 120).
// KS = ceil(m / 4) k-steps of four echoes (operands past m are zero).
// ------------------------------------------------------------------------------------------
struct FaGemmArgs {
    int n, m, nfa;
    const double *Dtfa;       // [nfa * n][m]
    const double *data; int64_t vs, es;
    double *H;                // [T][nfa * n]
    int64_t v0, v_end;
};

template <int KS>
__global__ __launch_bounds__(256) void fa_project_kernel(FaGemmArgs A)
{
    const int lane = lane_id(), li = lane & 15, lk = lane >> 4;
    const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t vA = A.v0 + wv * 32;
    if (vA >= A.v_end) return;
    const int M = A.nfa * A.n, m = A.m;
    // A operands: voxel vA + li (and vA + 16 + li), echoes 4 q + lk
    double a0[KS], a1[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) {
        const int e = 4 * q + lk;
        const int64_t v0 = vA + li, v1 = vA + 16 + li;
        a0[q] = (e < m && v0 < A.v_end) ? A.data[v0 * A.vs + e * A.es] : 0.0;
        a1[q] = (e < m && v1 < A.v_end) ? A.data[v1 * A.vs + e * A.es] : 0.0;
    }
    double *Hrow0[4], *Hrow1[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        Hrow0[v] = A.H + (size_t)(vA - A.v0 + lk + 4 * v) * M;
        Hrow1[v] = A.H + (size_t)(vA - A.v0 + 16 + lk + 4 * v) * M;
    }
    for (int r0 = 0; r0 < M; r0 += 16) {
        const int r = min(r0 + li, M - 1);
        const double *drow = A.Dtfa + (size_t)r * m;
        double bop[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) { const int e = 4 * q + lk; bop[q] = (e < m) ? drow[min(e, m - 1)] : 0.0; }
        met2_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], bop[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], bop[q], acc1, 0, 0, 0);
        }
        if (r0 + li < M) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (vA + lk + 4 * v < A.v_end) Hrow0[v][r0 + li] = acc0[v];
                if (vA + 16 + lk + 4 * v < A.v_end) Hrow1[v][r0 + li] = acc1[v];
            }
        }
    }
}

// PRUNE: the walk skips flip angles by their lower bounds (A.Hq); a separate instantiation keeps the exhaustive walk (all residuals wanted: the
// spline path; few flip angles) free of the masks and the extra round
template <int VPW, int NB, int WAVES, bool PRUNE>
__global__ __launch_bounds__(64 * WAVES) void fa_kernel(FaArgs A)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int n = A.n, m = A.m;
    double *sR = smem + (size_t)wave * A.wave_doubles;
    WaveShared S;
    S.R = sR; S.n = n; S.m = m; S.kmax = A.kmax; S.rcap = A.wave_doubles; S.K = A.Kd; S.kband = nullptr; S.Dt = nullptr; S.DtG = A.Dtfa; S.dtstride = m; S.buffer_rows = false; S.reorder = false; S.have_bdiag = true; S.bdiag[0] = S.bdiag[1] = 0.0;
    S.B = A.Bfa; S.D = A.Dfa; S.bstride = n; S.dstride = n;
    Band<NB> bd;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int d = 0; d < 5; ++d) bd.lb[b][d] = 0.0;
    // every wave pulls its own VPW voxels and walks the flip angles alone, reading D, D^T and B through L1/L2 -- no barrier,
    // so no wave waits for the slowest solve of a tile (a tile kernel with D and B staged in LDS per flip angle spent 48 % of
    // its wave time in its two barriers per flip angle; removed in round 3).
    const int64_t ntiles = (A.v_end - A.v0 + VPW - 1) / VPW;     // this pass: voxels [A.v0, A.v_end)
    const int Mh = A.nfa * n;
    for (int64_t round = 0; round <= ntiles; ++round) {
        int t32 = 0;
        if (lane == 0) t32 = atomicAdd(A.queue, 1);
        const int64_t tile = __builtin_amdgcn_readfirstlane(t32);
        if (tile >= ntiles) break;
        const int64_t v0 = A.v0 + tile * VPW;
        double b[VPW], best_r[VPW], best_km[VPW];
        int best_fa[VPW];
        bool act[VPW];
        // the passive set and x of a voxel carry over from one flip angle to the next (neighbouring dictionaries give
        // nearly the same support): each solve is a warm start -- re-factorise, secondary loop, dual check -- instead
        // of rebuilding the support bin by bin.  NNLS has one minimiser, so only the rounding differs from a cold start.
        NnlsState<NB> st[VPW];
#pragma unroll
        for (int vv = 0; vv < VPW; ++vv) {
            st[vv].itmax_hit = 0;
            MET2_CYC_INIT(st[vv]);
            nnls_reset<NB>(st[vv]);
            const int64_t v = v0 + vv;
            const bool in = v < A.v_end;
            b[vv] = (in && lane < m) ? A.data[v * A.vs + lane * A.es] : 0.0;
            double sum = wave_sum(b[vv]);
            bool mk = in && (A.mask ? (A.mask[v] != 0) : true);
            bool fin = (ballot(!isfinite(b[vv])) == 0ull);
            act[vv] = mk && fin && (sum > 0.0);
            best_r[vv] = INFINITY; best_km[vv] = 0.0; best_fa[vv] = 0;
        }
        // Lower bounds (A.Hq): the NNLS residual of a flip angle is at least the unconstrained least-squares residual on its dictionary's
        // column space, ||b||^2 - ||Q_fa^T b||^2 in the plan's low-rank basis (exact to 1e-18 ||b||^2 while the basis leaves < 1e-9 of D).
        // The angle with the smallest bound is solved first (round -1); in rounds 0 .. nfa - 1 an angle whose bound exceeds the best
        // residual found so far cannot be the argmin and is skipped -- on the reference's recipe 28 of 91 angles per voxel are solved
        // (median 24).  Lane f (and f + 64) forms the bound of angle f and parks it in the voxel's own row of Hq (no registers held
        // through the walk).
        double slack[VPW];
        int fa0[VPW];
        u64 cand[VPW][2];                      // flip angles whose bound does not exceed the best residual so far (wave-uniform masks; all ones: no pruning)
        // the masks are formed again whenever the best residual improves: lane f compares its angle's parked bound
        auto candidates = [&](int vv) {
            const double tb = fma(best_r[vv], 1.0 + 1e-6, slack[vv]);     // sqrt(lb) > best (1 + 1e-6) + 1e-7 ||b||, compared as squares
            const double thr = tb * tb;
            const double *q = A.Lb + (size_t)(v0 + vv - A.v0) * (size_t)A.nfa;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int f = lane + 64 * sl;
                const double lb = q[min(f, A.nfa - 1)];
                cand[vv][sl] = ballot(f < A.nfa && !(lb > thr));
            }
        };
#pragma unroll
        for (int vv = 0; vv < VPW; ++vv) {
            slack[vv] = 0.0; fa0[vv] = -1; cand[vv][0] = cand[vv][1] = ~0ull;
            if (PRUNE && A.Hq && act[vv]) {
                const double bb2 = wave_sum(b[vv] * b[vv]);
                const double *q = A.Hq + (size_t)(v0 + vv - A.v0) * ((size_t)A.nfa * MET2_GCV_LR_RANK);
                double *qb = A.Lb + (size_t)(v0 + vv - A.v0) * (size_t)A.nfa;
                double l2[2];
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int f = lane + 64 * sl;
                    double cs = 0.0;
                    if (f < A.nfa)
                        for (int j = 0; j < MET2_GCV_LR_RANK; ++j) { const double c = q[(size_t)f * MET2_GCV_LR_RANK + j]; cs = fma(c, c, cs); }
                    l2[sl] = (f < A.nfa) ? bb2 - cs : INFINITY;
                    if (f < A.nfa) qb[f] = l2[sl];
                }
                // What the bound may be off by: the basis leaves E = (I - Q Q^T) D, ||E|| <= 1e-9 of the largest column, outside, and the residual
                // of ANY x is at least ||b_perp|| - ||E x||; with ||x||_1 <= a few ||b|| per unit column norm (x >= 0 and a dictionary of decaying
                // signals: the first echo alone bounds x) that is < 1e-8 ||b||.  1e-7 ||b|| covers it sixteen times over and the rounding of
                // the difference above (1e-16 ||b||^2, i.e. 1e-8 ||b|| at worst) as well; on noisy data it prunes nothing less.
                slack[vv] = 1e-7 * sqrt(bb2);
                const double mn = wave_min(fmin(l2[0], l2[1]));
                const u64 m0 = ballot(l2[0] == mn), m1 = ballot(l2[1] == mn);
                fa0[vv] = m0 ? first_lane(m0) : 64 + first_lane(m1);
            }
        }
        for (int i = (PRUNE && A.Hq) ? -1 : 0; i < A.nfa; ++i) {
#ifdef MET2_CYCSTATS
            const unsigned long long cv0 = __builtin_readcyclecounter();
#endif
            if (!A.H && !(PRUNE && A.Hq)) {    // few flip angles: no batched contraction; rows of D come from L2, each loaded once for all the wave's voxels
                const double *Bf = A.Bfa + (size_t)i * n * n;
                S.B = Bf; S.D = A.Dfa + (size_t)i * m * n; S.Dt = A.Dtfa + (size_t)i * m * n;
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) { const int j = min(lane + 64 * bb, n - 1); S.bdiag[bb] = Bf[(size_t)j * n + j]; }
                double hh[VPW][NB];
                project_multi<NB, VPW>(S, b, lane, hh);
#pragma unroll
                for (int vv = 0; vv < VPW; ++vv)
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) st[vv].h[bb] = hh[vv][bb];
            }
#pragma unroll
            for (int vv = 0; vv < VPW; ++vv) {
                const int fa = (PRUNE && i < 0) ? fa0[vv] : i;
                if (!act[vv] || (PRUNE && (fa < 0 || (i >= 0 && fa == fa0[vv])))) continue;
                if (PRUNE && !((cand[vv][fa >> 6] >> (fa & 63)) & 1ull)) continue;   // its lower bound exceeds the best residual: cannot be the argmin
                if (A.H || (PRUNE && A.Hq)) {
                    const double *Bf = A.Bfa + (size_t)fa * n * n;
                    S.B = Bf; S.D = A.Dfa + (size_t)fa * m * n; S.Dt = A.Dtfa + (size_t)fa * m * n;
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) { const int j = min(lane + 64 * bb, n - 1); S.bdiag[bb] = Bf[(size_t)j * n + j]; }
                    if (PRUNE && A.Hq) {
                        // h = A_fa^T c_fa from the plan's low-rank basis: D_fa = Q_fa A_fa to 1e-10 of its largest column, so this is D_fa^T (Q Q^T b), and the
                        // NNLS against the projected data has the minimiser of the NNLS against b (b - Q Q^T b is orthogonal to everything D_fa x can
                        // reach): residuals equal to 1e-14 relative (numpy, both shapes).  16 FMAs per bin from a 128-byte row of a 1.4 MB table that
                        // stays in L2, where the batched contraction wrote 87 KB of H per voxel at 91 x 120 and the walk read a third of it back.
                        const double *cq = A.Hq + (size_t)(v0 + vv - A.v0) * ((size_t)A.nfa * MET2_GCV_LR_RANK) + (size_t)fa * MET2_GCV_LR_RANK;
                        const double cl = cq[lane & (MET2_GCV_LR_RANK - 1)];
                        const met2_d2 *aq = (const met2_d2 *)(A.Aq + (size_t)fa * n * MET2_GCV_LR_RANK);
#pragma unroll
                        for (int bb = 0; bb < NB; ++bb) {
                            const int j = min(lane + 64 * bb, n - 1);
                            met2_d2 av[MET2_GCV_LR_RANK / 2];
#pragma unroll
                            for (int t = 0; t < MET2_GCV_LR_RANK / 2; ++t) av[t] = aq[(size_t)j * (MET2_GCV_LR_RANK / 2) + t];
                            double h0 = 0.0, h1 = 0.0;
#pragma unroll
                            for (int t = 0; t < MET2_GCV_LR_RANK / 2; ++t) { h0 = fma(av[t].x, bcast(cl, 2 * t), h0); h1 = fma(av[t].y, bcast(cl, 2 * t + 1), h1); }
                            st[vv].h[bb] = (lane + 64 * bb < n) ? h0 + h1 : 0.0;
                        }
                    } else {                   // h of this flip angle from the batched MFMA contraction: one contiguous row per voxel
                        const double *hrow = A.H + (size_t)(v0 + vv - A.v0) * Mh + (size_t)fa * n;
#pragma unroll
                        for (int bb = 0; bb < NB; ++bb) { const int j = lane + 64 * bb; const double hv = hrow[min(j, n - 1)]; st[vv].h[bb] = (j < n) ? hv : 0.0; }
                    }
                }
                nnls_solve_warm<NB, (NB == 2 && VPW == 1)>(S, bd, st[vv], 0.0, false, lane);   // one position slot (k <= nTE); with two voxels per wave the second code path stops the voxel loop from unrolling
                const double rn = sqrt(sse_of<NB>(S, st[vv], b[vv], lane));
                if (A.resid && lane == 0) A.resid[(size_t)(v0 + vv) * A.nfa + fa] = rn;
                if (rn < best_r[vv] || (rn == best_r[vv] && fa < best_fa[vv])) {      // np.argmin: the first minimum wins, whatever order the angles are visited in
                    best_r[vv] = rn; best_fa[vv] = fa;
                    double t = 0.0;
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) t += (lane + 64 * bb < n) ? st[vv].x[bb] : 0.0;
                    best_km[vv] = wave_sum(t);
                    if (PRUNE && A.Hq) candidates(vv);
                }
            }
#ifdef MET2_CYCSTATS
            st[0].cyc[0] += __builtin_readcyclecounter() - cv0;
#endif
        }
#ifdef MET2_CYCSTATS
#pragma unroll
        for (int vv = 0; vv < VPW; ++vv) MET2_CYC_FLUSH(st[vv]);
#endif
#pragma unroll
        for (int vv = 0; vv < VPW; ++vv) {
            const int64_t v = v0 + vv;
            if (v < A.v_end && lane == 0) {
                A.fa_index[v] = act[vv] ? (double)best_fa[vv] : 0.0;
                if (A.km) A.km[v] = act[vv] ? best_km[vv] : 0.0;
            }
            if (v < A.v_end && A.resid && !act[vv]) for (int f = lane; f < A.nfa; f += 64) A.resid[(size_t)v * A.nfa + f] = 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// spline flip-angle selection (flip_angle_algorithms/fa_estimation.py:54-59): cubic (not-a-knot)
// interpolation of the coarse-grid NNLS residuals, bounded Brent over [90, 180] (scipy
// minimize_scalar(method='Bounded'): xatol 1e-5, maxiter 500), snap to the fine FA grid.
// One thread per voxel; the residuals come from fa_kernel on the coarse dictionary.
// ------------------------------------------------------------------------------------------
// NESMA filter (motor:305-333): every voxel with mask == 1 becomes the mean of the voxels of its
// [x-6, x+6) x [y-6, y+6) x [z-6, z+6) window (clipped to the volume) whose relative L1 distance
// 100 * sum|s_nb - s_c| / sum(s_c) is below 2.5 %.  One wave per output voxel, lane <-> echo; the
// window is walked in batches of 8 neighbours: |s_nb - s_c| goes through the wave's LDS strip so
// that lane group q (8 lanes) can sum neighbour q in the order of numpy's pairwise sum (8 running
// sums, tree combine, tail in order), which keeps the 2.5 % test and the running mean bit-identical
// to the reference's np.sum / np.mean.  The four waves of a workgroup take consecutive z, so their
// windows overlap by 11/12 and the re-reads hit L1/L2.
// ------------------------------------------------------------------------------------------
#define MET2_NESMA_HW 6
#define MET2_NESMA_MAX_NT 128
struct NesmaArgs {
    int nx, ny, nz, nt;
    const double *data;       // [nx][ny][nz][nt]
    const uint8_t *mask;      // [nx][ny][nz], filtered where == 1 (NULL = everywhere)
    double *out;              // [nx][ny][nz][nt]
    int64_t nvox;
    int srow;                 // LDS row stride in doubles: >= nt, = 8 mod 32
};

// numpy pairwise sum of S[q][0..nt) for the lane's group q = lane / 8 (nt <= 128): valid on every lane of the group
__device__ __forceinline__ double np_rowsum_group(const double *Sq, int nt, int j)
{
    double r;
    if (nt < 8) {
        r = 0.0;
        for (int i = 0; i < nt; ++i) r += Sq[i];
        return r;
    }
    const int n8 = nt - (nt & 7);
    r = Sq[j];
    for (int i = 8; i < n8; i += 8) r += Sq[i + j];
    r = r + dpp_mov<0xB1>(r);       // r0+r1 | r2+r3 | ...
    r = r + dpp_mov<0x4E>(r);       // (r0+r1)+(r2+r3) | (r4+r5)+(r6+r7)
    r = r + dpp_mov<0x141>(r);      // both halves of the group
    for (int i = n8; i < nt; ++i) r += Sq[i];
    return r;
}

// RE < 2.5 with RE = fl(fl(100 S) / sumc), decided without the division whenever 100 S is clear of 2.5 sumc by
// more than the rounding of both sides (the exact quotient is only formed for the voxels inside that band)
struct NesmaTest {
    double sumc, lo, hi;
    bool fast;
    __device__ __forceinline__ void init(double sc)
    {
        sumc = sc; fast = sc > 0.0 && sc < 1e300;
        lo = 2.5 * sc * (1.0 - 1e-15); hi = 2.5 * sc * (1.0 + 1e-15);
    }
    __device__ __forceinline__ u64 similar(double S) const
    {
        const double t = 100.0 * S;
        const bool yes = t < lo, no = !(t <= hi);
        if (fast && ballot(!yes && !no) == 0ull) return ballot(yes);
        return ballot(t / sumc < 2.5);
    }
};

// the value the lane 32 above holds (v_permlane32_swap: upper half of one operand <-> lower half of the other)
__device__ __forceinline__ double from_upper_half(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}

// One batch = the z-run of one (x, y) of the window: up to 12 neighbours that are contiguous in memory, so the
// window walk is two nested loops with a pointer bump and the batch geometry (run length L) is fixed per voxel.
// PACK = 1: lane <-> echo (NE echoes per lane), one neighbour per load.  PACK = 2 (nt <= 32): the two halves of the
// wave load two consecutive neighbours at once; the running sum lives in the lower half and takes the odd neighbours
// through v_permlane32_swap so that the additions keep the reference's order.  LDS rows are padded to a stride of
// 8 mod 32 doubles, which spreads the 8 lane groups of a row-sum read over all banks.
// NT > 0 fixes the echo count at compile time (row sums fully unrolled); NT = 0 reads it from the arguments.
template <int NE, int PACK, int NT>
__global__ __launch_bounds__(256) void nesma_kernel(NesmaArgs A)
{
    extern __shared__ double nesma_lds[];
    constexpr int STEPS = 2 * MET2_NESMA_HW / PACK;
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const int nt = NT > 0 ? NT : A.nt, SR = NT > 0 ? ((NT + 23) / 32) * 32 + 8 : A.srow;
    double *S = nesma_lds + (size_t)w * 2 * MET2_NESMA_HW * SR;
    // workgroups are dealt round-robin to the 8 XCDs: give XCD i the i-th contiguous eighth of the volume so that
    // the windows its CUs walk at the same time overlap in its own L2
    const unsigned nb = gridDim.x, per = (nb + 7u) / 8u;
    const unsigned bid = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const int64_t v = (int64_t)bid * 4 + w;
    if (bid >= nb || v >= A.nvox) return;                      // wave-uniform
    const int h = PACK == 2 ? lane >> 5 : 0;
    const int e0 = PACK == 2 ? (lane & 31) : lane, e1 = lane + 64;
    const bool has0 = e0 < nt, has1 = NE > 1 && e1 < nt;
    const bool writer = h == 0;
    double *orow = A.out + v * nt;
    if (A.mask && A.mask[v] != 1) {
        if (has0 && writer) orow[e0] = 0.0;
        if (has1) orow[e1] = 0.0;
        return;
    }
    const int z = (int)(v % A.nz), y = (int)((v / A.nz) % A.ny), x = (int)(v / ((int64_t)A.nz * A.ny));
    const int x0 = max(x - MET2_NESMA_HW, 0), x1 = min(x + MET2_NESMA_HW, A.nx);
    const int y0 = max(y - MET2_NESMA_HW, 0), y1 = min(y + MET2_NESMA_HW, A.ny);
    const int z0 = max(z - MET2_NESMA_HW, 0), z1 = min(z + MET2_NESMA_HW, A.nz);
    const int L = z1 - z0;                                     // 1..12 neighbours per batch
    const int q = lane >> 3, j = lane & 7;
    const double *crow = A.data + v * nt;
    const double c0 = has0 ? crow[e0] : 0.0, c1 = has1 ? crow[e1] : 0.0;
    if (has0 && writer) S[e0] = c0;
    if (has1) S[e1] = c1;
    __builtin_amdgcn_wave_barrier();
    NesmaTest T;
    T.init(bcast(np_rowsum_group(S, nt, j), 0));
    __builtin_amdgcn_wave_barrier();
    const double *S0 = S + min(q, L - 1) * SR, *S1 = S + min(8 + (q & 3), L - 1) * SR;
    const int64_t ystride = (int64_t)A.nz * nt, xstride = ystride * A.ny;
    unsigned off0[STEPS], off1[STEPS];                         // the lane's element of each row of a run
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
        const int r = min(PACK * t + h, L - 1);
        off0[t] = (unsigned)(r * nt + min(e0, nt - 1));
        off1[t] = (unsigned)(r * nt + min(e1, nt - 1));
    }
    u64 live0 = 0, live1 = 0;                                  // bit 8 r set for the rows a run has
    for (int r = 0; r < L; ++r) { if (r < 8) live0 |= 1ull << (8 * r); else live1 |= 1ull << (8 * (r - 8)); }
    double acc0 = 0.0, acc1 = 0.0;
    int cnt = 0;
    for (int i = x0; i < x1; ++i) {
        const double *strip = A.data + i * xstride + y0 * ystride + (int64_t)z0 * nt;
        for (int jj = y0; jj < y1; ++jj, strip += ystride) {
            // all loads of the run first (rows past the run re-read its last row; they are never summed), then the
            // |difference| rows into LDS
            double nb0[STEPS], nb1[STEPS];
#pragma unroll
            for (int t = 0; t < STEPS; ++t) {
                nb0[t] = strip[off0[t]];
                nb1[t] = NE > 1 ? strip[off1[t]] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < STEPS; ++t) {
                const int r = PACK * t + h;
                if (has0) S[r * SR + e0] = fabs(nb0[t] - c0);
                if (NE > 1 && has1) S[r * SR + e1] = fabs(nb1[t] - c1);
            }
            __builtin_amdgcn_wave_barrier();
            const u64 ok0 = T.similar(np_rowsum_group(S0, nt, j));           // 8 identical bits per lane group
            const u64 ok1 = L > 8 ? T.similar(np_rowsum_group(S1, nt, j)) : 0ull;
            __builtin_amdgcn_wave_barrier();
            const u64 v0 = ok0 & live0, v1 = ok1 & live1;                    // bit 8 r: neighbour r (8 + r) of the run is similar
            cnt += __builtin_popcountll(v0) + __builtin_popcountll(v1);
#pragma unroll
            for (int t = 0; t < STEPS; ++t) {
                if (PACK == 1) {
                    if (((t < 8 ? v0 : v1) >> (8 * (t & 7))) & 1ull) { acc0 += nb0[t]; if (NE > 1) acc1 += nb1[t]; }
                } else {
                    const int ra = 2 * t, rb = 2 * t + 1;
                    if (((ra < 8 ? v0 : v1) >> (8 * (ra & 7))) & 1ull) acc0 += nb0[t];
                    if (((rb < 8 ? v0 : v1) >> (8 * (rb & 7))) & 1ull) acc0 += from_upper_half(nb0[t]);
                }
            }
        }
    }
    const double dc = (double)cnt;                             // empty set -> 0/0 = nan, as np.mean of nothing
    if (has0 && writer) orow[e0] = acc0 / dc;
    if (has1) orow[e1] = acc1 / dc;
}

// ------------------------------------------------------------------------------------------
// Gaussian pre-smoothing of the FA step (motor:337-343: scipy.ndimage.gaussian_filter(vol, 2.0) on every echo volume):
// one separable pass per axis over the whole [nx][ny][nz][nt] array (nt innermost: all echoes at once, coalesced for
// every axis), 'reflect' boundary, accumulation in the order of scipy's correlate1d for symmetric kernels (centre, then
// the pairs from the outermost inwards) with separate multiply and add roundings, so the result is bit-identical.
// ------------------------------------------------------------------------------------------
#define MET2_SMOOTH_MAX_RADIUS 32
struct SmoothArgs {
    int n, radius;
    int64_t stride, total;
    const double *src;
    double *dst;
    double w[2 * MET2_SMOOTH_MAX_RADIUS + 1];
};

__device__ __forceinline__ int reflect_index_dev(int i, int n)
{
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period; if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

__global__ __launch_bounds__(256) void smooth_axis_kernel(SmoothArgs A)
{
#pragma clang fp contract(off)                                      // scipy's C loop rounds the product and the sum separately
    const int r = A.radius, n = A.n;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < A.total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)((idx / A.stride) % n);
        const double *base = A.src + (idx - (int64_t)i * A.stride);
        double t = A.src[idx] * A.w[r];
        if (i >= r && i + r < n) {                                  // interior: no reflection
            for (int j = -r; j < 0; ++j) {
                const double pr = (base[(int64_t)(i + j) * A.stride] + base[(int64_t)(i - j) * A.stride]) * A.w[j + r];
                t = t + pr;
            }
        } else {
            for (int j = -r; j < 0; ++j) {
                const double pr = (base[(int64_t)reflect_index_dev(i + j, n) * A.stride] + base[(int64_t)reflect_index_dev(i - j, n) * A.stride]) * A.w[j + r];
                t = t + pr;
            }
        }
        A.dst[idx] = t;
    }
}

// ------------------------------------------------------------------------------------------
#define MET2_MAX_LR 32
struct SplineArgs {
    int nlr, nhr, nte;
    const double *alpha_lr;   // [nlr] device
    const double *W;          // [nlr][nlr] device: knot slopes = W y
    const double *alpha_hr;   // [nhr] device
    const double *resid;      // [nvox][nlr]
    const double *data;       // echo e of voxel v at data[v * vs + e * es]
    int64_t vs, es;
    const uint8_t *mask;
    double *fa_index, *xmin;
    int64_t nvox;
};

__device__ __forceinline__ double spline_eval_dev(int n, const double *x, const double *y, const double *s, double xx)
{
    int i = 0;
    while (i < n - 2 && xx >= x[i + 1]) ++i;
    const double h = x[i + 1] - x[i], t = (xx - x[i]) / h;
    const double h00 = (1.0 + 2.0 * t) * (1.0 - t) * (1.0 - t), h10 = t * (1.0 - t) * (1.0 - t);
    const double h01 = t * t * (3.0 - 2.0 * t), h11 = t * t * (t - 1.0);
    return h00 * y[i] + h10 * h * s[i] + h01 * y[i + 1] + h11 * h * s[i + 1];
}

__global__ __launch_bounds__(128) void fa_spline_kernel(SplineArgs A)
{
    __shared__ double sx[MET2_MAX_LR];
    for (int i = threadIdx.x; i < A.nlr; i += blockDim.x) sx[i] = A.alpha_lr[i];
    __syncthreads();
    const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (v >= A.nvox) return;
    double sum = 0.0;
    for (int e = 0; e < A.nte; ++e) sum += A.data[v * A.vs + e * A.es];
    const bool mk = A.mask ? (A.mask[v] != 0) : true;
    if (!(mk && sum > 0.0)) { A.fa_index[v] = 0.0; if (A.xmin) A.xmin[v] = 0.0; return; }
    double y[MET2_MAX_LR], s[MET2_MAX_LR];
    const int n = A.nlr;
    for (int i = 0; i < n; ++i) y[i] = A.resid[(size_t)v * n + i];
    for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int j = 0; j < n; ++j) t = fma(A.W[i * n + j], y[j], t);
        s[i] = t;
    }
    int flag;
    const double xs = fminbound_dev<false>([&](double x) { return spline_eval_dev(n, sx, y, s, x); }, []() {}, 90.0, 180.0, 1e-5, 500, flag);
    int best = 0; double dbest = fabs(A.alpha_hr[0] - xs);            // np.argmin(|alpha - x|): first minimum
    for (int a = 1; a < A.nhr; ++a) { double d = fabs(A.alpha_hr[a] - xs); if (d < dbest) { dbest = d; best = a; } }
    A.fa_index[v] = (double)best;
    if (A.xmin) A.xmin[v] = xs;
}

// zeros for gated-out voxels (motor:115-117) and the metrics an all-zero spectrum gets at
// motor:448-468 when mask > 0 (T2_M = T2_IE = exp(0) = 1, TWC = 1e-16)
__global__ __launch_bounds__(256) void finalize_unfitted_kernel(int64_t nvox, int n, int m, const int *__restrict__ key,
                                                                const uint8_t *__restrict__ mask, double *__restrict__ fsol,
                                                                double *__restrict__ sig, double *__restrict__ reg,
                                                                double *__restrict__ lam, double *__restrict__ maps)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < nvox * n; i += stride) if (key[i / n] < 0) fsol[i] = 0.0;
    if (sig) for (int64_t i = t0; i < nvox * m; i += stride) if (key[i / m] < 0) sig[i] = 0.0;
    for (int64_t v = t0; v < nvox; v += stride) {
        if (key[v] >= 0) continue;
        reg[v] = 0.0;
        if (lam) lam[v] = 0.0;
        if (maps) {
            bool mk = mask ? (mask[v] != 0) : true;
            maps[0 * nvox + v] = 0.0; maps[1 * nvox + v] = 0.0; maps[2 * nvox + v] = 0.0;
            maps[3 * nvox + v] = mk ? 1.0 : 0.0; maps[4 * nvox + v] = mk ? 1.0 : 0.0;
            maps[5 * nvox + v] = mk ? 1.0e-16 : 0.0;
        }
    }
}

// motor:443-472 standalone: one wavefront per voxel
template <int NB>
__global__ __launch_bounds__(256) void metrics_kernel(int64_t nvox, int n, const double *__restrict__ t2s, double cut_m, double cut_ie,
                                                      const double *__restrict__ fsol, const uint8_t *__restrict__ mask,
                                                      double *__restrict__ maps)
{
    const int lane = lane_id();
    MetricLanes<NB> ml;
    metric_lanes<NB>(ml, t2s, n, cut_m, cut_ie, lane);
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; v < nvox; v += nw) {
        const bool mk = mask ? (mask[v] != 0) : true;
        double xs[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) xs[b] = (lane + 64 * b < n && mk) ? fsol[(size_t)v * n + lane + 64 * b] : 0.0;
        write_metrics<NB>(ml, xs, mk, maps, nvox, v, lane);
    }
}


// ------------------------------------------------------------------------------------------
// ROI mode (motor/motor_recon_met2_real_data_ROI.py:405-420): per ROI the mean signal over its voxels and the mean
// EPG kernel (every voxel contributes the dictionary slice of its own flip angle).  Deterministic two-level
// reduction: a wave per (slice of the voxel list, ROI) accumulates the echoes of its matching voxels in voxel order
// (lane <-> echo) and a flip-angle histogram; one workgroup per ROI then adds the slices in order and forms
// mean kernel = sum_fa count[fa] D[fa] / nv.
// ------------------------------------------------------------------------------------------
struct RoiArgs {
    int nte, nt2, nfa, nroi, nslice;
    int64_t nvox, vs, es;
    const double *data;
    const int32_t *roi;        // [nvox] ROI ordinal 0..nroi-1, negative = none
    const double *fa_index;    // [nvox] or NULL
    double *part_sig;          // [nroi][nslice][64]
    int *part_cnt;             // [nroi][nslice][nfa]
    const double *Dsrc;        // [nfa][nte][nt2]
    double *Ddst;              // [nroi][nte][nt2]
    double *mean_sig;          // [nroi][nte]
    double *count;             // [nroi] voxels per ROI
    int *err;
};

__global__ __launch_bounds__(64) void roi_partial_kernel(RoiArgs A)
{
    extern __shared__ int roi_hist[];
    const int lane = lane_id();
    const int sl = (int)blockIdx.x, r = (int)blockIdx.y;
    for (int f = lane; f < A.nfa; f += 64) roi_hist[f] = 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t per = (A.nvox + A.nslice - 1) / A.nslice;
    const int64_t lo = sl * per, hi = min(A.nvox, lo + per);
    double acc = 0.0;
    for (int64_t base = lo; base < hi; base += 64) {
        const int64_t v = base + lane;
        u64 hit = ballot(v < hi && A.roi[v] == r);
        while (hit) {
            const int bit = first_lane(hit);
            hit &= hit - 1ull;
            const int64_t vv = base + bit;
            if (lane < A.nte) acc += A.data[vv * A.vs + lane * A.es];
            const int fa = A.fa_index ? (int)A.fa_index[vv] : 0;
            if (fa < 0 || fa >= A.nfa) { if (lane == 0) atomicOr(A.err, 1); }
            else if (lane == 0) roi_hist[fa] += 1;
        }
    }
    __builtin_amdgcn_wave_barrier();
    A.part_sig[((size_t)r * A.nslice + sl) * 64 + lane] = acc;
    for (int f = lane; f < A.nfa; f += 64) A.part_cnt[((size_t)r * A.nslice + sl) * A.nfa + f] = roi_hist[f];
}

__global__ __launch_bounds__(256) void roi_finish_kernel(RoiArgs A)
{
    extern __shared__ int roi_tot[];           // [nfa]
    __shared__ double s_nv;
    const int r = (int)blockIdx.x;
    for (int f = threadIdx.x; f < A.nfa; f += blockDim.x) {
        int c = 0;
        for (int sl = 0; sl < A.nslice; ++sl) c += A.part_cnt[((size_t)r * A.nslice + sl) * A.nfa + f];
        roi_tot[f] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long nv = 0;
        for (int f = 0; f < A.nfa; ++f) nv += roi_tot[f];
        s_nv = (double)nv;
        A.count[r] = (double)nv;
    }
    __syncthreads();
    const double nv = s_nv;
    if ((int)threadIdx.x < A.nte) {
        double t = 0.0;
        for (int sl = 0; sl < A.nslice; ++sl) t += A.part_sig[((size_t)r * A.nslice + sl) * 64 + threadIdx.x];
        A.mean_sig[(size_t)r * A.nte + threadIdx.x] = t / nv;
    }
    const int sz = A.nte * A.nt2;
    for (int i = threadIdx.x; i < sz; i += blockDim.x) {
        double t = 0.0;
        for (int f = 0; f < A.nfa; ++f) { const int c = roi_tot[f]; if (c) t = fma((double)c, A.Dsrc[(size_t)f * sz + i], t); }
        A.Ddst[(size_t)r * sz + i] = t / nv;
    }
}

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
struct met2_plan {
    int n_te, n_t2, n_fa;
    met2_options opt;
    int cus;
    double *dD = nullptr;       // [nfa][nte][nt2]
    double *dB = nullptr;       // [nfa][nt2][nt2]
    double *dKband = nullptr;   // [5][64]
    double *dLband = nullptr;   // [5][64]
    double *dDt = nullptr;      // [n_fa][n_t2][n_te] transposed dictionary
    double *dKd = nullptr;      // [n_t2][n_t2] dense L^T L (row loads of the warm-start refactorisation)
    double *dLam = nullptr;     // [nlam]
    double *dT2 = nullptr;      // [nt2]
    int nlam = 0;
    bool have_dict = false, have_pen = false, have_t2 = false;
    std::vector<double> Lhost;  // dense penalty as given
    double log_detL = 0.0;      // log(det(L)) as bayesian_interpolation.py:100,123 uses it (-inf for L2)
    // sort buffers (grown on demand)
    int64_t cap_vox = 0;
    int *dKey = nullptr, *dPerm = nullptr, *dSmall = nullptr, *dOvf = nullptr;
    char *dSeed = nullptr;                                // seed_kernel's output: [4][nfa] SeedRec
    bool seeds_valid = false; double seeds_key[7] = {0};  // (t2sparc_lambda and the six interval ends the seeds and tables were built for)
    bool seeds_ok = false;                                // B + lambda K is positive definite at the seed lambdas (checked on the host for
                                                          // flip angle 0): only then is the seeded start the cold start's solution
    double *dH = nullptr; int64_t cap_h = 0;              // FA walk: h of every flip angle for one pass of voxels (fa_project_kernel), grown on demand
    double *dBtab = nullptr; int btab_stride = 0;         // BayesReg factor tables [nfa][MET2_BAYES_TABLE][btab_stride] (built with the seeds)
    double *dQt = nullptr;                                // [nfa][16][n_te]: the basis vectors themselves (lower bounds of the brute-force FA search)
    double *dAq = nullptr, *dAqRes = nullptr;             // GCV: [nfa][n_t2][16] = (Q^T D)^T in the flip angle's low-rank basis, [nfa] what the basis leaves of D
    double gcv_res = 0.0;                                 // the largest of dAqRes
    bool gcv_lr = false;                                  // every flip angle's dictionary is of numerical rank <= 16: the GCV trace takes the 17 x 17 form
    double *dChol = nullptr; int64_t cap_chol = 0;        // BayesReg at two bins per lane: one packed factor per resident wave (chol_lean), grown on demand
    double *dLcSave = nullptr; int64_t cap_lc = 0;        // L-curve at two bins per lane: the sweep states of queued voxels (fit_kernel.hpp: FitArgs::lc_save) and, behind them, lc_at
    double *dBig = nullptr; int64_t cap_big = 0;          // the waves' spill-over slots: factor columns beyond the LDS capacity (nnls_big.hpp), grown on demand
    int last_spill = 0;                                   // voxels of the last finished fit(s) that used them
    double blam[MET2_BAYES_TABLE > 0 ? MET2_BAYES_TABLE : 1];
    int *hErr = nullptr;                                  // pinned [4]: the FA-range error word of an enqueued fit lands here, and its spill-over queue's tail (= count) and head
    bool err_pending = false;
    hipStream_t err_stream = nullptr;                     // the stream the fits since the last finish were enqueued on (one plan serves one stream at a time)
    int32_t *dStatus = nullptr; int64_t cap_status = 0;   // internal status words when the caller passes none   // dSmall: hist|cursor|bucket_start|chunk_start|queue|err
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    bool timed = false, timed2 = false;
};

// determinant by LU with partial pivoting (scipy.linalg.det at bayesian_interpolation.py:100)
static double det_lu(int n, const double *Ain)
{
    std::vector<double> A(Ain, Ain + (size_t)n * n);
    double det = 1.0;
    for (int c = 0; c < n; ++c) {
        int piv = c; double mx = fabs(A[(size_t)c * n + c]);
        for (int r = c + 1; r < n; ++r) if (fabs(A[(size_t)r * n + c]) > mx) { mx = fabs(A[(size_t)r * n + c]); piv = r; }
        if (mx == 0.0) return 0.0;
        if (piv != c) { for (int j = 0; j < n; ++j) std::swap(A[(size_t)c * n + j], A[(size_t)piv * n + j]); det = -det; }
        double d = A[(size_t)c * n + c];
        det *= d;
        for (int r = c + 1; r < n; ++r) {
            double l = A[(size_t)r * n + c] / d;
            if (l != 0.0) for (int j = c + 1; j < n; ++j) A[(size_t)r * n + j] -= l * A[(size_t)c * n + j];
        }
    }
    return det;
}

static void default_lambda_grid(std::vector<double> &g)
{
    g.assign(50, 0.0);   // motor:248-251
    const double l0 = log10(1e-8), l1 = log10(10.0);
    for (int i = 0; i < 49; ++i) g[i + 1] = pow(10.0, l0 + (l1 - l0) * (double)i / 48.0);
    g[49] = 10.0;
}

static int ensure_sort_bufs(met2_plan *p, int64_t nvox)
{
    if (nvox <= p->cap_vox) return MET2_OK;
    if (p->dKey) HIPCHK(hipFree(p->dKey));
    if (p->dPerm) HIPCHK(hipFree(p->dPerm));
    if (p->dOvf) HIPCHK(hipFree(p->dOvf));
    HIPCHK(hipMalloc(&p->dKey, sizeof(int) * (size_t)nvox));
    HIPCHK(hipMalloc(&p->dPerm, sizeof(int) * (size_t)nvox));
    HIPCHK(hipMalloc(&p->dOvf, sizeof(int) * (size_t)nvox));
    p->cap_vox = nvox;
    return MET2_OK;
}

// met2_fit_host reserves a plan's per-voxel scratch for its largest block before the block loop (met2_host.hip)
namespace met2 { __attribute__((visibility("hidden"))) int plan_reserve(met2_plan *p, int64_t nvox)
{
    if (!p) return fail(MET2_E_INVALID, "NULL plan");
    USE_DEVICE(p->opt.device);
    return ensure_sort_bufs(p, nvox);
} }

static SortBufs sort_bufs(met2_plan *p)
{
    SortBufs sb;
    const int nf = p->n_fa + 1;
    sb.key = p->dKey; sb.perm = p->dPerm; sb.ovf = p->dOvf;
    sb.hist = p->dSmall; sb.cursor = p->dSmall + nf; sb.bucket_start = p->dSmall + 2 * nf; sb.chunk_start = p->dSmall + 3 * nf;
    sb.queue = p->dSmall + 4 * nf; sb.err = p->dSmall + 4 * nf + 1; sb.xq = p->dSmall + 4 * nf + 8;
    return sb;
}


// Development knobs (first-pass capacity, waves per CU, queue granularity) exist only in builds with -DMET2_TUNING; the shipped
// library reads two environment variables: MET2_NO_SEED (A/B test of the plan-level seeds) and MET2_DEBUG (synchronous progress lines).
static int tuning_env(const char *name, int lo, int hi, int dflt)
{
#ifdef MET2_TUNING
    if (const char *e = getenv(name)) { const int v = atoi(e); if (v >= lo && v <= hi) return v; }
#else
    (void)name; (void)lo; (void)hi;
#endif
    return dflt;
}

// kmax_cap > 0: capacity of the passive set for this launch (fast path); 0: full capacity n.
// `method` is the kernel's template method (10 + base for the objective-grid kernels), so that the waves per workgroup come from
// the same number as the kernel's __launch_bounds__.
static int fit_geometry(const met2_plan *p, int method, LaunchGeom &g, int kmax_cap = 0, int wave_cap = 0, int64_t nvox = -1)
{
    const int n = p->n_t2, m = p->n_te;
    const int base = method >= 10 ? method - 10 : method;
    g.nb = n > 64 ? 2 : 1;
    g.kmax = (kmax_cap > 0 && kmax_cap < n) ? kmax_cap : n;
    g.wave_doubles = col_base(g.kmax);                                 // the factor, packed by columns without padding (nnls_wave.hpp)
    if (base == MET2_GCV && gcv_lds_doubles(m, n) > g.wave_doubles) g.wave_doubles = gcv_lds_doubles(m, n);      // M ((m+1)^2) + vectors + support list
    if (base == MET2_GCV_LR && gcv_lds_doubles(MET2_GCV_LR_RANK, n) > g.wave_doubles) g.wave_doubles = gcv_lds_doubles(MET2_GCV_LR_RANK, n);
    if (method == MET2_BAYESREG && g.nb == 2 && chol_panel_doubles(n) > g.wave_doubles) g.wave_doubles = chol_panel_doubles(n);   // chol_lean's 16-row panel
    g.wave_doubles = (g.wave_doubles + 1) & ~1;                        // every wave's region starts 16-byte aligned
    // D, D^T, B and K are read through L1/L2: with warm starts a lambda evaluation reads only ~k rows of B, and the LDS is worth
    // more as room for resident waves (measured on X2/L2 in round 1: staged 8 waves 1.555 M voxels/s, unstaged 11 waves 1.821 M)
    const size_t per_wave = sizeof(double) * (size_t)g.wave_doubles;
    const size_t budget = 160 * 1024 - 64;
    if (per_wave > budget) return fail(MET2_E_UNSUPPORTED, "shape does not fit the LDS budget");
    int w = (int)(budget / per_wave);
    const int wmax = wave_cap > 0 ? wave_cap : method_max_waves(method, g.nb);     // (the FA kernels have their own launch bounds)
    if (w > wmax) w = wmax;
    // small voxel lists (every wave pulls single voxels): no more waves per CU than the list gives every CU, in whole
    // multiples of the four SIMDs -- with 16 waves per CU racing for 1 024 voxels the CUs that win run four voxels per SIMD while
    // others idle (configs[0]: 1.16 ms at 16 waves, 0.85 ms at 4)
    if (nvox >= 0) {
        const int cus = p->cus > 0 ? p->cus : 256;
        int64_t per_cu = (nvox + cus - 1) / cus;
        int ww = (int)(((per_cu + 3) / 4) * 4);
        if (ww < 4) ww = 4;
        if (ww < w) w = ww;
    }
    w = tuning_env("MET2_WAVES", 1, w, w);
    g.waves = w; g.block = 64 * w;
    g.lds = (int)(per_wave * w + 64);
    g.grid = p->cus > 0 ? p->cus : 256;
    return MET2_OK;
}

// Fast-path capacity: the passive set rarely exceeds ~0.75 n (X2/L2 at nT2=60 peaks at 42-45 on the first,
// large-lambda Brent points), and LDS per wave grows with kmax^2.  Methods whose second phase needs the
// full n x n region (BayesReg) or their own matrix (GCV) do not use it.
static int fast_kmax(const met2_plan *p, int method)
{
    if (method >= 10 || (method == MET2_BAYESREG && p->n_t2 <= 64)) return 0;     // (BayesReg at one bin per lane factorises n x n in the wave's region)
    { const int kk = tuning_env("MET2_KMAX", 8, p->n_t2 - 1, -1); if (kk > 0) return kk; }
    // the largest capacity that still lets 16 waves share the LDS, but not below 0.6 n
    // (measured on X2/L2, nT2 = 60: kmax 48 -> 1.95 M voxels/s, 50 -> 2.10 M, 52 (14 waves) -> 2.03 M, 60 (11 waves) -> 1.84 M;
    //  nT2 = 120 with the per-wave queue, where the clean-up pass is cheap: kmax 56 / 64 / 72 / 80 / 96 ->
    //  320 / 464 / 583 / 534 / 391 k voxels/s at 12 / 9 / 7 / 6 / 4 waves per CU)
    int k16 = 8;
    while (16 * sizeof(double) * (size_t)col_base(k16 + 1) <= 160 * 1024 - 64) ++k16;
    int k = (3 * p->n_t2 + 4) / 5;
    if (k16 > k) k = k16;
    if (p->n_t2 > 64) {
        // two bins per lane: those kernels are compiled for 8 waves per CU (method_max_waves), so the capacity that still lets EIGHT
        // regions share the LDS -- 71 at nT2 = 120, where 0.6 n = 72 gave 7 (round 3, X2/L2 at 48 x 120 on 32 768 voxels: first pass
        // 26.1 -> 23.2 ms, clean-up 5.2 -> 6.4 ms; 64 -> 22.4 + 17.7 ms; 80 / 90 / 110 -> 29.0 / 41.4 / 53.3 ms with a clean-up pass
        // that stays at ~5 ms: a few voxels whose set at the first abscissae is nearly all of the grid.  A middle pass at capacity
        // 88 or 100 between the two made the clean-up slower, 6.7 -> 10 ms: those voxels outgrow it too and are solved three times) ...
        int k8 = 8;
        while (8 * sizeof(double) * (size_t)col_base(k8 + 1) <= 160 * 1024 - 64) ++k8;
        k = k8;
        // ... and GCV's region is set by its (m + 1)^2 matrix anyway: the largest factor that fits in it
        if (method == MET2_GCV) while (col_base(k + 1) <= gcv_lds_doubles(p->n_te, p->n_t2)) ++k;
    }
    return k < p->n_t2 ? k : 0;
}

// Middle rung of the capacity ladder, GCV at two bins per lane only (BayesReg's few clean-up voxels measure the same with and without): the full n x n factor of the clean-up pass (58 KB at nT2 = 120)
// leaves a CU two waves; the largest capacity that gives it three is 116, and on the [1e-8, 10] interval of algorithms.py:279 the
// sets that outgrow the first pass stay under it (131 072 voxels of configs[4] with a FIRST pass at capacity 116: 0.07 ms of clean-up
// left).  X2's interval visits 6.18 for every voxel, where those sets are the whole grid (clean-up 4.96 ms after a first pass at
// 118): for it a middle pass means solving them three times (measured: 6.7 -> 10 ms), so it gets none.  0: no middle pass.
static int mid_kmax(const met2_plan *p, int method, int kfast)
{
    if (!kfast || p->n_t2 <= 64 || (method != MET2_GCV && method != MET2_GCV_LR)) return 0;
    { const int kk = tuning_env("MET2_KMID", 0, p->n_t2 - 1, -1); if (kk >= 0) return kk > kfast ? kk : 0; }
    int k3 = kfast;
    while (k3 + 1 < p->n_t2 && 3 * sizeof(double) * (size_t)col_base(k3 + 1) <= 160 * 1024 - 64) ++k3;
    return (k3 >= kfast + 8 && k3 < p->n_t2) ? k3 : 0;
}

template <int METHOD>
static int launch_fit(const FitArgs &A, const LaunchGeom &g, hipStream_t s, bool second = false)
{
    if (second) {      // the spill-over kernel (fit_kernel.hpp): every method but the objective grids
        if (METHOD >= 10) return fail(MET2_E_INVALID, "no spill-over kernel for this method");
        constexpr int M2 = METHOD >= 10 ? MET2_NNLS : METHOD;
        return g.nb == 2 ? launch_fit_nb<M2, 2, true>(A, g, s) : launch_fit_nb<M2, 1, true>(A, g, s);
    }
    return g.nb == 2 ? launch_fit_nb<METHOD, 2, false>(A, g, s) : launch_fit_nb<METHOD, 1, false>(A, g, s);
}

// -DMET2_ONLY=<method number> builds the fit kernel of that method only (plus plain NNLS): a development switch that
// cuts the compile time of an experiment from ~80 s to ~15 s; the shipped library is built without it
#ifdef MET2_ONLY
#define MET2_HAS(m) ((m) == MET2_ONLY || (m) == 0)
#else
#define MET2_HAS(m) 1
#endif

static int launch_method(int method, const FitArgs &A, const LaunchGeom &g, hipStream_t s, bool second = false)
{
    switch (method) {
#if MET2_HAS(2)
    case 10 + MET2_X2: return launch_fit<10 + MET2_X2>(A, g, s);
    case MET2_X2: return launch_fit<MET2_X2>(A, g, s, second);
#endif
#if MET2_HAS(4)
    case 10 + MET2_GCV: return launch_fit<10 + MET2_GCV>(A, g, s);
    case MET2_GCV: return launch_fit<MET2_GCV>(A, g, s, second);
    case 10 + MET2_GCV_LR: return launch_fit<10 + MET2_GCV_LR>(A, g, s);
    case MET2_GCV_LR: return launch_fit<MET2_GCV_LR>(A, g, s, second);
#endif
#if MET2_HAS(5)
    case 10 + MET2_BAYESREG: return launch_fit<10 + MET2_BAYESREG>(A, g, s);
    case MET2_BAYESREG: return launch_fit<MET2_BAYESREG>(A, g, s, second);
#endif
    case MET2_NNLS: return launch_fit<MET2_NNLS>(A, g, s, second);
#if MET2_HAS(1)
    case MET2_T2SPARC: return launch_fit<MET2_T2SPARC>(A, g, s, second);
#endif
#if MET2_HAS(3)
    case MET2_LCURVE: return launch_fit<MET2_LCURVE>(A, g, s, second);
#endif
    default: return fail(MET2_E_UNSUPPORTED, "method not built");
    }
}


// slope weights of the not-a-knot cubic spline: knot slopes = W y (the system depends on the knots only)
static void spline_weights_host(int n, const double *x, std::vector<double> &W)
{
    std::vector<double> A((size_t)n * n, 0.0), Rm((size_t)n * n, 0.0);   // A s = Rm y
    auto h = [&](int i) { return x[i + 1] - x[i]; };
    {   // not-a-knot at x[1]
        const double h0 = h(0), h1 = h(1);
        A[0] = h1; A[1] = h0 + h1;
        // rhs = ((3h0+2h1) h1 d0 + h0^2 d1)/(h0+h1), d0 = (y1-y0)/h0, d1 = (y2-y1)/h1
        const double c0 = (3.0 * h0 + 2.0 * h1) * h1 / (h0 + h1) / h0, c1 = h0 * h0 / (h0 + h1) / h1;
        Rm[0] += -c0; Rm[1] += c0 - c1; Rm[2] += c1;
    }
    for (int i = 1; i < n - 1; ++i) {
        const double hm = h(i - 1), hp = h(i);
        A[(size_t)i * n + i - 1] = hp; A[(size_t)i * n + i] = 2.0 * (hm + hp); A[(size_t)i * n + i + 1] = hm;
        const double cm = 3.0 * hp / hm, cp = 3.0 * hm / hp;               // rhs = 3 (hp dm + hm dp)
        Rm[(size_t)i * n + i - 1] += -cm; Rm[(size_t)i * n + i] += cm - cp; Rm[(size_t)i * n + i + 1] += cp;
    }
    {   // not-a-knot at x[n-2]
        const double ha = h(n - 3), hb = h(n - 2);
        A[(size_t)(n - 1) * n + n - 2] = ha + hb; A[(size_t)(n - 1) * n + n - 1] = ha;
        const double ca = hb * hb / (ha + hb) / ha, cb = (2.0 * ha + 3.0 * hb) * ha / (ha + hb) / hb;
        Rm[(size_t)(n - 1) * n + n - 3] += -ca; Rm[(size_t)(n - 1) * n + n - 2] += ca - cb; Rm[(size_t)(n - 1) * n + n - 1] += cb;
    }
    // W = A^-1 Rm by Gaussian elimination with partial pivoting on [A | Rm]
    for (int c = 0; c < n; ++c) {
        int p = c; double mx = fabs(A[(size_t)c * n + c]);
        for (int r = c + 1; r < n; ++r) if (fabs(A[(size_t)r * n + c]) > mx) { mx = fabs(A[(size_t)r * n + c]); p = r; }
        if (p != c) for (int j = 0; j < n; ++j) { std::swap(A[(size_t)c * n + j], A[(size_t)p * n + j]); std::swap(Rm[(size_t)c * n + j], Rm[(size_t)p * n + j]); }
        for (int r = c + 1; r < n; ++r) {
            const double l = A[(size_t)r * n + c] / A[(size_t)c * n + c];
            if (l == 0.0) continue;
            for (int j = c; j < n; ++j) A[(size_t)r * n + j] -= l * A[(size_t)c * n + j];
            for (int j = 0; j < n; ++j) Rm[(size_t)r * n + j] -= l * Rm[(size_t)c * n + j];
        }
    }
    W.assign((size_t)n * n, 0.0);
    for (int c = n - 1; c >= 0; --c)
        for (int j = 0; j < n; ++j) {
            double t = Rm[(size_t)c * n + j];
            for (int q = c + 1; q < n; ++q) t -= A[(size_t)c * n + q] * W[(size_t)q * n + j];
            W[(size_t)c * n + j] = t / A[(size_t)c * n + c];
        }
}

extern "C" int met2_smooth_separable(int32_t device, int32_t nx, int32_t ny, int32_t nz, int32_t nt, int32_t radius, const double *weights,
                                     const double *data, double *out, double *work, void *stream)
{
    if (nx < 0 || ny < 0 || nz < 0 || nt < 1) return fail(MET2_E_INVALID, "bad shape");
    if (radius < 0 || radius > MET2_SMOOTH_MAX_RADIUS) return fail(MET2_E_UNSUPPORTED, "kernel radius must be 0..32");
    const int64_t total = (int64_t)nx * ny * nz * nt;
    if (total == 0) return MET2_OK;
    if (!weights || !data || !out) return fail(MET2_E_INVALID, "NULL argument");
    if (data == out || work == out || work == data) return fail(MET2_E_INVALID, "data, out and work must be distinct");
    USE_DEVICE(device);
    hipStream_t s = (hipStream_t)stream;
    double *tmp = work;
    if (!tmp) HIPCHK(hipMalloc(&tmp, sizeof(double) * (size_t)total));
    SmoothArgs A;
    A.radius = radius; A.total = total;
    for (int i = 0; i < 2 * radius + 1; ++i) A.w[i] = weights[i];
    const int dims[3] = {nx, ny, nz};
    const int64_t strides[3] = {(int64_t)ny * nz * nt, (int64_t)nz * nt, (int64_t)nt};
    const double *src = data;
    double *dst = out;                                              // x: data -> out, y: out -> work, z: work -> out
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 64);
    for (int ax = 0; ax < 3; ++ax) {
        A.n = dims[ax]; A.stride = strides[ax]; A.src = src; A.dst = dst;
        hipLaunchKernelGGL(smooth_axis_kernel, dim3(grid), dim3(256), 0, s, A);
        src = dst;
        dst = (dst == out) ? tmp : out;
    }
    HIPCHK(hipGetLastError());
    if (!work) { HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipFree(tmp)); }
    return MET2_OK;
}

extern "C" int met2_nesma(int32_t device, int32_t nx, int32_t ny, int32_t nz, int32_t nt, const double *data,
                          const uint8_t *mask, double *out, void *stream)
{
    if (nx < 0 || ny < 0 || nz < 0 || nt < 1) return fail(MET2_E_INVALID, "bad shape");
    if (nt > MET2_NESMA_MAX_NT) return fail(MET2_E_UNSUPPORTED, "NESMA supports at most 128 echoes");
    const int64_t nvox = (int64_t)nx * ny * nz;
    if (nvox == 0) return MET2_OK;
    if (!data || !out) return fail(MET2_E_INVALID, "NULL argument");
    if (data == out) return fail(MET2_E_INVALID, "NESMA cannot run in place");
    if ((nvox + 3) / 4 > 0x7fffffffLL) return fail(MET2_E_UNSUPPORTED, "volume too large for one launch");
    USE_DEVICE(device);
    hipStream_t s = (hipStream_t)stream;
    NesmaArgs A;
    A.nx = nx; A.ny = ny; A.nz = nz; A.nt = nt; A.data = data; A.mask = mask; A.out = out; A.nvox = nvox;
    const dim3 grid((unsigned)((((nvox + 3) / 4 + 7) / 8) * 8)), block(256);      // a multiple of 8 for the XCD remap
    A.srow = ((nt + 23) / 32) * 32 + 8;
    const size_t lds = sizeof(double) * 4 * 2 * MET2_NESMA_HW * (size_t)A.srow;
    if (nt == 32)      hipLaunchKernelGGL((nesma_kernel<1, 2, 32>), grid, block, lds, s, A);
    else if (nt < 32)  hipLaunchKernelGGL((nesma_kernel<1, 2, 0>), grid, block, lds, s, A);
    else if (nt == 48) hipLaunchKernelGGL((nesma_kernel<1, 1, 48>), grid, block, lds, s, A);
    else if (nt <= 64) hipLaunchKernelGGL((nesma_kernel<1, 1, 0>), grid, block, lds, s, A);
    else               hipLaunchKernelGGL((nesma_kernel<2, 1, 0>), grid, block, lds, s, A);
    HIPCHK(hipGetLastError());
    return MET2_OK;
}

struct SplineTables { int device = -1; double *d = nullptr; size_t cap = 0; std::vector<double> last; };
static thread_local SplineTables g_spline_tables;
// for host threads that end (met2_fit_host's per-plan threads): frees the calling thread's tables
namespace met2 { __attribute__((visibility("hidden"))) void spline_tables_release()
{
    SplineTables &T = g_spline_tables;
    if (T.d) { DevGuard g(T.device); (void)hipDeviceSynchronize(); (void)hipFree(T.d); }
    T.d = nullptr; T.cap = 0; T.device = -1; T.last.clear();
} }

extern "C" int met2_fa_spline_select_strided(int32_t device, int64_t nvox, int32_t n_lr, const double *alpha_lr, const double *resid,
                                             int32_t n_hr, const double *alpha_hr, int32_t n_te, const double *data, int64_t voxel_stride,
                                             int64_t echo_stride, const uint8_t *mask, double *fa_index, double *xmin, void *stream)
{
    if (nvox == 0) return MET2_OK;
    if (!alpha_lr || !resid || !alpha_hr || !data || !fa_index) return fail(MET2_E_INVALID, "NULL argument");
    if (n_lr < 4 || n_lr > MET2_MAX_LR) return fail(MET2_E_UNSUPPORTED, "coarse FA grid must have 4..32 points");
    if (n_hr < 1 || n_te < 1) return fail(MET2_E_INVALID, "bad shape");
    for (int i = 1; i < n_lr; ++i) if (!(alpha_lr[i] > alpha_lr[i - 1])) return fail(MET2_E_INVALID, "coarse FA grid must increase");
    if (nvox <= 0) return MET2_OK;
    USE_DEVICE(device);
    hipStream_t s = (hipStream_t)stream;
    std::vector<double> W;
    spline_weights_host(n_lr, alpha_lr, W);
    const size_t nd = (size_t)n_lr + (size_t)n_lr * n_lr + (size_t)n_hr;
    std::vector<double> hb(nd);
    memcpy(hb.data(), alpha_lr, sizeof(double) * n_lr);
    memcpy(hb.data() + n_lr, W.data(), sizeof(double) * n_lr * n_lr);
    memcpy(hb.data() + n_lr + (size_t)n_lr * n_lr, alpha_hr, sizeof(double) * n_hr);
    // The tables (two grids and the spline's weight matrix, a few KB) live in a per-thread device buffer that is written only when they
    // CHANGE: a driver calls this once per chunk with the same grids, and an allocation + blocking wait + hipFree per call stood in the
    // way of its pipeline (hipFree waits for the whole device).  Changing them waits for the device first (a kernel of an earlier call may
    // still read the old ones).
    SplineTables &T = g_spline_tables;
    if (T.device != device || T.cap < nd) {
        if (T.d) { DevGuard old(T.device); HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(T.d)); T.d = nullptr; T.cap = 0; T.last.clear(); }
        HIPCHK(hipMalloc((void **)&T.d, sizeof(double) * std::max<size_t>(nd, 2048)));
        T.cap = std::max<size_t>(nd, 2048); T.device = device;
    }
    if (T.last != hb) {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(T.d, hb.data(), sizeof(double) * nd, hipMemcpyHostToDevice));
        T.last = hb;
    }
    double *dbuf = T.d;
    SplineArgs A;
    A.nlr = n_lr; A.nhr = n_hr; A.nte = n_te;
    A.alpha_lr = dbuf; A.W = dbuf + n_lr; A.alpha_hr = dbuf + n_lr + (size_t)n_lr * n_lr;
    A.resid = resid; A.data = data; A.vs = voxel_stride; A.es = echo_stride; A.mask = mask; A.fa_index = fa_index; A.xmin = xmin; A.nvox = nvox;
    hipLaunchKernelGGL(fa_spline_kernel, dim3((unsigned)((nvox + 127) / 128)), dim3(128), 0, s, A);
    HIPCHK(hipGetLastError());
    return MET2_OK;
}

extern "C" int met2_fa_spline_select(int32_t device, int64_t nvox, int32_t n_lr, const double *alpha_lr, const double *resid,
                                     int32_t n_hr, const double *alpha_hr, int32_t n_te, const double *data, const uint8_t *mask,
                                     double *fa_index, double *xmin, void *stream)
{
    return met2_fa_spline_select_strided(device, nvox, n_lr, alpha_lr, resid, n_hr, alpha_hr, n_te, data, n_te, 1, mask, fa_index, xmin, stream);
}


// Plan-level seeds (seed_kernel) are built where the dictionary, the penalty or T2SPARC's lambda are SET -- entries that
// synchronise -- never inside a fit: a fit on another stream or thread only ever reads finished records.
// Seeded starts equal the cold-start solution only when B + lambda K is positive definite (one minimiser).  That holds for the
// reference's I / L1 / L2 / InvT2; met2_plan_set_penalty_dense takes any banded L, e.g. a pure second difference whose null
// space meets that of D.  Checked here on the host by a Cholesky factorisation of B_0 + lambda K (flip angle 0, the three seed
// lambdas) with a relative pivot threshold; when it fails the plan simply runs unseeded.
static bool host_posdef(int n, std::vector<double> G)
{
    double dmax = 0.0;
    for (int i = 0; i < n; ++i) dmax = std::max(dmax, fabs(G[(size_t)i * n + i]));
    for (int j = 0; j < n; ++j) {
        double d = G[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= G[(size_t)j * n + k] * G[(size_t)j * n + k];
        if (!(d > 1e-13 * dmax)) return false;
        d = sqrt(d);
        G[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double t = G[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) t -= G[(size_t)i * n + k] * G[(size_t)j * n + k];
            G[(size_t)i * n + j] = t / d;
        }
    }
    return true;
}

static int ensure_seeds(met2_plan *p, hipStream_t s)
{
    if (!p->have_dict || !p->have_pen) return MET2_OK;
    const double key[7] = {p->opt.t2sparc_lambda, p->opt.x2_lo, p->opt.x2_hi, p->opt.gcv_lo, p->opt.gcv_hi, p->opt.bayes_lo, p->opt.bayes_hi};
    if (p->seeds_valid && memcmp(key, p->seeds_key, sizeof(key)) == 0) return MET2_OK;
    const int n = p->n_t2;
    const double gm = 0.5 * (3.0 - sqrt(5.0));
    SeedArgs SA;
    SA.n = n; SA.m = p->n_te; SA.nfa = p->n_fa;
    SA.Dfa = p->dD; SA.Bfa = p->dB; SA.Dtfa = p->dDt; SA.kband = p->dKband; SA.lband = p->dLband; SA.Kd = p->dKd;
    // the first abscissa of scipy's bounded Brent on each method's interval, a + g (b - a); T2SPARC's fixed lambda
    SA.lam[0] = p->opt.x2_lo + gm * (p->opt.x2_hi - p->opt.x2_lo); SA.lam[1] = p->opt.bayes_lo + gm * (p->opt.bayes_hi - p->opt.bayes_lo);
    SA.lam[2] = p->opt.t2sparc_lambda; SA.lam[3] = p->opt.gcv_lo + gm * (p->opt.gcv_hi - p->opt.gcv_lo);
    SA.out = p->dSeed;
    {
        std::vector<double> B0((size_t)n * n), K((size_t)n * n);
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMemcpy(B0.data(), p->dB, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(K.data(), p->dKd, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost));
        bool ok = true;
        for (int q = 0; q < 4 && ok; ++q) {
            std::vector<double> G((size_t)n * n);
            for (size_t i = 0; i < G.size(); ++i) G[i] = B0[i] + SA.lam[q] * K[i];
            ok = SA.lam[q] > 0.0 && host_posdef(n, G);
        }
        p->seeds_ok = ok;
    }
    const int lds = (int)sizeof(double) * col_base(n) + 64;
    if (n <= 64) {
        HIPCHK(hipFuncSetAttribute((const void *)seed_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(seed_kernel<1>, dim3(p->n_fa, 4), dim3(64), lds, s, SA);
    } else {
        HIPCHK(hipFuncSetAttribute((const void *)seed_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(seed_kernel<2>, dim3(p->n_fa, 4), dim3(64), lds, s, SA);
    }
    HIPCHK(hipGetLastError());
    if (MET2_BAYES_TABLE > 0) {
        // BayesReg's shared abscissae on its interval ([1e-8, 2] in the reference; fminbound_dev with an objective that decreases towards
        // the lower bound): a + g (b - a), one golden step up, then golden steps down
        const double a = p->opt.bayes_lo, b = p->opt.bayes_hi;
        BayesTabArgs TA;
        TA.n = n; TA.m = p->n_te; TA.nfa = p->n_fa; TA.nj = MET2_BAYES_TABLE;
        TA.Bfa = p->dB; TA.Kd = p->dKd; TA.kband = p->dKband; TA.lband = p->dLband;
        double t0 = a + gm * (b - a);
        for (int j = 0; j < MET2_BAYES_TABLE; ++j) {
            double t;
            if (j == 0) t = t0;
            else if (j == 1) t = t0 + gm * (b - t0);
            else { const double prev = (j == 2) ? t0 : p->blam[j - 1]; t = prev + gm * (a - prev); }
            p->blam[j] = t; TA.lam[j] = t;
        }
        p->btab_stride = col_base(n) + 1;
        if (!p->dBtab) HIPCHK(hipMalloc(&p->dBtab, sizeof(double) * (size_t)p->n_fa * MET2_BAYES_TABLE * p->btab_stride));
        TA.stride = p->btab_stride; TA.out = p->dBtab;
        if (n <= 64) {
            HIPCHK(hipFuncSetAttribute((const void *)bayes_table_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(bayes_table_kernel<1>, dim3(p->n_fa, MET2_BAYES_TABLE), dim3(64), lds, s, TA);
        } else {
            HIPCHK(hipFuncSetAttribute((const void *)bayes_table_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(bayes_table_kernel<2>, dim3(p->n_fa, MET2_BAYES_TABLE), dim3(64), lds, s, TA);
        }
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s));
    p->seeds_valid = true; memcpy(p->seeds_key, key, sizeof(key));
    return MET2_OK;
}

extern "C" {

void met2_default_options(met2_options *o)
{
    memset(o, 0, sizeof(*o));
    o->struct_size = (int32_t)sizeof(met2_options);
    o->device = 0;
    o->x2_factor = 1.02;
    o->t2sparc_lambda = 1.8;
    o->brent_xtol = 1e-5;
    o->brent_maxfun = 0;
    o->t2_myelin_cut = 40.0;
    o->t2_ie_cut = 200.0;
    o->x2_lo = 0.0; o->x2_hi = 10.0;             // algorithms.py:219
    o->gcv_lo = 1e-8; o->gcv_hi = 10.0;          // algorithms.py:280
    o->bayes_lo = 1e-8; o->bayes_hi = 2.0;       // bayesian_interpolation.py:101
}

// a caller's options struct may be shorter than ours (an earlier ABI): what it does not carry keeps the value already in `dst`
static int take_options(met2_options *dst, const met2_options *src)
{
    if (src->struct_size < (int32_t)(2 * sizeof(int32_t))) return fail(MET2_E_INVALID, "met2_options.struct_size not set");
    memcpy(dst, src, sizeof(met2_options) < (size_t)src->struct_size ? sizeof(met2_options) : (size_t)src->struct_size);
    dst->struct_size = (int32_t)sizeof(met2_options);
    const double iv[3][2] = {{dst->x2_lo, dst->x2_hi}, {dst->gcv_lo, dst->gcv_hi}, {dst->bayes_lo, dst->bayes_hi}};
    for (int q = 0; q < 3; ++q)
        if (!(iv[q][0] >= 0.0) || !(iv[q][1] > iv[q][0]) || !std::isfinite(iv[q][1])) return fail(MET2_E_INVALID, "lambda-search interval: 0 <= lo < hi (finite) required");
    return MET2_OK;
}

int met2_abi_version(void) { return MET2_ABI_VERSION; }

int met2_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *met2_last_error(void) { return g_err.c_str(); }

int met2_plan_create(met2_plan **out, int32_t n_te, int32_t n_t2, int32_t n_fa, const met2_options *opt)
{
    if (!out) return fail(MET2_E_INVALID, "out is NULL");
    if (n_te < 2 || n_t2 < 2 || n_fa < 1) return fail(MET2_E_INVALID, "bad shape");
    if (n_t2 > 128) return fail(MET2_E_UNSUPPORTED, "n_t2 > 128 unsupported (two T2 bins per lane)");
    if (n_te > 63) return fail(MET2_E_UNSUPPORTED, "n_te > 63 unsupported (EPG orders live on the 64 lanes)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MET2_E_NODEVICE, "no HIP device visible");
    met2_plan *p = new met2_plan();
    p->n_te = n_te; p->n_t2 = n_t2; p->n_fa = n_fa;
    met2_default_options(&p->opt);
    if (opt) { const int rco = take_options(&p->opt, opt); if (rco) { delete p; return rco; } }
    if (p->opt.device < 0 || p->opt.device >= ndev) { delete p; return fail(MET2_E_INVALID, "device ordinal out of range"); }
    DevGuard dev_guard_(p->opt.device);
    if (dev_guard_.err != hipSuccess) { delete p; return fail(MET2_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(dev_guard_.err)); }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, p->opt.device));
    p->cus = prop.multiProcessorCount;
    HIPCHK(hipMalloc(&p->dD, sizeof(double) * (size_t)n_fa * n_te * n_t2));
    HIPCHK(hipMalloc(&p->dB, sizeof(double) * (size_t)n_fa * n_t2 * n_t2));
    HIPCHK(hipMalloc(&p->dDt, sizeof(double) * (size_t)n_fa * n_te * n_t2));
    HIPCHK(hipMalloc(&p->dAq, sizeof(double) * (size_t)n_fa * n_t2 * MET2_GCV_LR_RANK));
    HIPCHK(hipMalloc(&p->dAqRes, sizeof(double) * (size_t)n_fa));
    HIPCHK(hipMalloc(&p->dQt, sizeof(double) * (size_t)n_fa * MET2_GCV_LR_RANK * n_te));
    HIPCHK(hipMalloc(&p->dKband, sizeof(double) * 5 * 128));
    HIPCHK(hipMalloc(&p->dLband, sizeof(double) * 5 * 128));
    HIPCHK(hipMalloc(&p->dKd, sizeof(double) * (size_t)n_t2 * n_t2));
    HIPCHK(hipMemset(p->dKd, 0, sizeof(double) * (size_t)n_t2 * n_t2));
    HIPCHK(hipMalloc(&p->dT2, sizeof(double) * 128));
    HIPCHK(hipMalloc(&p->dSmall, sizeof(int) * (4 * (size_t)(n_fa + 1) + 16)));
    HIPCHK(hipMalloc(&p->dSeed, sizeof(SeedRec) * 4 * (size_t)n_fa));
    HIPCHK(hipHostMalloc((void **)&p->hErr, 4 * sizeof(int), hipHostMallocDefault));
    p->hErr[0] = p->hErr[1] = p->hErr[2] = p->hErr[3] = 0;
    HIPCHK(hipEventCreate(&p->ev0));
    HIPCHK(hipEventCreate(&p->ev1));
    HIPCHK(hipEventCreate(&p->ev2));
    std::vector<double> g;
    default_lambda_grid(g);
    *out = p;
    return met2_plan_set_lambda_grid(p, g.data(), (int)g.size());
}

int met2_plan_set_options(met2_plan *p, const met2_options *opt)
{
    if (!p || !opt) return fail(MET2_E_INVALID, "NULL argument");
    if (opt->device != p->opt.device) return fail(MET2_E_INVALID, "a plan cannot change device");
    {
        met2_options o = p->opt;
        const int rco = take_options(&o, opt);
        if (rco) return rco;
        if (o.device != p->opt.device) return fail(MET2_E_INVALID, "a plan stays on the device it was created on");
        p->opt = o;
    }
    USE_DEVICE(p->opt.device);
    return ensure_seeds(p, 0);       // T2SPARC's seed belongs to its lambda
}

int met2_plan_get_options(met2_plan *p, met2_options *opt)
{
    if (!p || !opt) return fail(MET2_E_INVALID, "NULL argument");
    *opt = p->opt;
    return MET2_OK;
}

int met2_plan_get_shape(met2_plan *p, int32_t *n_te, int32_t *n_t2, int32_t *n_fa)
{
    if (!p) return fail(MET2_E_INVALID, "NULL plan");
    if (n_te) *n_te = p->n_te;
    if (n_t2) *n_t2 = p->n_t2;
    if (n_fa) *n_fa = p->n_fa;
    return MET2_OK;
}

int met2_plan_destroy(met2_plan *p)
{
    if (!p) return MET2_OK;
    met2::host_release(p);           // what met2_fit_host keeps with the plan (met2_host.hip)
    DevGuard dev_guard_(p->opt.device);
    void *bufs[] = {p->dQt, p->dAq, p->dAqRes, p->dD, p->dB, p->dDt, p->dKband, p->dLband, p->dKd, p->dLam, p->dT2, p->dKey, p->dPerm, p->dOvf, p->dSmall, p->dStatus, p->dSeed, p->dBtab, p->dH, p->dChol, p->dBig, p->dLcSave};
    for (void *b : bufs) (void)hipFree(b);
    if (p->hErr) (void)hipHostFree(p->hErr);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->ev2) (void)hipEventDestroy(p->ev2);
    delete p;
    return MET2_OK;
}

// Everything that is derived from the dictionary alone: Gram matrices, the transposed copy, GCV's low-rank basis.  Blocks (its callers do).
static int build_gram(met2_plan *p, hipStream_t s)
{
    hipLaunchKernelGGL(gram_kernel, dim3(p->n_fa), dim3(256), 0, s, p->n_te, p->n_t2, p->dD, p->dB, p->dDt);
    HIPCHK(hipGetLastError());
    const int lds = (int)sizeof(double) * (p->n_te * p->n_t2 + MET2_GCV_LR_RANK * 64);
    HIPCHK(hipFuncSetAttribute((const void *)gcv_basis_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(gcv_basis_kernel, dim3(p->n_fa), dim3(64), lds, s, p->n_te, p->n_t2, p->dD, p->dAq, p->dQt, p->dAqRes);
    HIPCHK(hipGetLastError());
    std::vector<double> res(p->n_fa);
    HIPCHK(hipMemcpyAsync(res.data(), p->dAqRes, sizeof(double) * p->n_fa, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    bool ok = true;
    p->gcv_res = 0.0;
    for (double r : res) { ok = ok && (r <= MET2_GCV_LR_TOL); p->gcv_res = (r > p->gcv_res || r != r) ? r : p->gcv_res; }      // (nan: not low rank)
    p->gcv_lr = ok;
    p->have_dict = true; p->seeds_valid = false;
    return MET2_OK;
}

int met2_plan_set_t2_grid(met2_plan *p, const double *T2s)
{
    if (!p || !T2s) return fail(MET2_E_INVALID, "NULL argument");
    USE_DEVICE(p->opt.device);
    HIPCHK(hipMemcpy(p->dT2, T2s, sizeof(double) * p->n_t2, hipMemcpyHostToDevice));
    p->have_t2 = true;
    return MET2_OK;
}

int met2_plan_build_dictionary_epg(met2_plan *p, const double *T2s, const double *T1s, double tau, const double *alpha_deg,
                                   double TR, void *stream)
{
    if (!p || !T2s || !T1s || !alpha_deg) return fail(MET2_E_INVALID, "NULL argument");
    USE_DEVICE(p->opt.device);
    hipStream_t s = (hipStream_t)stream;
    double *tmp = nullptr;
    size_t nb = sizeof(double) * (size_t)(2 * p->n_t2 + p->n_fa);
    HIPCHK(hipMalloc(&tmp, nb));
    std::vector<double> h(2 * p->n_t2 + p->n_fa);
    memcpy(h.data(), T2s, sizeof(double) * p->n_t2);
    memcpy(h.data() + p->n_t2, T1s, sizeof(double) * p->n_t2);
    memcpy(h.data() + 2 * p->n_t2, alpha_deg, sizeof(double) * p->n_fa);
    HIPCHK(hipMemcpy(tmp, h.data(), nb, hipMemcpyHostToDevice));
    int waves = p->n_fa * p->n_t2;
    int blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(epg_dictionary_kernel, dim3(blocks), dim3(256), 0, s, p->n_te, p->n_t2, p->n_fa, tmp, tmp + p->n_t2, tau,
                       tmp + 2 * p->n_t2, TR, p->dD);
    HIPCHK(hipGetLastError());
    int rc = build_gram(p, s);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(tmp));
    if (rc) return rc;
    rc = ensure_seeds(p, s);
    if (rc) return rc;
    return met2_plan_set_t2_grid(p, T2s);
}

int met2_plan_set_dictionary(met2_plan *p, const double *dic)
{
    if (!p || !dic) return fail(MET2_E_INVALID, "NULL argument");
    USE_DEVICE(p->opt.device);
    size_t nb = sizeof(double) * (size_t)p->n_fa * p->n_te * p->n_t2;
    double *tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, nb));
    HIPCHK(hipMemcpy(tmp, dic, nb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(relayout_kernel, dim3(256), dim3(256), 0, 0, p->n_te, p->n_t2, p->n_fa, tmp, p->dD, 1);
    HIPCHK(hipGetLastError());
    int rc = build_gram(p, 0);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipFree(tmp));
    if (rc) return rc;
    return ensure_seeds(p, 0);
}

int met2_plan_get_dictionary(met2_plan *p, double *dic)
{
    if (!p || !dic) return fail(MET2_E_INVALID, "NULL argument");
    if (!p->have_dict) return fail(MET2_E_STATE, "no dictionary in the plan");
    USE_DEVICE(p->opt.device);
    size_t nb = sizeof(double) * (size_t)p->n_fa * p->n_te * p->n_t2;
    double *tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, nb));
    hipLaunchKernelGGL(relayout_kernel, dim3(256), dim3(256), 0, 0, p->n_te, p->n_t2, p->n_fa, p->dD, tmp, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(dic, tmp, nb, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(tmp));
    return MET2_OK;
}

int met2_plan_set_penalty_dense(met2_plan *p, const double *L)
{
    if (!p || !L) return fail(MET2_E_INVALID, "NULL argument");
    const int n = p->n_t2;
    for (int i = 0; i < n * n; ++i) if (!std::isfinite(L[i])) return fail(MET2_E_INVALID, "non-finite penalty matrix");
    std::vector<double> K((size_t)n * n, 0.0);
    for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += L[(size_t)i * n + a] * L[(size_t)i * n + b];
        K[(size_t)a * n + b] = t;
    }
    for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) if (abs(a - b) > 2) {
        if (K[(size_t)a * n + b] != 0.0) return fail(MET2_E_UNSUPPORTED, "L^T L has bandwidth > 2");
        if (L[(size_t)a * n + b] != 0.0) return fail(MET2_E_UNSUPPORTED, "penalty matrix has bandwidth > 2");
    }
    std::vector<double> kb(5 * 128, 0.0), lb(5 * 128, 0.0);
    for (int j = 0; j < n; ++j) for (int d = 0; d < 5; ++d) {
        int c = j + d - 2;
        if (c < 0 || c >= n) continue;
        kb[d * 128 + j] = K[(size_t)j * n + c];
        lb[d * 128 + j] = L[(size_t)j * n + c];
    }
    USE_DEVICE(p->opt.device);
    HIPCHK(hipMemcpy(p->dKband, kb.data(), sizeof(double) * 5 * 128, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->dLband, lb.data(), sizeof(double) * 5 * 128, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->dKd, K.data(), sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice));
    p->Lhost.assign(L, L + (size_t)n * n);
    p->log_detL = log(det_lu(n, L));
    p->have_pen = true; p->seeds_valid = false;
    return ensure_seeds(p, 0);
}

int met2_plan_set_penalty(met2_plan *p, int32_t which, const double *T2s)
{
    if (!p) return fail(MET2_E_INVALID, "NULL plan");
    const int n = p->n_t2;
    std::vector<double> L((size_t)n * n, 0.0);
    switch (which) {   // motor:86-111, 263-269
    case MET2_PEN_I: for (int i = 0; i < n; ++i) L[(size_t)i * n + i] = 1.0; break;
    case MET2_PEN_L1: for (int i = 0; i < n; ++i) { L[(size_t)i * n + i] = 1.0; if (i > 0) L[(size_t)i * n + i - 1] = -1.0; } break;
    case MET2_PEN_L2:
        for (int i = 0; i < n; ++i) {
            L[(size_t)i * n + i] = 2.0;
            if (i > 0) L[(size_t)i * n + i - 1] = -1.0;
            if (i < n - 1) L[(size_t)i * n + i + 1] = -1.0;
        }
        L[0] = 1.0; L[(size_t)(n - 1) * n + n - 1] = 1.0;
        break;
    case MET2_PEN_INVT2:
        if (!T2s) return fail(MET2_E_INVALID, "InvT2 needs the T2 grid");
        for (int i = 0; i < n; ++i) {
            double prev = (i == 0) ? T2s[0] - 1.0 : T2s[i - 1];
            double d = T2s[i] - prev;
            if (i == 0) d = T2s[1] - T2s[0];
            L[(size_t)i * n + i] = 1.0 / d;
        }
        break;
    default: return fail(MET2_E_INVALID, "unknown penalty");
    }
    return met2_plan_set_penalty_dense(p, L.data());
}

int met2_plan_get_penalty(met2_plan *p, double *L)
{
    if (!p || !L) return fail(MET2_E_INVALID, "NULL argument");
    if (!p->have_pen) return fail(MET2_E_STATE, "no penalty set");
    memcpy(L, p->Lhost.data(), sizeof(double) * p->Lhost.size());
    return MET2_OK;
}

int met2_plan_set_lambda_grid(met2_plan *p, const double *lam, int32_t n)
{
    if (!p || !lam || n < 3) return fail(MET2_E_INVALID, "bad lambda grid");
    if (n > 64) return fail(MET2_E_UNSUPPORTED, "lambda grid longer than 64 points");
    USE_DEVICE(p->opt.device);
    if (p->dLam) HIPCHK(hipFree(p->dLam));
    HIPCHK(hipMalloc(&p->dLam, sizeof(double) * n));
    HIPCHK(hipMemcpy(p->dLam, lam, sizeof(double) * n, hipMemcpyHostToDevice));
    p->nlam = n;
    return MET2_OK;
}

int met2_fit(met2_plan *p, int32_t method, int64_t nvox, const double *data, const double *fa_index, const uint8_t *mask,
             double *fsol, double *sig, double *reg, double *lam, double *maps, int32_t *status, void *stream)
{
    if (!p) return fail(MET2_E_INVALID, "NULL plan");
    return met2_fit_strided(p, method, nvox, data, p->n_te, 1, fa_index, mask, fsol, sig, reg, lam, maps, status, stream);
}

static int fit_impl(met2_plan *p, int32_t method, int64_t nvox, const double *data, int64_t voxel_stride, int64_t echo_stride,
                    const double *fa_index, const uint8_t *mask, double *fsol, double *sig, double *reg, double *lam, double *maps,
                    int32_t *status, void *stream, bool sync);

int met2_fit_strided(met2_plan *p, int32_t method, int64_t nvox, const double *data, int64_t voxel_stride, int64_t echo_stride,
                     const double *fa_index, const uint8_t *mask, double *fsol, double *sig, double *reg, double *lam, double *maps,
                     int32_t *status, void *stream)
{
    return fit_impl(p, method, nvox, data, voxel_stride, echo_stride, fa_index, mask, fsol, sig, reg, lam, maps, status, stream, true);
}

int met2_fit_enqueue_strided(met2_plan *p, int32_t method, int64_t nvox, const double *data, int64_t voxel_stride, int64_t echo_stride,
                             const double *fa_index, const uint8_t *mask, double *fsol, double *sig, double *reg, double *lam, double *maps,
                             int32_t *status, void *stream)
{
    return fit_impl(p, method, nvox, data, voxel_stride, echo_stride, fa_index, mask, fsol, sig, reg, lam, maps, status, stream, false);
}

int met2_plan_finish(met2_plan *p, void *stream)
{
    if (!p) return fail(MET2_E_INVALID, "NULL plan");
    USE_DEVICE(p->opt.device);
    if (p->err_pending && p->err_stream != (hipStream_t)stream)
        return fail(MET2_E_STATE, "met2_plan_finish on another stream than the one the plan's fits were enqueued on");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (p->err_pending) {
        p->err_pending = false;
        const int herr = p->hErr[0];
        p->last_spill = p->hErr[1];
        p->hErr[0] = 0;
        if (herr & 1) return fail(MET2_E_INVALID, "FA index outside the dictionary's flip-angle axis");
    }
    return MET2_OK;
}

static int fit_impl(met2_plan *p, int32_t method, int64_t nvox, const double *data, int64_t voxel_stride, int64_t echo_stride,
                    const double *fa_index, const uint8_t *mask, double *fsol, double *sig, double *reg, double *lam, double *maps,
                    int32_t *status, void *stream, bool sync)
{
    if (!p) return fail(MET2_E_INVALID, "NULL plan");
    if (voxel_stride == 0 || echo_stride == 0) return fail(MET2_E_INVALID, "zero stride");
    if (nvox == 0) return MET2_OK;                       // empty voxel list: nothing to do (pointers may be NULL)
    if (!data || !fsol || !reg) return fail(MET2_E_INVALID, "NULL argument");
    if (nvox < 0 || nvox > 0x7fffffff) return fail(MET2_E_INVALID, "nvox out of range");
    if (!p->have_dict) return fail(MET2_E_STATE, "no dictionary in the plan");
    if (method != MET2_NNLS && !p->have_pen) return fail(MET2_E_STATE, "no penalty matrix set");
    if (maps && !p->have_t2) return fail(MET2_E_STATE, "metrics requested but no T2 grid set");
    const bool objgrid = method >= 10;
    if (objgrid) { method -= 10; if (method != MET2_X2 && method != MET2_GCV && method != MET2_BAYESREG) return fail(MET2_E_INVALID, "no objective for this method"); }
    if (method < 0 || method > MET2_BAYESREG) return fail(MET2_E_INVALID, "unknown method");
    if (objgrid && p->nlam > p->n_t2) return fail(MET2_E_UNSUPPORTED, "objective grid longer than n_t2");
    if (nvox == 0) return MET2_OK;
    USE_DEVICE(p->opt.device);
    hipStream_t s = (hipStream_t)stream;
    if (p->err_pending && p->err_stream != s)
        return fail(MET2_E_STATE, "fits are pending on another stream of this plan: call met2_plan_finish on it first (one plan serves one stream at a time)");
    int rc = ensure_sort_bufs(p, nvox);
    if (rc) return rc;
    // capacity scheme: pass 1 with a passive-set capacity kfast < n (more waves per CU), pass 2 with the full
    // capacity for the voxels that hit it
    // the kernel variant: GCV takes its trace from the 17 x 17 low-rank form when the plan's dictionary allows it (MET2_GCV_FULL=1: test
    // switch, the (m + 1) x (m + 1) form always)
    const int kmeth = (method == MET2_GCV && p->gcv_lr && !getenv("MET2_GCV_FULL")) ? MET2_GCV_LR : method;
    const int kfast = objgrid ? 0 : fast_kmax(p, kmeth);
    LaunchGeom g, g2;
    rc = fit_geometry(p, objgrid ? kmeth + 10 : kmeth, g, kfast, 0, nvox);
    if (rc) return rc;
    if (kfast) { rc = fit_geometry(p, kmeth, g2, 0, 0, nvox); if (rc) return rc; }
    const bool two_pass = getenv("MET2_TWO_PASS") != nullptr;      // test switch (A/B): the two-launch capacity ladder of rounds 1-4 instead of the spill-over slots
    if (kfast && two_pass && !status) {        // the second pass is driven by the status words
        if (p->cap_status < nvox) {
            if (p->dStatus) HIPCHK(hipFree(p->dStatus));
            HIPCHK(hipMalloc(&p->dStatus, sizeof(int32_t) * (size_t)nvox));
            p->cap_status = nvox;
        }
        status = p->dStatus;
    }
    if (!p->have_pen) {   // plain NNLS never touches the bands, but the kernel loads them
        HIPCHK(hipMemsetAsync(p->dKband, 0, sizeof(double) * 5 * 128, s));
        HIPCHK(hipMemsetAsync(p->dLband, 0, sizeof(double) * 5 * 128, s));
        HIPCHK(hipMemsetAsync(p->dKd, 0, sizeof(double) * (size_t)p->n_t2 * p->n_t2, s));
    }
    if (!p->have_t2) HIPCHK(hipMemsetAsync(p->dT2, 0, sizeof(double) * 128, s));
    SortBufs sb = sort_bufs(p);
    // counters and cursors start at zero; the error word (index 4 (nfa + 1) + 1) keeps what earlier ENQUEUED fits may have set
    hipLaunchKernelGGL(reset_sort_kernel, dim3(1), dim3(256), 0, s, sb, p->n_fa, p->err_pending ? 1 : 0);
    const int nb = (int)((nvox + 255) / 256);
    const int nbs = (int)((nvox + MET2_SORT_BLOCK - 1) / MET2_SORT_BLOCK);          // classify / scatter: one global atomic per (workgroup, flip angle)
    const size_t sort_lds = sizeof(int) * 2 * ((size_t)p->n_fa + 1);                // per-flip-angle tables of the sort kernels in LDS
    if (sort_lds > 48 * 1024) {
        if (sort_lds > 150 * 1024) return fail(MET2_E_UNSUPPORTED, "more than 19 000 flip angles (or ROIs) in one plan");
        HIPCHK(hipFuncSetAttribute((const void *)classify_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
        HIPCHK(hipFuncSetAttribute((const void *)scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
        HIPCHK(hipFuncSetAttribute((const void *)scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
    }
    // work-queue granularity: every wave pulls
    // its own voxels -- one at a time for the methods that spend ~1 ms per voxel (X2/L2 on configs[1]: 1 / 2 / 4 / 8 / 16
    // voxels per pull -> 5.37 / 5.32 / 5.22 / 5.04 / 4.71 M voxels/s; HBM writes 1.01 / 0.93 / 0.88 / 0.89 GB because the
    // 8-byte per-voxel outputs of neighbouring voxels then leave from different XCDs), eight at a time for NNLS and
    // T2SPARC, where one atomic per voxel on the queue word would cap the kernel at 23 M voxels/s (8 -> 82 M)
    int chunk = (method <= MET2_T2SPARC) ? 8 : 1;
    chunk = tuning_env("MET2_CHUNK", 1, 1024, chunk);
    const bool dbg = getenv("MET2_DEBUG") != nullptr;
    if (dbg) { HIPCHK(hipStreamSynchronize(s)); fprintf(stderr, "[met2] fit: nvox=%lld grid=%d block=%d lds=%d\n", (long long)nvox, g.grid, g.block, g.lds); fflush(stderr); }
    hipLaunchKernelGGL(classify_kernel, dim3(nbs), dim3(MET2_SORT_BLOCK), sort_lds, s, nvox, p->n_te, p->n_fa, data, voxel_stride, echo_stride, fa_index, mask, 1, sb, status);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(256), sort_lds, s, p->n_fa, chunk, sb);
    if (dbg) { HIPCHK(hipStreamSynchronize(s)); fprintf(stderr, "[met2] classify done\n"); fflush(stderr); }
    hipLaunchKernelGGL(scatter_kernel, dim3(nbs), dim3(MET2_SORT_BLOCK), sort_lds, s, nvox, p->n_fa, sb);
    HIPCHK(hipGetLastError());
    if (dbg) {
        HIPCHK(hipStreamSynchronize(s));
        std::vector<int> hs(4 * (size_t)(p->n_fa + 1) + 8);
        HIPCHK(hipMemcpy(hs.data(), p->dSmall, sizeof(int) * hs.size(), hipMemcpyDeviceToHost));
        fprintf(stderr, "[met2] sort done: fitted=%d chunks=%d\n", hs[2 * (p->n_fa + 1) + p->n_fa], hs[3 * (p->n_fa + 1) + p->n_fa]); fflush(stderr);
    }

    FitArgs A;
    A.n = p->n_t2; A.m = p->n_te; A.nfa = p->n_fa; A.kmax = g.kmax; A.waves = g.waves; A.chunk = chunk; A.wave_doubles = g.wave_doubles;
    A.method = method; A.nlam = p->nlam;
    A.maxfun = p->opt.brent_maxfun > 0 ? p->opt.brent_maxfun : (method == MET2_BAYESREG ? 200 : 300);
    A.x2_factor = p->opt.x2_factor; A.t2sparc_lambda = p->opt.t2sparc_lambda; A.xtol = p->opt.brent_xtol;
    A.cut_m = p->opt.t2_myelin_cut; A.cut_ie = p->opt.t2_ie_cut;
    A.lam_lo = method == MET2_GCV ? p->opt.gcv_lo : (method == MET2_BAYESREG ? p->opt.bayes_lo : p->opt.x2_lo);       // the lambda search's interval
    A.lam_hi = method == MET2_GCV ? p->opt.gcv_hi : (method == MET2_BAYESREG ? p->opt.bayes_hi : p->opt.x2_hi);
    A.log_detL = p->log_detL;
    A.Dfa = p->dD; A.Bfa = p->dB; A.Dtfa = p->dDt; A.Aq = p->dAq; A.kband = p->dKband; A.lband = p->dLband; A.Kd = p->dKd; A.lam_grid = p->dLam; A.t2s = p->dT2;
    A.data = data; A.vs = voxel_stride; A.es = echo_stride; A.sb = sb; A.fsol = fsol; A.sig = sig; A.reg = reg; A.lam = lam; A.maps = maps; A.status = status; A.nvox = nvox;

    A.seed = nullptr;
    const bool no_seed = getenv("MET2_NO_SEED") != nullptr;      // test switch: every voxel grows its first passive set from the lambda = 0 solution
    if (!no_seed && !objgrid && p->have_pen && p->seeds_valid && p->seeds_ok && p->seeds_key[0] == p->opt.t2sparc_lambda &&
        (method == MET2_X2 || method == MET2_GCV || method == MET2_BAYESREG || method == MET2_T2SPARC)) {
        const int slot = method == MET2_BAYESREG ? 1 : (method == MET2_T2SPARC ? 2 : (method == MET2_GCV ? 3 : 0));      // records built by ensure_seeds() when the plan was configured
        A.seed = p->dSeed + sizeof(SeedRec) * (size_t)slot * p->n_fa;
    }
    A.btab = nullptr; A.nbtab = 0; A.btab_stride = p->btab_stride;
    A.chol = nullptr; A.chol_stride = 0;
    if (method == MET2_BAYESREG && !objgrid && g.nb == 2) {
        // chol_lean's scratch: one packed factor per wave of the widest launch (first pass; the clean-up pass has fewer waves)
        const int stride = (col_base(p->n_t2) + 15) & ~15;
        const int64_t need = (int64_t)g.grid * (kfast ? std::max(g.waves, g2.waves) : g.waves) * stride;
        if (p->cap_chol < need) {
            if (p->dChol) { HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipFree(p->dChol)); p->dChol = nullptr; p->cap_chol = 0; }
            HIPCHK(hipMalloc(&p->dChol, sizeof(double) * (size_t)need));
            p->cap_chol = need;
        }
        A.chol = p->dChol; A.chol_stride = stride;
    }
    // spill-over slots (nnls_big.hpp): a set that outgrows the LDS capacity of the launch goes on in place, its columns beyond the capacity in
    // the wave's slot -- one launch per fit.
    A.big = nullptr; A.big_stride = 0; A.all_queued = 0;
    if (kfast && !two_pass) {
        const int stride = (col_base(p->n_t2) - col_base(g.kmax) + 15) & ~15;
        LaunchGeom gw;                                                      // sized for the widest launch of this shape (a short voxel list runs fewer waves):
        rc = fit_geometry(p, kmeth, gw, kfast, 0, -1);                      // the slots are allocated once, not per block of a host pipeline
        if (rc) return rc;
        const int64_t need = (int64_t)gw.grid * gw.waves * stride;
        if (p->cap_big < need) {
            if (p->dBig) { HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipFree(p->dBig)); p->dBig = nullptr; p->cap_big = 0; }
            HIPCHK(hipMalloc(&p->dBig, sizeof(double) * (size_t)need));
            p->cap_big = need;
        }
        A.big = p->dBig; A.big_stride = stride;
    }
    A.lc_save = nullptr; A.lc_at = nullptr; A.lc_cap = 0;
    A.spill_w2 = getenv("MET2_SPILL_W2") ? atoi(getenv("MET2_SPILL_W2")) : 0;      // test switch (A/B)
    if (A.big && method == MET2_LCURVE && g.nb == 2 && !objgrid && !getenv("MET2_LC_RESTART")) {
        // records for a sixteenth of the voxels (5 % of them overflow on measured spectra), between 4 096 and 262 144 (2.4 GB): entries beyond start over
        const int64_t cap = std::min<int64_t>(nvox, std::min<int64_t>(262144, std::max<int64_t>(4096, nvox / 16)));
        const int64_t need = cap * (LC_SAVE_DOUBLES * 64) + (cap + 1) / 2;
        if (p->cap_lc < need) {
            if (p->dLcSave) { HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipFree(p->dLcSave)); p->dLcSave = nullptr; p->cap_lc = 0; }
            HIPCHK(hipMalloc(&p->dLcSave, sizeof(double) * (size_t)need));
            p->cap_lc = need;
        }
        A.lc_save = p->dLcSave; A.lc_at = (int *)(p->dLcSave + cap * (LC_SAVE_DOUBLES * 64)); A.lc_cap = (int)cap;
    }
    const bool ladder = kfast && !A.big;
    for (int j = 0; j < (MET2_BAYES_TABLE > 0 ? MET2_BAYES_TABLE : 1); ++j) A.blam[j] = p->blam[j];
    if (MET2_BAYES_TABLE > 0 && method == MET2_BAYESREG && !objgrid && p->dBtab && p->seeds_valid && p->seeds_ok && !tuning_env("MET2_NO_BAYES_TABLE", 0, 1, 0)) {
        A.btab = p->dBtab; A.nbtab = MET2_BAYES_TABLE;
    }
    // The fit kernels carry the reference's lambda-search intervals as literals (fit_kernel.hpp: FitArgs::lam_lo); a plan whose options name other
    // intervals runs every voxel through the spill-over kernel, whose instance of the voxel routine reads them -- correct, and slower (the solver is
    // a function call there).
    const bool custom_iv = !objgrid && ((method == MET2_X2 && (p->opt.x2_lo != 0.0 || p->opt.x2_hi != 10.0)) || (method == MET2_GCV && (p->opt.gcv_lo != 1e-8 || p->opt.gcv_hi != 10.0)) ||
                                        (method == MET2_BAYESREG && (p->opt.bayes_lo != 1e-8 || p->opt.bayes_hi != 2.0)));
    HIPCHK(hipEventRecord(p->ev0, s));
    if (objgrid) { A.sig = nullptr; A.maps = nullptr; A.lam = nullptr; }
    if (custom_iv) {
        if (two_pass) return fail(MET2_E_UNSUPPORTED, "MET2_TWO_PASS (A/B switch) does not go with non-default lambda-search intervals");
        A.all_queued = 1;
        HIPCHK(hipEventRecord(p->ev1, s));
        rc = launch_method(kmeth, A, g, s, true);
        if (rc) return rc;
        HIPCHK(hipEventRecord(p->ev2, s));
        p->timed2 = true;
    } else {
        rc = launch_method(objgrid ? kmeth + 10 : kmeth, A, g, s);
        if (rc) return rc;
        HIPCHK(hipEventRecord(p->ev1, s));
    }
    if (custom_iv) {
    } else if (A.big) {
        // the spill-over kernel: the voxels the first kernel queued (their passive set outgrew the LDS capacity), same geometry, the solver with its
        // spill-over legs -- no re-sort, no second capacity; it finds an empty queue in most launches at one bin per lane
        rc = launch_method(kmeth, A, g, s, true);
        if (rc) return rc;
        HIPCHK(hipEventRecord(p->ev2, s));
        p->timed2 = true;
    } else if (ladder) {
        // gated-out voxels are finalised from the first pass's keys, then the key/perm buffers are reused
        hipLaunchKernelGGL(finalize_unfitted_kernel, dim3(p->cus * 4), dim3(256), 0, s, nvox, p->n_t2, p->n_te, p->dKey, mask, fsol,
                           sig, reg, lam, maps);
        hipLaunchKernelGGL(reset_sort_kernel, dim3(1), dim3(256), 0, s, sb, p->n_fa, 1);              // all but the error word
        hipLaunchKernelGGL(requeue_overflow_kernel, dim3(nb), dim3(256), 0, s, nvox, fa_index, status, sb);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(256), sort_lds, s, p->n_fa, chunk, sb);
        hipLaunchKernelGGL(scatter_kernel, dim3(nbs), dim3(MET2_SORT_BLOCK), sort_lds, s, nvox, p->n_fa, sb);
        HIPCHK(hipGetLastError());
        FitArgs A2 = A;
        const int kmid = mid_kmax(p, kmeth, kfast);
        if (kmid) {                                         // middle rung: same kernel at capacity kmid; what still overflows is queued once more
            LaunchGeom gm;
            rc = fit_geometry(p, kmeth, gm, kmid, 0, -1);
            if (rc) return rc;
            A2.kmax = gm.kmax; A2.waves = gm.waves; A2.wave_doubles = gm.wave_doubles;
            rc = launch_method(kmeth, A2, gm, s);
            if (rc) return rc;
            hipLaunchKernelGGL(reset_sort_kernel, dim3(1), dim3(256), 0, s, sb, p->n_fa, 1);
            hipLaunchKernelGGL(requeue_overflow_kernel, dim3(nb), dim3(256), 0, s, nvox, fa_index, status, sb);
            hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(256), sort_lds, s, p->n_fa, chunk, sb);
            hipLaunchKernelGGL(scatter_kernel, dim3(nbs), dim3(MET2_SORT_BLOCK), sort_lds, s, nvox, p->n_fa, sb);
            HIPCHK(hipGetLastError());
        }
        A2.kmax = g2.kmax; A2.waves = g2.waves; A2.wave_doubles = g2.wave_doubles;
        rc = launch_method(kmeth, A2, g2, s);
        if (rc) return rc;
        HIPCHK(hipEventRecord(p->ev2, s));
        p->timed2 = true;
    } else p->timed2 = false;
    p->timed = true;
    if (dbg) {
        HIPCHK(hipStreamSynchronize(s)); fprintf(stderr, "[met2] fit kernel done\n");
#ifdef MET2_CYCSTATS
        unsigned long long cy[16];
        HIPCHK(hipMemcpyFromSymbol(cy, HIP_SYMBOL(met2::g_cyc), sizeof(cy)));
        fprintf(stderr, "[met2] gcv trace: gram(mfma)=%llu tridiag=%llu bisect=%llu weights=%llu\n", cy[8], cy[9], cy[10], cy[11]);
        fprintf(stderr, "[met2] calls: warm solves=%llu duals=%llu append rounds=%llu inner loops after an append=%llu\n", cy[12], cy[13], cy[14], cy[15]);
        fprintf(stderr, "[met2] wave cycles: voxel=%llu refactor=%llu inner=%llu dual=%llu append=%llu | slots 5-7 (x2: sse, removals, removal cycles; bayes: chol, upper_times, erf/log; gcv small path: cycles, evaluations, sweeps; append slot += sum k)=%llu %llu %llu\n",
                cy[0], cy[1], cy[2], cy[3], cy[4], cy[5], cy[6], cy[7]);
        if (method == MET2_X2) {
            static unsigned long long ev[40][12];
            HIPCHK(hipMemcpyFromSymbol(ev, HIP_SYMBOL(met2::g_ev), sizeof(ev)));
            fprintf(stderr, "[met2] X2 per Brent evaluation index: evals canonical | cycles per eval: refactor inner(incl. removal) dual append sse | removals/eval removal-cycles/eval | k start -> end\n");
            for (int e = 0; e < 40; ++e) {
                if (!ev[e][0]) continue;
                const double c = (double)ev[e][0];
                fprintf(stderr, "[met2]  ev %2d: %9llu %9llu | %8.0f %8.0f %8.0f %8.0f %8.0f | %5.2f %8.0f | %5.1f -> %5.1f\n", e, ev[e][0], ev[e][10], ev[e][1] / c, ev[e][2] / c,
                        ev[e][3] / c, ev[e][4] / c, ev[e][5] / c, ev[e][6] / c, ev[e][7] / c, ev[e][8] / c, ev[e][9] / c);
            }
        }
#endif
#ifdef MET2_BIGSTATS
        { unsigned long long bs[16]; HIPCHK(hipMemcpyFromSymbol(bs, HIP_SYMBOL(met2::g_bigstats), sizeof(bs)));
          fprintf(stderr, "[met2] spill-over voxels: solver calls=%llu, with spill-over legs=%llu | cycles plain=%llu spill=%llu (refactor in slot %llu) | appends=%llu removals=%llu refactors=%llu | k start sum=%llu end sum=%llu\n",
                  bs[0], bs[1], bs[2], bs[3], bs[9], bs[4], bs[5], bs[6], bs[8], bs[7]); }
#endif
#ifdef MET2_LOOPSTATS
        int ls[8];
        HIPCHK(hipMemcpyFromSymbol(ls, HIP_SYMBOL(met2::g_loopstats), sizeof(ls)));
        fprintf(stderr, "[met2] loop maxima: tries=%d sweeps=%d outer=%d iter=%d taken=%d round=%d jacobi_sweeps=%d last_sweep_rotations=%d\n", ls[0], ls[1], ls[2], ls[3], ls[4], ls[5], ls[6], ls[7]);
#endif
        fflush(stderr);
    }
    if (!ladder) {
        hipLaunchKernelGGL(finalize_unfitted_kernel, dim3(p->cus * 4), dim3(256), 0, s, nvox, p->n_t2, p->n_te, p->dKey, mask, fsol,
                           objgrid ? nullptr : sig, reg, objgrid ? nullptr : lam, objgrid ? nullptr : maps);
        HIPCHK(hipGetLastError());
    }
    // FA index range errors (IndexError in the reference): the error word lands in the plan's pinned host word; the blocking entries
    // wait for it here, an enqueued fit leaves it to met2_plan_finish (errors of several enqueued fits accumulate: the kernel ORs)
    HIPCHK(hipMemcpyAsync(p->hErr, sb.err, 4 * sizeof(int), hipMemcpyDeviceToHost, s));      // the error word and the spill-over queue.s tail and head
    p->err_pending = true; p->err_stream = s;
    if (!sync) return MET2_OK;
    return met2_plan_finish(p, stream);
}

int met2_fa_bruteforce(met2_plan *p, int64_t nvox, const double *data, const uint8_t *mask, double *fa_index, double *km,
                       double *resid, void *stream)
{
    if (!p) return fail(MET2_E_INVALID, "NULL plan");
    return met2_fa_bruteforce_strided(p, nvox, data, p->n_te, 1, mask, fa_index, km, resid, stream);
}

int met2_fa_bruteforce_strided(met2_plan *p, int64_t nvox, const double *data, int64_t voxel_stride, int64_t echo_stride,
                               const uint8_t *mask, double *fa_index, double *km, double *resid, void *stream)
{
    if (!p) return fail(MET2_E_INVALID, "NULL plan");
    if (nvox == 0) return MET2_OK;                       // empty voxel list: nothing to do (pointers may be NULL)
    if (nvox < 0 || !data || !fa_index) return fail(MET2_E_INVALID, "NULL argument");
    if (voxel_stride == 0 || echo_stride == 0) return fail(MET2_E_INVALID, "zero stride");
    if (!p->have_dict) return fail(MET2_E_STATE, "no dictionary in the plan");
    USE_DEVICE(p->opt.device);
    hipStream_t s = (hipStream_t)stream;
    LaunchGeom g;
    // plain NNLS stops at min(n_t2, n_te) passive bins (Lawson-Hanson's k >= rows test), so the factor needs that
    // capacity only: 4 KB per wave at nTE = 32 instead of 14.6 KB; every wave walks the flip angles on its own, 16 waves per CU.
    const int kcap = p->n_te < p->n_t2 ? p->n_te : 0;
    int rc = fit_geometry(p, MET2_NNLS, g, kcap, 16);
    if (rc) return rc;
    const int fa_waves = g.waves >= 16 ? 16 : (g.waves >= 8 ? 8 : g.waves);
    if (g.waves != fa_waves) {
        g.waves = fa_waves; g.block = 64 * fa_waves;
        g.lds = (int)(sizeof(double) * (size_t)g.wave_doubles * fa_waves + 64);
    }
    SortBufs sb = sort_bufs(p);
    FaArgs A;
    A.n = p->n_t2; A.m = p->n_te; A.nfa = p->n_fa; A.kmax = g.kmax; A.waves = g.waves; A.wave_doubles = g.wave_doubles;
    A.Dfa = p->dD; A.Bfa = p->dB; A.Dtfa = p->dDt; A.Kd = p->dKd; A.data = data; A.vs = voxel_stride; A.es = echo_stride; A.mask = mask; A.fa_index = fa_index; A.km = km; A.resid = resid;
    A.queue = sb.queue; A.nvox = nvox;
    // h = D_fa^T b of all flip angles by one MFMA GEMM per pass of voxels (fa_project_kernel) when the plan has more than a handful
    // of flip angles; the pass size bounds the scratch (8 nfa n bytes per voxel: 87 KB at 91 x 120)
    // Mh: doubles of H per voxel -- h of every flip angle, and behind it (when the walk prunes) the 16 coefficients of the voxel in every
    // flip angle's low-rank basis.  Pruning needs the basis to span the dictionary (p->gcv_lr), all residuals NOT to be asked for, and the
    // bounds of all angles to fit two per lane; MET2_FA_NOPRUNE=1: test switch.
    const bool prune = p->gcv_lr && !resid && p->n_fa >= 8 && p->n_fa <= 128 && !getenv("MET2_FA_NOPRUNE");      // (<= 128: the bounds are formed two per lane)
    // round 5: a pruning walk forms its h from the low-rank basis (fa_kernel), so its scratch is the 16 coefficients per angle and the bounds --
    // 12.4 KB per voxel at 91 angles where H took 87 KB more; the exhaustive walk (all residuals asked for; a dictionary the basis does not span) keeps H
    const int64_t Mh0 = prune ? 0 : (int64_t)p->n_fa * p->n_t2;
    const int64_t Mh = Mh0 + (prune ? (int64_t)p->n_fa * (MET2_GCV_LR_RANK + 1) : 0);
    const bool gemm = p->n_fa >= 8 && tuning_env("MET2_FA_GEMM", 0, 1, 1) != 0;
    int64_t pass = nvox;
    if (gemm) {
        // scratch: at most 6 GiB and at most a quarter of what the device has free right now (several plans or ranks may share it);
        // a failed allocation halves the pass instead of failing the call
        size_t mem_free = 0, mem_total = 0;
        HIPCHK(hipMemGetInfo(&mem_free, &mem_total));
        const int64_t budget = std::min<int64_t>((int64_t)6 << 30, (int64_t)(mem_free / 4) + (int64_t)sizeof(double) * p->cap_h);
        pass = std::max<int64_t>(32, std::min<int64_t>(nvox, (budget / (8 * Mh)) & ~(int64_t)31));
        if (p->cap_h >= 32 * Mh && p->cap_h < pass * Mh && p->cap_h >= std::min<int64_t>(nvox, 65536) * Mh)
            pass = (p->cap_h / Mh) & ~(int64_t)31;                                  // what the plan already holds serves 64 Ki voxels a pass: keep it
        if (p->cap_h < pass * Mh) {
            if (p->dH) { HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipFree(p->dH)); p->dH = nullptr; p->cap_h = 0; }
            for (;;) {
                const hipError_t e = hipMalloc(&p->dH, sizeof(double) * (size_t)(pass * Mh));
                if (e == hipSuccess) break;
                (void)hipGetLastError();
                p->dH = nullptr;
                if (pass <= 32) return fail(MET2_E_HIP, std::string("hipMalloc of the FA walk's scratch: ") + hipGetErrorString(e));
                pass = std::max<int64_t>(32, (pass / 2) & ~(int64_t)31);
            }
            p->cap_h = pass * Mh;
        }
    }
    HIPCHK(hipEventRecord(p->ev0, s));
#define MET2_FA_LAUNCH(VPW, NB, WAVES)                                                                                   \
    do {                                                                                                                \
        if (A.Hq) {                                                                                                     \
            HIPCHK(hipFuncSetAttribute((const void *)fa_kernel<VPW, NB, WAVES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds)); \
            hipLaunchKernelGGL((fa_kernel<VPW, NB, WAVES, true>), dim3(g.grid), dim3(g.block), g.lds, s, A);              \
        } else {                                                                                                        \
            HIPCHK(hipFuncSetAttribute((const void *)fa_kernel<VPW, NB, WAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds)); \
            hipLaunchKernelGGL((fa_kernel<VPW, NB, WAVES, false>), dim3(g.grid), dim3(g.block), g.lds, s, A);             \
        }                                                                                                               \
    } while (0)
    for (int64_t v0 = 0; v0 < nvox; v0 += pass) {
        A.v0 = v0; A.v_end = std::min<int64_t>(nvox, v0 + pass); A.H = nullptr; A.Hq = nullptr; A.Lb = nullptr; A.Aq = p->dAq;
        HIPCHK(hipMemsetAsync(sb.queue, 0, sizeof(int), s));
        if (gemm) {
            FaGemmArgs G;
            G.n = p->n_t2; G.m = p->n_te; G.nfa = p->n_fa; G.Dtfa = p->dDt; G.data = data; G.vs = voxel_stride; G.es = echo_stride; G.H = p->dH;
            G.v0 = v0; G.v_end = A.v_end;
            const int64_t waves = (A.v_end - v0 + 31) / 32;
            const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
            const int ks = (p->n_te + 3) / 4;
            if (prune) {                                                            // the contraction on the stacked bases: [T x m] . [m x nfa 16]
                G.n = MET2_GCV_LR_RANK; G.Dtfa = p->dQt;
                A.Hq = p->dH; A.Lb = p->dH + (size_t)pass * p->n_fa * MET2_GCV_LR_RANK;
            } else A.H = p->dH;
            if (ks <= 8)       hipLaunchKernelGGL(fa_project_kernel<8>, grid, block, 0, s, G);
            else if (ks <= 12) hipLaunchKernelGGL(fa_project_kernel<12>, grid, block, 0, s, G);
            else               hipLaunchKernelGGL(fa_project_kernel<16>, grid, block, 0, s, G);
            HIPCHK(hipGetLastError());
        }
        if (g.nb == 1) { if (g.waves == 16) MET2_FA_LAUNCH(2, 1, 16); else MET2_FA_LAUNCH(4, 1, 8); }
        else           { if (g.waves == 16) MET2_FA_LAUNCH(1, 2, 16); else MET2_FA_LAUNCH(2, 2, 8); }   // two voxels per wave at two bins per lane: same 57 ms (measured)
    }
#undef MET2_FA_LAUNCH
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(p->ev1, s));
#ifdef MET2_CYCSTATS
    if (getenv("MET2_DEBUG")) {
        HIPCHK(hipStreamSynchronize(s));
        unsigned long long cy[8];
        HIPCHK(hipMemcpyFromSymbol(cy, HIP_SYMBOL(met2::g_cyc), sizeof(cy)));
        fprintf(stderr, "[met2] fa wave cycles: solves=%llu barrier-wait=%llu staging=%llu | refactor=%llu inner=%llu dual=%llu append=%llu\n", cy[0], cy[5], cy[6], cy[1], cy[2], cy[3], cy[4]);
    }
#endif
    p->timed = true;
    return MET2_OK;
}

int met2_roi_reduce(met2_plan *src, met2_plan *dst, int64_t nvox, const double *data, int64_t voxel_stride, int64_t echo_stride,
                    const int32_t *roi_index, const double *fa_index, double *mean_signal, double *count, void *stream)
{
    if (!src || !dst || !data || !roi_index || !mean_signal || !count) return fail(MET2_E_INVALID, "NULL argument");
    if (!src->have_dict) return fail(MET2_E_STATE, "no dictionary in the source plan");
    if (src->n_te != dst->n_te || src->n_t2 != dst->n_t2) return fail(MET2_E_INVALID, "source and destination plans differ in shape");
    if (src->opt.device != dst->opt.device) return fail(MET2_E_INVALID, "source and destination plans live on different devices");
    if (voxel_stride == 0 || echo_stride == 0) return fail(MET2_E_INVALID, "zero stride");
    if (nvox <= 0) return fail(MET2_E_INVALID, "empty voxel list");
    USE_DEVICE(src->opt.device);
    hipStream_t s = (hipStream_t)stream;
    RoiArgs A;
    A.nte = src->n_te; A.nt2 = src->n_t2; A.nfa = src->n_fa; A.nroi = dst->n_fa;
    A.nslice = (int)std::min<int64_t>(256, (nvox + 4095) / 4096);
    A.nvox = nvox; A.vs = voxel_stride; A.es = echo_stride; A.data = data; A.roi = roi_index; A.fa_index = fa_index;
    A.Dsrc = src->dD; A.Ddst = dst->dD; A.mean_sig = mean_signal; A.count = count;
    void *scratch = nullptr;
    const size_t nsig = sizeof(double) * (size_t)A.nroi * A.nslice * 64, ncnt = sizeof(int) * (size_t)A.nroi * A.nslice * A.nfa;
    HIPCHK(hipMalloc(&scratch, nsig + ncnt + 16));
    A.part_sig = (double *)scratch; A.part_cnt = (int *)((char *)scratch + nsig); A.err = (int *)((char *)scratch + nsig + ncnt);
    HIPCHK(hipMemsetAsync(A.err, 0, sizeof(int), s));
    hipLaunchKernelGGL(roi_partial_kernel, dim3(A.nslice, A.nroi), dim3(64), sizeof(int) * A.nfa, s, A);
    hipLaunchKernelGGL(roi_finish_kernel, dim3(A.nroi), dim3(256), sizeof(int) * A.nfa, s, A);
    HIPCHK(hipGetLastError());
    int rc = build_gram(dst, s);
    int herr = 0;
    HIPCHK(hipMemcpyAsync(&herr, A.err, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(scratch));
    if (rc) return rc;
    rc = ensure_seeds(dst, s);
    if (rc) return rc;
    if (herr & 1) return fail(MET2_E_INVALID, "FA index outside the dictionary's flip-angle axis");
    return MET2_OK;
}

int met2_metrics(met2_plan *p, int64_t nvox, const double *fsol, const uint8_t *mask, double *maps, void *stream)
{
    if (!p || !fsol || !maps) return fail(MET2_E_INVALID, "NULL argument");
    if (!p->have_t2) return fail(MET2_E_STATE, "no T2 grid set");
    if (nvox <= 0) return MET2_OK;
    USE_DEVICE(p->opt.device);
    int blocks = (int)((nvox + 3) / 4);
    if (blocks > p->cus * 16) blocks = p->cus * 16;
    if (p->n_t2 <= 64)
        hipLaunchKernelGGL(metrics_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, nvox, p->n_t2, p->dT2, p->opt.t2_myelin_cut,
                           p->opt.t2_ie_cut, fsol, mask, maps);
    else
        hipLaunchKernelGGL(metrics_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, nvox, p->n_t2, p->dT2, p->opt.t2_myelin_cut,
                           p->opt.t2_ie_cut, fsol, mask, maps);
    HIPCHK(hipGetLastError());
    return MET2_OK;
}

int met2_plan_last_kernel_ms(met2_plan *p, double *ms)
{
    if (!p || !ms) return fail(MET2_E_INVALID, "NULL argument");
    if (!p->timed) return fail(MET2_E_STATE, "no timed launch yet");
    USE_DEVICE(p->opt.device);
    HIPCHK(hipEventSynchronize(p->ev1));
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, p->ev0, p->ev1));
    *ms = (double)f;
    return MET2_OK;
}

int met2_plan_last_spill_count(met2_plan *p, int64_t *count)
{
    if (!p || !count) return fail(MET2_E_INVALID, "NULL argument");
    *count = p->last_spill;
    return MET2_OK;
}

int met2_plan_last_second_pass_ms(met2_plan *p, double *ms)
{
    if (!p || !ms) return fail(MET2_E_INVALID, "NULL argument");
    *ms = 0.0;
    if (!p->timed || !p->timed2) return MET2_OK;
    USE_DEVICE(p->opt.device);
    HIPCHK(hipEventSynchronize(p->ev2));
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, p->ev1, p->ev2));
    *ms = (double)f;
    return MET2_OK;
}

int met2_plan_gcv_form(met2_plan *p, int32_t *low_rank, double *residual)
{
    if (!p) return fail(MET2_E_INVALID, "NULL plan");
    if (!p->have_dict) return fail(MET2_E_STATE, "no dictionary in the plan");
    if (low_rank) *low_rank = (p->gcv_lr && !getenv("MET2_GCV_FULL")) ? 1 : 0;
    if (residual) *residual = p->gcv_res;
    return MET2_OK;
}

int met2_plan_launch_info(met2_plan *p, int32_t method, int32_t *grid, int32_t *block, int32_t *lds_bytes)
{
    if (!p) return fail(MET2_E_INVALID, "NULL plan");
    LaunchGeom g;
    if (method == MET2_GCV && p->gcv_lr && !getenv("MET2_GCV_FULL")) method = MET2_GCV_LR;      // the variant fit_impl launches
    const int kf = fast_kmax(p, method);
    int rc = fit_geometry(p, method, g, kf);
    if (rc) return rc;
    if (grid) *grid = g.grid;
    if (block) *block = g.block;
    if (lds_bytes) *lds_bytes = g.lds;
    return MET2_OK;
}

} // extern "C"
