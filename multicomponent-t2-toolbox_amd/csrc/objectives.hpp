// objectives.hpp -- lambda-selection objectives that need more than the NNLS solve:
//   BayesReg  bayesian_interpolation.py:107-126  full n x n Cholesky of beta (B + lambda K), erf, log
//   GCV       algorithms.py:285-296              truncated pseudo-inverse of the support Gram matrix: MFMA Gram contraction,
//                                                Householder tridiagonalisation, bisection, eigenvector weights
// Both reuse the wave's LDS region once the NNLS solution is in st.x (the factor is rebuilt by the
// next warm start).
#pragma once
#include "nnls_wave.hpp"

namespace met2 {

__device__ __forceinline__ double op_mul(double a, double b) { return a * b; }
__device__ __forceinline__ double wave_prod(double v)
{
    MET2_ROW_REDUCE(v, op_mul)
    return (bcast(v, 0) * bcast(v, 16)) * (bcast(v, 32) * bcast(v, 48));
}

// Upper Cholesky factor U (A = U^T U) of A = beta*B + (beta*lam)*K, packed by columns into S.R like the solver's factor
// (entry (r, c) at col_base(c) + r; needs kmax == n), by a blocked right-looking Cholesky with the trailing update on the
// matrix cores (the row-wise routine of round 1 measured 11 % slower on configs[3] and has been removed).  A = beta B +
// beta lam K is first written into the wave's LDS region (upper triangle, packed by columns like U).  Then, per block row of 16:
//   (a) its rows are finished row by row, lane <-> column, with the inner products restricted to the rows of the block
//       (<= 15 terms instead of up to n - 1: everything above the block has already been subtracted by the trailing updates);
//   (b) every trailing 16 x 16 tile C(ti, tj), ti <= tj, takes C -= U(block, ti)^T U(block, tj) as four v_mfma_f64_16x16x4
//       (operands one f64 per lane straight from the packed columns, accumulator = the tile, four f64 per lane).
// n = 60: 10 tile updates = 40 MFMAs and <= 15-term row sums instead of 59-term ones; n = 120: 84 tile updates = 336 MFMAs.
template <int NB>
__device__ __forceinline__ bool chol_full(const WaveShared &S, const Band<NB> &bd, double beta, double lam, int lane, double &det_u)
{
    lane = lane_opaque(lane);
    const int n = S.n;
    const double bl = beta * lam;
    int cbl[NB], cbc[NB];
    unsigned jc[NB];
    double diag[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        cbl[b] = col_base(pl);
        cbc[b] = col_base(min(pl, n - 1));
        jc[b] = (unsigned)min(pl, n - 1);
        diag[b] = 1.0;
    }
    // ---- A into LDS, four rows of B and K in flight
    for (int j = 0; j < n; j += 4) {
        double vb[4][NB], vk[4][NB];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jj = min(j + q, n - 1);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                vb[q][b] = ld_row_sel(S.buffer_rows, S.B, jj * S.bstride, jc[b]);
                vk[q][b] = ld_row_sel(S.buffer_rows, S.K, jj * n, jc[b]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int c = lane + 64 * b;
                if (j + q < n && c >= j + q && c < n) S.R[cbl[b] + j + q] = fma(bl, vk[q][b], beta * vb[q][b]);
            }
    }
    __builtin_amdgcn_wave_barrier();
    const int li = lane & 15, lk = lane >> 4;
    const int nt = (n + 15) >> 4;
    for (int kb = 0; kb < nt; ++kb) {
        const int r0 = 16 * kb, r1 = min(n, r0 + 16);
        // (a) the block's rows, two at a time as in refactor(): rows r and r + 1 share the reads of the block's rows above them and
        //     row r + 1 takes row r's term from registers -- the block is a chain of dependent LDS round trips (one per finished
        //     row when done singly: 59 % of the BayesReg kernel's wave cycles at 32 x 60), pairs and four-row reads halve their number
        auto finish = [&](int r, double (&a)[NB], double (&u)[NB]) -> bool {
            const double d = bcastN<NB>(a, r);
            const double rinv = rsqrt_nr(d);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int c = lane + 64 * b;
                u[b] = a[b] * rinv;                                   // lane r: d * rinv = U[r][r]
                if (c >= r && c < n) S.R[cbl[b] + r] = u[b];
                if (c == r) diag[b] = u[b];
            }
            return d > 0.0;                                           // scipy raises LinAlgError otherwise
        };
        int r = r0;
        int cbr = col_base(r0);
        for (; r + 1 < r1; r += 2) {                                  // r is even (r0 is a multiple of 16)
            double a[NB], c2[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int c = lane + 64 * b;
                double a0, c0;
                lds_pair(S.R + cbc[b] + r, a0, c0);                   // rows r, r + 1 of the lane's column
                a[b] = (c >= r && c < n) ? a0 : 0.0;
                c2[b] = (c > r && c < n) ? c0 : 0.0;
            }
            const double *cr = S.R + cbr, *cr1 = cr + col_len(r);     // columns r and r + 1
            int j = r0;
#pragma clang loop unroll(disable)
            for (; j + 4 <= r; j += 4) {
                double s0, s1, s2, s3, t0, t1, t2, t3;
                lds_quad(cr + j, s0, s1, s2, s3);
                lds_quad(cr1 + j, t0, t1, t2, t3);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    double q0, q1, q2, q3;
                    lds_quad(S.R + cbc[b] + j, q0, q1, q2, q3);
                    a[b] = fma(-s0, q0, a[b]); c2[b] = fma(-t0, q0, c2[b]);
                    a[b] = fma(-s1, q1, a[b]); c2[b] = fma(-t1, q1, c2[b]);
                    a[b] = fma(-s2, q2, a[b]); c2[b] = fma(-t2, q2, c2[b]);
                    a[b] = fma(-s3, q3, a[b]); c2[b] = fma(-t3, q3, c2[b]);
                }
            }
            if (j < r) {                                              // r - r0 is even: two rows left
                double s0, s1, t0, t1;
                lds_pair(cr + j, s0, s1);
                lds_pair(cr1 + j, t0, t1);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    double q0, q1;
                    lds_pair(S.R + cbc[b] + j, q0, q1);
                    a[b] = fma(-s0, q0, a[b]); c2[b] = fma(-t0, q0, c2[b]);
                    a[b] = fma(-s1, q1, a[b]); c2[b] = fma(-t1, q1, c2[b]);
                }
            }
            double u[NB], w[NB];
            if (!finish(r, a, u)) return false;
            const double su = bcastN<NB>(u, r + 1);                   // U[r][r+1]
#pragma unroll
            for (int b = 0; b < NB; ++b) c2[b] = fma(-su, u[b], c2[b]);
            if (!finish(r + 1, c2, w)) return false;
            __builtin_amdgcn_wave_barrier();
            cbr += col_len(r) + col_len(r + 1);
        }
        if (r < r1) {                                                 // odd n: the last row on its own
            double a[NB], u[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { const int c = lane + 64 * b; const double a0 = S.R[cbc[b] + r]; a[b] = (c >= r && c < n) ? a0 : 0.0; }
            const double *cr = S.R + cbr;
            for (int j = r0; j + 2 <= r; j += 2) {
                double s0, s1;
                lds_pair(cr + j, s0, s1);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    double q0, q1;
                    lds_pair(S.R + cbc[b] + j, q0, q1);
                    a[b] = fma(-s0, q0, a[b]);
                    a[b] = fma(-s1, q1, a[b]);
                }
            }
            if (!finish(r, a, u)) return false;
            __builtin_amdgcn_wave_barrier();
        }
        // (b) trailing tiles on the matrix cores
        for (int ti = kb + 1; ti < nt; ++ti) {
            const int ca = 16 * ti + li;                              // this lane's column inside tile column ti
            const int cba = col_base(min(ca, n - 1));
            double aop[4];
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int kr = r0 + 4 * sidx + lk;
                const double v = S.R[cba + min(kr, r1 - 1)];
                aop[sidx] = (ca < n && kr < r1) ? -v : 0.0;
            }
            for (int tj = ti; tj < nt; ++tj) {
                const int cc = 16 * tj + li;
                const int cbb = col_base(min(cc, n - 1));
                double bop[4];
                met2_d4 acc;
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int kr = r0 + 4 * sidx + lk;
                    const double v = S.R[cbb + min(kr, r1 - 1)];
                    bop[sidx] = (cc < n && kr < r1) ? v : 0.0;
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * ti + lk + 4 * v;
                    acc[v] = (cc < n && row <= cc) ? S.R[cbb + row] : 0.0;
                }
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[sidx], bop[sidx], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * ti + lk + 4 * v;
                    if (cc < n && row <= cc) S.R[cbb + row] = acc[v];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    double dp = 1.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) dp *= (lane + 64 * b < n) ? diag[b] : 1.0;
    det_u = wave_prod(dp);
    return true;
}

// The same factor with the LDS holding ONE block row at a time (two bins per lane: the full 120 x 120 triangle is 58 KB, two waves per
// CU; the 16-row panel is 16 KB and shares the region of the solver's factor, eight waves per CU).  Left-looking by block rows of 16:
//   1. the panel P (rows r0 .. r0 + 15 of A = beta B + beta lam K, columns >= r0; column c at P + 17 c: the odd stride keeps a
//      wave-wide read of one row of all columns off the same banks) is filled from the L2-resident rows of B and K;
//   2. every 16 x 16 tile of the panel takes  C -= U(j, block)^T U(j, tile)  from every FINISHED block row j -- four
//      v_mfma_f64_16x16x4 per (tile, j), operands straight from the wave's scratch in global memory G (packed by columns like the
//      plan-level tables: entry (r, c) at col_base(c) + r), the same 336 MFMAs at n = 120 as the right-looking form;
//   3. the block's rows are finished in the panel exactly as in chol_full (a);
//   4. the finished rows go to G, 16 lanes per column (128 contiguous bytes each).
// G is written and read by this wave only; the workgroup-scope fences order its stores before the loads of the next block row and
// of upper_times (one CU, one vector L1).  bayes_objective reads (U f) from G as it does from a table.
// BayesReg/InvT2 at 48 x 120: 260 k -> 742 k voxels/s (profiles/r03_other_ab.txt).  At one bin per lane the 14.6 KB factor already lets 11
// waves share the LDS and 168 VGPRs allow 12: measured the same with and without (223.9 ms per Mi voxels), so chol_full stays there.
constexpr int MET2_PANEL_STRIDE = 17;
__host__ __device__ inline int chol_panel_doubles(int n) { return n * MET2_PANEL_STRIDE; }
template <int NB>
__device__ __forceinline__ bool chol_lean(const WaveShared &S, double beta, double lam, int lane, double &det_u, double *__restrict__ G)
{
    lane = lane_opaque(lane);
    constexpr int PS = MET2_PANEL_STRIDE;
    const int n = S.n;
    const double bl = beta * lam;
    double *P = S.R;
    int pc[NB];
    unsigned jc[NB];
    double diag[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        jc[b] = (unsigned)min(pl, n - 1);
        pc[b] = (int)jc[b] * PS;                                      // an existing column for the unpredicated reads
        diag[b] = 1.0;
    }
    const int li = lane & 15, lk = lane >> 4;
    const int nt = (n + 15) >> 4;
    for (int kb = 0; kb < nt; ++kb) {
        const int r0 = 16 * kb, r1 = min(n, r0 + 16);
        // ---- 1. rows r0 .. r1 - 1 of A into the panel, four rows of B and K in flight
        for (int j = r0; j < r1; j += 4) {
            double vb[4][NB], vk[4][NB];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jj = min(j + q, n - 1);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    vb[q][b] = ld_row_sel(S.buffer_rows, S.B, jj * S.bstride, jc[b]);
                    vk[q][b] = ld_row_sel(S.buffer_rows, S.K, jj * n, jc[b]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int c = lane + 64 * b;
                    if (j + q < r1 && c >= r0 && c < n) P[pc[b] + (j + q - r0)] = fma(bl, vk[q][b], beta * vb[q][b]);
                }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- 2. the finished block rows' contribution, on the matrix cores
        if (kb > 0) {
            const int ca = min(r0 + li, n - 1);                       // this lane's column inside the block's own tile column
            const double *ga = G + col_base(ca) + lk;
            const bool va = r0 + li < n;
            for (int tj = kb; tj < nt; ++tj) {
                const int cc = 16 * tj + li;
                const int ccl = min(cc, n - 1);
                const double *gb = G + col_base(ccl) + lk;
                const bool vbn = cc < n;
                met2_d4 acc;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int rl = lk + 4 * v;                        // row r0 + rl of the panel
                    const double t = P[ccl * PS + rl];
                    acc[v] = (vbn && r0 + rl < r1) ? t : 0.0;
                }
#pragma clang loop unroll_count(2)
                for (int j = 0; j < kb; ++j) {
                    double aop[4], bop[4];
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx) {            // rows 16 j + 4 sidx + lk of the finished factor: always above the columns read
                        const double a = ga[16 * j + 4 * sidx], bq = gb[16 * j + 4 * sidx];
                        aop[sidx] = va ? -a : 0.0;
                        bop[sidx] = vbn ? bq : 0.0;
                    }
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[sidx], bop[sidx], acc, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int rl = lk + 4 * v;
                    if (vbn && r0 + rl < r1) P[ccl * PS + rl] = acc[v];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- 3. the block's rows, two at a time (chol_full (a), on the panel)
        auto finish = [&](int r, double (&a)[NB], double (&u)[NB]) -> bool {
            const double d = bcastN<NB>(a, r);
            const double rinv = rsqrt_nr(d);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int c = lane + 64 * b;
                u[b] = a[b] * rinv;                                   // lane r: d * rinv = U[r][r]
                if (c >= r && c < n) P[pc[b] + (r - r0)] = u[b];
                if (c == r) diag[b] = u[b];
            }
            return d > 0.0;                                           // scipy raises LinAlgError otherwise
        };
        int r = r0;
        for (; r + 1 < r1; r += 2) {                                  // r is even (r0 is a multiple of 16)
            double a[NB], c2[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int c = lane + 64 * b;
                double a0, c0;
                lds_pair(P + pc[b] + (r - r0), a0, c0);               // rows r, r + 1 of the lane's column
                a[b] = (c >= r && c < n) ? a0 : 0.0;
                c2[b] = (c > r && c < n) ? c0 : 0.0;
            }
            const double *cr = P + r * PS, *cr1 = cr + PS;            // columns r and r + 1 (panel rows 0 .. r - r0 - 1 = rows above in the block)
            int j = 0;
            const int ja = r - r0;
#pragma clang loop unroll(disable)
            for (; j + 4 <= ja; j += 4) {
                double s0, s1, s2, s3, t0, t1, t2, t3;
                lds_quad(cr + j, s0, s1, s2, s3);
                lds_quad(cr1 + j, t0, t1, t2, t3);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    double q0, q1, q2, q3;
                    lds_quad(P + pc[b] + j, q0, q1, q2, q3);
                    a[b] = fma(-s0, q0, a[b]); c2[b] = fma(-t0, q0, c2[b]);
                    a[b] = fma(-s1, q1, a[b]); c2[b] = fma(-t1, q1, c2[b]);
                    a[b] = fma(-s2, q2, a[b]); c2[b] = fma(-t2, q2, c2[b]);
                    a[b] = fma(-s3, q3, a[b]); c2[b] = fma(-t3, q3, c2[b]);
                }
            }
            if (j < ja) {                                             // r - r0 is even: two rows left
                double s0, s1, t0, t1;
                lds_pair(cr + j, s0, s1);
                lds_pair(cr1 + j, t0, t1);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    double q0, q1;
                    lds_pair(P + pc[b] + j, q0, q1);
                    a[b] = fma(-s0, q0, a[b]); c2[b] = fma(-t0, q0, c2[b]);
                    a[b] = fma(-s1, q1, a[b]); c2[b] = fma(-t1, q1, c2[b]);
                }
            }
            double u[NB], w[NB];
            if (!finish(r, a, u)) return false;
            const double su = bcastN<NB>(u, r + 1);                   // U[r][r+1]
#pragma unroll
            for (int b = 0; b < NB; ++b) c2[b] = fma(-su, u[b], c2[b]);
            if (!finish(r + 1, c2, w)) return false;
            __builtin_amdgcn_wave_barrier();
        }
        if (r < r1) {                                                 // odd n: the last row on its own
            double a[NB], u[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { const int c = lane + 64 * b; const double a0 = P[pc[b] + (r - r0)]; a[b] = (c >= r && c < n) ? a0 : 0.0; }
            const double *cr = P + r * PS;
            for (int j = 0; j + 2 <= r - r0; j += 2) {
                double s0, s1;
                lds_pair(cr + j, s0, s1);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    double q0, q1;
                    lds_pair(P + pc[b] + j, q0, q1);
                    a[b] = fma(-s0, q0, a[b]);
                    a[b] = fma(-s1, q1, a[b]);
                }
            }
            if (!finish(r, a, u)) return false;
            __builtin_amdgcn_wave_barrier();
        }
        // ---- 4. the finished rows to G: lane -> (column c0 + (lane >> 4), row r0 + (lane & 15))
        for (int c0 = r0; c0 < n; c0 += 4) {
            const int c = c0 + lk, row = r0 + li;
            if (c < n && row < r1 && row <= c) G[col_base(c) + row] = P[c * PS + li];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    double dp = 1.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) dp *= (lane + 64 * b < n) ? diag[b] : 1.0;
    det_u = wave_prod(dp);
    return true;
}

// (U f)_i for a factor packed by columns at `U` (the wave's LDS region, or a plan-level table in global memory); the owner of row i
// gets row i . f.  Only the columns of the passive bins are visited -- f is zero elsewhere (k ~ 20 of n = 60 columns).
template <int NB>
__device__ __forceinline__ void upper_times(const double *U, const NnlsState<NB> &st, int lane, double (&out)[NB])
{
#pragma unroll
    for (int b = 0; b < NB; ++b) out[b] = 0.0;
    const int k = st.k;
#pragma clang loop unroll(disable)
    for (int p = 0; p < k; p += 4) {                                    // four columns in flight (positions past the set carry f = 0)
        double u[4][NB], fj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = bcastN_i<NB>(st.ord, p + q);
            const int cbj = col_base(j);
            const double fv = bcastN<NB>(st.x, j);
            fj[q] = (p + q < k) ? fv : 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int i = lane + 64 * b;
                u[q][b] = U[cbj + min(i, j)];                           // column j: rows 0..j (clamped reads, masked below)
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = bcastN_i<NB>(st.ord, p + q);
#pragma unroll
            for (int b = 0; b < NB; ++b) out[b] = fma((lane + 64 * b <= j) ? u[q][b] : 0.0, fj[q], out[b]);
        }
    }
}

struct BayesCtx {
    double beta, log_detL;
    int failed;     // Cholesky failure seen
};

// Plan-level factors for the shared Brent abscissae (bayes_table_kernel): U0 = chol(B + lambda_j K) and log det U0 per flip angle.
// The matrix the reference factorises at every evaluation, beta (B + lambda K) (bayesian_interpolation.py:113-115), depends on the
// voxel through the scalar beta alone: chol(beta A) = sqrt(beta) chol(A).  scipy's bounded Brent visits the same abscissae
// a + 0.382 (b - a), then the golden-section points towards the lower bound, for every voxel until its first accepted parabolic
// step (6-10 of the ~17 evaluations of a voxel); at those the factor comes from the table: (U f) = sqrt(beta) (U0 f) and
// log det U = log det U0 + (n / 2) log beta -- no factorisation (it was ~50 % of the BayesReg kernel).
struct BayesTable {
    const double *U0;       // packed factor of this (flip angle, abscissa), or NULL: factorise in LDS
    double logdet0;
};

// bayesian_interpolation.py:107-126, given the NNLS solution st.x at lambda = x
template <int NB>
__device__ __forceinline__ double bayes_objective(const WaveShared &S, const Band<NB> &bd, const NnlsState<NB> &st, BayesCtx &bc,
                                                  double x, double b, int lane, BayesTable tab = BayesTable{nullptr, 0.0}, double *G = nullptr)
{
    const int n = S.n, m = S.m;
    const double beta = bc.beta;
    const double ED = 0.5 * sse_of<NB>(S, st, b, lane);
    const double EW = 0.5 * seminorm2<NB>(bd, st.x, n, lane);
    double log_det_u;
    double uf[NB];
#ifdef MET2_CYCSTATS
    NnlsState<NB> &stw = const_cast<NnlsState<NB> &>(st);
    const unsigned long long c0 = __builtin_readcyclecounter();
#endif
    if (tab.U0) {
        upper_times<NB>(tab.U0, st, lane, uf);
        const double sb = sqrt(beta);
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) uf[bb] *= sb;
        log_det_u = tab.logdet0 + 0.5 * (double)n * log(beta);
#ifdef MET2_CYCSTATS
        stw.cyc[6] += __builtin_readcyclecounter() - c0;
#endif
    } else {
        double det_u;
        // G: the wave's scratch in global memory -- the factor is built one block row at a time (two bins per lane)
        if (!(G ? chol_lean<NB>(S, beta, x, lane, det_u, G) : chol_full<NB>(S, bd, beta, x, lane, det_u))) { bc.failed = 1; return NAN; }
#ifdef MET2_CYCSTATS
        const unsigned long long c1 = __builtin_readcyclecounter();
        stw.cyc[5] += c1 - c0;
#endif
        upper_times<NB>(G ? G : S.R, st, lane, uf);
        log_det_u = log(det_u);
#ifdef MET2_CYCSTATS
        stw.cyc[6] += __builtin_readcyclecounter() - c1;
#endif
    }
#ifdef MET2_CYCSTATS
    const unsigned long long c2 = __builtin_readcyclecounter();
#endif
    double term = 0.0;
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) term += (lane + 64 * bb < n) ? log(1.0 + erf((1.0 / sqrt(2.0)) * uf[bb])) : 0.0;
    const double series = wave_sum(term);
#ifdef MET2_CYCSTATS
    stw.cyc[7] += __builtin_readcyclecounter() - c2;
#endif
    const double PI = M_PI;
    double cost1 = beta * ED + beta * x * EW + log_det_u - (n / 2.0) * log(PI / 2.0) - series;
    double cost2 = (m / 2.0) * log(2.0 * PI) - (m / 2.0) * log(beta) + (n / 2.0) * log(PI) - (n / 2.0) * log(2 * beta * x) - bc.log_detL;
    return cost1 + cost2;
}

// ------------------------------------------------------------------------------------------
// GCV (algorithms.py:285-296).  With E = [Dr; sqrt(c) 1^T] ((m+1) x k; Dr = D restricted to the support of the NNLS
// solution, c = lambda * sum_{j in S} L_jj^2 -- the scalar-broadcast quirk of algorithms.py:289-293) one has
// G = Dr^T Dr + c 11^T = E^T E, so for E = U S V^T the truncated pseudo-inverse of np.linalg.lstsq(G, Dr^T, rcond=None)
// gives   trace(Dr G^+ Dr^T) = sum_{kept i} (1 - U[m][i]^2),   kept: sigma_i^2 > eps k sigma_max^2.
// sigma_i^2 and U[m][i] are the eigenvalues of M = E E^T ((m+1) x (m+1), 33 x 33 or 49 x 49 whatever the support size) and the
// m-th components of its eigenvectors.  Direct method, no sweeps:
//   1. M = A A^T with A = E, its sqrt(c) row moved to the front: M = [[c k, sqrt(c) b^T], [sqrt(c) b, C]], b = Dr 1, C = Dr Dr^T.
//      C is the batched Gram contraction of the path, on the matrix cores: v_mfma_f64_16x16x4 tiles whose operands are gathered
//      straight from the L2-resident D^T rows of the support (nothing of E is staged); the symmetric result lands in the wave's LDS
//      region.  b stays in a register per lane: it feeds the first reflector only;
//   2. Householder tridiagonalisation T = Q^T M Q, lane <-> row.  The reflectors never touch index 0, so e_0^T Q = e_0^T and
//      U[m][i] is the FIRST component of the i-th eigenvector of T;
//   3. every lane isolates one eigenvalue of T by bisection on Sturm counts (LAPACK dstebz's recurrence; the midpoints are
//      taken on the IEEE bit patterns, i.e. geometrically, because the spectrum spans 16 decades and the cut sits at its foot);
//   4. the squared first component of an eigenvector is 1 / r_0'(mu) for the continued fraction r_j(x) = x - d_j -
//      e_j^2 / r_{j+1}(x) (chi_T / chi_T' restricted to e_0), one backward recurrence per lane.
// Forming M squares the condition number, which puts the rounding noise of the smallest kept eigenvalues where the
// reference's own SVD of the explicitly formed G has it (measured on 900 (voxel, lambda) pairs in numpy: the objective differs
// from the reference's formula by median 5.4e-6 / p99 6e-4 with this method and by 5.4e-6 / 6e-4 with an exact SVD of E; the
// numerical rank agreed in every one).
// ------------------------------------------------------------------------------------------


// LDS doubles a wave needs for it.  Only C = Dr Dr^T (m x m) is kept in LDS: row and column 0 of M -- sqrt(c) (Dr 1) -- feed the FIRST
// reflector alone and live in a register per lane (round 3; with the 49 x 54 block of M a wave's region was 22.9 KB and a CU held seven;
// 48 x 50 + vectors = 20.4 KB lets it hold the eight waves the kernel is compiled for: DESIGN 6.3).
// Columns: m rounded up to the four-wide sweeps (zero padding, as for the three Householder vectors).  Row stride: >= that, even (every
// row 16-byte aligned: the sweeps read and write with ds_read/write_b128) and = 2 mod 4: then the 16 lanes of a b128 lane group, each
// on its own row, fall on 16 different four-dword bank groups (2 np mod 64 is an odd multiple of 4) -- conflict-free.
// The support list (read by the Gram contraction only) shares the room of the vectors (written after it).
//
// Round 4: the same trace from a 17 x 17 matrix (LR = true).  The dictionary of a flip angle is numerically of low rank -- its singular
// values fall by a factor 4-8 per index (1e-10 at index 16 for 48 x 120, 1e-12 for 32 x 60) -- so with an orthonormal basis Q (m x 16) of
// its dominant column space (gcv_basis_kernel: pivoted Gram-Schmidt, once per plan and flip angle) and A = Q^T D (16 x n) one has
// Dr = Q A_S up to 1e-10 sigma_max, C = Dr Dr^T = Q (A_S A_S^T) Q^T and b = Q (A_S 1): M is blockdiag(1, Q) M' blockdiag(1, Q)^T with
//     M' = [[c k, sqrt(c) beta^T], [sqrt(c) beta, W]],   W = A_S A_S^T (16 x 16: ONE mfma tile),   beta = A_S 1,
// and the eigenvalues of M above the cut and the first components of their eigenvectors are those of M' (the part of Dr outside Q
// is orthogonal to it, so it moves the eigenvalues only in second order: <= 1e-20 sigma_max^2, four decades under the rounding noise
// of the reference's own G).  Numpy on 360 (voxel, lambda) pairs at 48 x 120: |trace(M') - trace(M)| median 8e-10, the numerical
// rank differs in 1 of 360 (an eigenvalue on the cut, where M and the reference's lstsq differ from each other as well), and the
// distance to the reference's formula is the same for both (median 5.8e-5, p90 6.9e-4).  Everything downstream -- tridiagonalisation,
// multisection, weights -- is the code below with m = 16: 15 Householder steps on 16 rows instead of 47 on 48, Sturm chains of 17 rows
// instead of 49, 2.7 KB of LDS instead of 20.4.  A plan whose dictionary is NOT of numerical rank <= 16 keeps the full form.
#define MET2_GCV_LR_RANK 16
__host__ __device__ inline int gcv_cols(int m) { return (m + 3) & ~3; }
__host__ __device__ inline int gcv_row_stride(int m) { int np = gcv_cols(m); while ((np & 3) != 2) ++np; return np; }
__host__ __device__ inline int gcv_vec_len(int m) { return gcv_cols(m); }
__host__ __device__ inline int gcv_lds_doubles(int m, int kcap)
{
    const int vec = 3 * gcv_vec_len(m), lst = (kcap + 1) / 2 + 2;
    return m * gcv_row_stride(m) + (vec > lst ? vec : lst);
}

// Sturm count: number of eigenvalues < x (<= x up to the measure-zero case of an exactly vanishing minor) of the symmetric
// tridiagonal matrix given as (d_j, e_{j-1}^2) pairs, by the sign changes of the leading principal minors
// p_j(x) = (d_j - x) p_{j-1} - e_{j-1}^2 p_{j-2} (the quotients p_j / p_{j-1} are the pivots of LAPACK dstebz's recurrence; products
// instead of divisions keep the dependent chain at two operations per row).  Two registers alternate as p_{j-1} / p_{j-2}, a sign
// change is one xor of the high words, and the pair is rescaled by a power of two every eight rows (per row the minors grow by at
// most the Gershgorin bound and shrink by no more than ~1e-16 of it).
// C shifts per lane at once: the chains are independent (they hide each other's fp64 latency) and share the row broadcasts.
template <int C>
__device__ __forceinline__ void sturm_count(double dreg, double e2reg, int n, const double (&x)[C], int (&out)[C])
{
    // lane j holds (d_j, e_{j-1}^2); a row's pair reaches the scalar registers through v_readlane (no memory round trip)
    // The signs of the minors are shifted into a word, one v_alignbit per row and shift (the xor / shift / add of counting as it goes
    // were 2.5 of the 5.5 instructions per row and shift); the sign changes are counted with one popcount per 24 rows.
    double pa[C], pb[C];                                         // p_{j-1}, p_{j-2}; after a row the roles swap
    unsigned cnt[C], w[C];                                       // w: bit 0 = sign of the latest minor, bit i = sign of the one i rows before
#pragma unroll
    for (int c = 0; c < C; ++c) { pa[c] = 1.0; pb[c] = 0.0; cnt[c] = 0u; w[c] = 0u; }      // p_{-1} = 1: positive
    int nb = 0;                                                  // rows shifted in since the last count
    auto flush = [&]() {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            cnt[c] += (unsigned)__builtin_popcount((w[c] ^ (w[c] >> 1)) & ((1u << nb) - 1u));      // nb adjacent pairs
            w[c] &= 1u;                                          // the latest sign opens the next stretch
        }
        nb = 0;
    };
    auto row2 = [&](int j) {                                     // rows j (-> pb) and j + 1 (-> pa)
        const double d0 = bcast(dreg, j), f0 = bcast(e2reg, j), d1 = bcast(dreg, j + 1), f1 = bcast(e2reg, j + 1);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            pb[c] = fma(d0 - x[c], pa[c], -(f0 * pb[c]));
            w[c] = __builtin_amdgcn_alignbit(w[c], (unsigned)__double2hiint(pb[c]), 31);
            pa[c] = fma(d1 - x[c], pb[c], -(f1 * pa[c]));
            w[c] = __builtin_amdgcn_alignbit(w[c], (unsigned)__double2hiint(pa[c]), 31);
        }
    };
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        row2(j); row2(j + 2); row2(j + 4); row2(j + 6);
        nb += 8;
        if (nb == 24) flush();
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int ex = __builtin_amdgcn_frexp_exp(fmax(fabs(pa[c]), fabs(pb[c])));
            pa[c] = ldexp(pa[c], -ex); pb[c] = ldexp(pb[c], -ex);
        }
    }
    for (; j + 2 <= n; j += 2) { row2(j); nb += 2; }             // at most 6 rows: nb <= 22
    if (j < n) {
        const double d0 = bcast(dreg, j), f0 = bcast(e2reg, j);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            pb[c] = fma(d0 - x[c], pa[c], -(f0 * pb[c]));
            w[c] = __builtin_amdgcn_alignbit(w[c], (unsigned)__double2hiint(pb[c]), 31);
        }
        nb += 1;
    }
    flush();
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = (int)cnt[c];
}

// The tridiagonal form of M for a support, at c = 1, kept from one Brent evaluation to the next: M = [[c k, sqrt(c) b^T],
// [sqrt(c) b, C]] with b = Dr 1 and C = Dr Dr^T.  The first reflector only depends on the direction of b and every later step
// works on H1 C H1, so c enters T through d_0 = c k and e_0 = sqrt(c) e_0(c = 1) alone.  While the support of the solution does
// not change (about half of the evaluations of a voxel) the Gram contraction and the tridiagonalisation -- 39 % of the GCV
// kernel at 48 x 120 -- are skipped and only the bisection and the weights are redone.
#ifndef MET2_GCV_CACHE
#define MET2_GCV_CACHE 2
#endif
template <int NB>
struct GcvCache {                  // a few entries: Brent's last steps often alternate between neighbouring supports
    u64 Sm[MET2_GCV_CACHE][NB];            // supports the cached forms belong to
    double d[MET2_GCV_CACHE], e[MET2_GCV_CACHE]; // lane j: T[j][j], T[j+1][j] at c = 1
    int valid;         // bit e: entry e holds a form
    int next;          // entry the next miss overwrites: the one after the entry used last
};

template <int NB, bool LR = false>
__device__ __forceinline__ double gcv_trace_direct(const WaveShared &S, int k, double c, const int *list, int lane, GcvCache<NB> &tc, bool reuse, int slot,
                                                   unsigned long long *cyc = nullptr)
{
    const double sc = 1.0;                   // the Gram matrix and its tridiagonal form are built at c = 1 (see GcvCache)
#ifdef MET2_CYCSTATS
    unsigned long long tc0 = __builtin_readcyclecounter();
#define MET2_GCV_LAP(slot) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (cyc) cyc[slot] += t_ - tc0; tc0 = t_; } while (0)
#else
#define MET2_GCV_LAP(slot)
#endif
    const int m = LR ? MET2_GCV_LR_RANK : S.m, n = m + 1, np = gcv_row_stride(m);      // n: order of T
    double *M = S.R;                         // C (LR: W) = M[1.., 1..]: [m][np], lane a <-> row a of C = row a + 1 of M
    double *vb = M + m * np;                 // [gcv_cols] Householder vector, zero padded (the support list sits here during step 1)
    double *wb = vb + gcv_vec_len(m);
    double *vb2 = wb + gcv_vec_len(m);       // the Householder vectors alternate between vb and vb2
    double dj = 0.0, ej = 0.0;               // lane j: T[j][j], T[j+1][j] of the (m + 1) x (m + 1) tridiagonal form
    if (!reuse) {
    const int n = m;                         // (inside this block: the order of C)
    // ---- 1. M = A A^T on the matrix cores
    {
        const int li = lane & 15, lk = lane >> 4;
        double bvec = 0.0;
        if (LR) {
            // W = A_S A_S^T: one 16 x 16 tile; S.DtG is the flip angle's A^T, [bin][16] -- four support bins = 64 contiguous doubles per k-step.
            // beta = A_S 1 falls out of the same operands.
            met2_d4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int r0 = 0; r0 < k; r0 += 16) {
                double a[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rr = r0 + 4 * q + lk;
                    const double va = S.DtG[(size_t)list[min(rr, k - 1)] * MET2_GCV_LR_RANK + li];
                    a[q] = (rr < k) ? va : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (r0 + 4 * q < k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], a[q], acc, 0, 0, 0);
                    bvec += a[q];
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) M[(lk + 4 * v) * np + li] = acc[v];
            bvec += gather(bvec, lane ^ 16);
            bvec += gather(bvec, lane ^ 32);                             // lanes li, li + 16, ...: beta[li]
        } else {
        const int nt = (n + 15) >> 4;
        for (int ti = 0; ti < nt; ++ti)
            for (int tj = ti; tj < nt; ++tj) {
                met2_d4 acc = {0.0, 0.0, 0.0, 0.0};
                const int ra = 16 * ti + li, rb = 16 * tj + li;
                const int ea = min(ra, m - 1), eb = min(rb, m - 1);
                for (int r0 = 0; r0 < k; r0 += 16) {            // four k-steps per batch: eight operand loads in flight
                    double a[4], b[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rr = r0 + 4 * q + lk;
                        const double *col = S.DtG + (size_t)list[min(rr, k - 1)] * S.dtstride;    // column s_rr of D, contiguous in D^T
                        const double va = col[ea];
                        const double vbb = col[eb];
                        a[q] = (rr < k && ra < n) ? va : 0.0;
                        b[q] = (rr < k && rb < n) ? vbb : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (r0 + 4 * q < k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * ti + lk + 4 * v, c = 16 * tj + li;
                    if (row < n && c < n) { M[row * np + c] = acc[v]; M[c * np + row] = acc[v]; }
                }
            }
        // column 0 of M below its diagonal: b = Dr 1 (row sums of D over the support), lane e <-> echo e; parked in wb until the
        // first reflector has been built from it (wb takes that step's w afterwards)
        {
            const int le = min(lane, m - 1);
            for (int r0 = 0; r0 < k; r0 += 4) {
                double t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = S.DtG[(size_t)list[min(r0 + q, k - 1)] * S.dtstride + le];
#pragma unroll
                for (int q = 0; q < 4; ++q) bvec += (r0 + q < k) ? t[q] : 0.0;
            }
        }
        }
        __builtin_amdgcn_wave_barrier();                                 // (the list is dead from here: the vectors' room takes b and the padding)
        if (lane < m) wb[lane] = sc * bvec;
        const int pad = gcv_cols(m) - m;                                 // 0 .. 3 zero columns / entries
        for (int q = 0; q < pad; ++q) {
            if (lane < n) M[lane * np + n + q] = 0.0;
            if (lane == 0) { vb[n + q] = 0.0; wb[n + q] = 0.0; vb2[n + q] = 0.0; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    MET2_GCV_LAP(8);
    // ---- 2. Householder tridiagonalisation (dsytd2, lower), lane a <-> row a.
    // One pass over the trailing block per step: the rank-2 update of step j - 1 (M -= v w^T + w v^T) is applied while the
    // product p = M v of step j is accumulated from the updated values (the arithmetic per element and the order of the sums are
    // those of two separate passes: same bits).  Column j, which the new reflector is built from, gets the pending update from
    // registers first.  The steps are a chain of dependent LDS round trips at two waves per SIMD, so a pass less per step is
    // latency, not bandwidth.
    double *Mrow = M + min(lane, n - 1) * np;
    double *vprev = vb, *vnext = vb2;        // LDS copies of the pending and of the new reflector (uniform reads of their entries)
    double vp = 0.0, wp = 0.0;               // this lane's entries of the pending reflector (v, w)
    bool pend = false;
    // apply the pending update to columns >= c0 (aligned down to four) without a product
    auto flush = [&](int c0) {
        if (lane < n) {
            for (int b = c0 & ~3; b < n; b += 4) {
                double m0, m1, m2, m3, u0, u1, u2, u3, q0, q1, q2, q3;
                lds_quad128(Mrow + b, m0, m1, m2, m3);
                lds_quad128(vprev + b, u0, u1, u2, u3);
                lds_quad128(wb + b, q0, q1, q2, q3);
                met2_d2 *dst = (met2_d2 *)__builtin_assume_aligned(Mrow + b, 16);
                met2_d2 o0, o1;
                o0.x = fma(-wp, u0, fma(-vp, q0, m0)); o0.y = fma(-wp, u1, fma(-vp, q1, m1));
                o1.x = fma(-wp, u2, fma(-vp, q2, m2)); o1.y = fma(-wp, u3, fma(-vp, q3, m3));
                dst[0] = o0; dst[1] = o1;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    double e_pre = 0.0;                                               // T[1][0] (at c = 1): comes out of the reflector built from b
    for (int j = -1; j + 2 < n; ++j) {                                // j = -1: column 0 of M, i.e. b; then the columns of C
        const bool below = lane > j && lane < n;
        double mj = (j < 0) ? wb[min(lane, n - 1)] : Mrow[max(j, 0)];
        if (pend) {                                                   // column j of the matrix the pending update produces
            const double vj = bcast(vp, j), wj = bcast(wp, j);
            mj = fma(-wp, vj, fma(-vp, wj, mj));
        }
        const double x = below ? mj : 0.0;
        if (lane == j) dj = mj;
        const double x0 = bcast(x, j + 1);
        const double xn2 = wave_sum((lane > j + 1) ? x * x : 0.0);
        if (xn2 == 0.0) {                                             // column already in tridiagonal form
            if (lane == j) ej = x0;
            if (j < 0) e_pre = x0;
            if (pend) { flush(j + 1); pend = false; }
            continue;
        }
        const double s2 = fma(x0, x0, xn2);
        const double nx = s2 * rsqrt_nr(s2);
        const double alpha = (x0 >= 0.0) ? -nx : nx;
        const double v0 = x0 - alpha;
        const double tau = 2.0 * rcp_nr(fma(v0, v0, xn2));            // H = I - tau v v^T
        const double v = (lane == j + 1) ? v0 : x;                    // zero outside (j, n)
        // p = tau M v, w = p - (tau/2)(p.v) v: lane a walks its own row, v and w of the other rows are uniform-address LDS reads
        // (broadcasting them with v_readlane instead costs six VALU instructions per matrix element and was measured 1.5x slower)
        if (lane < n) vnext[lane] = v;
        __builtin_amdgcn_wave_barrier();
        double p = 0.0, p2 = 0.0;
        const int b0 = (j + 1) & ~3;                                  // 16-byte aligned start: the up to three columns <= j it takes in have v = 0
        if (pend) {
            const bool upd = lane >= j && lane < n;                   // rows the pending update (of step j - 1) touches
#pragma unroll 2
            for (int b = b0; b < n; b += 4) {                         // may run into the zero padding
                double m0, m1, m2, m3, u0, u1, u2, u3, q0, q1, q2, q3, t0, t1, t2, t3;
                lds_quad128(Mrow + b, m0, m1, m2, m3);
                lds_quad128(vprev + b, u0, u1, u2, u3);
                lds_quad128(wb + b, q0, q1, q2, q3);
                lds_quad128(vnext + b, t0, t1, t2, t3);
                met2_d2 o0, o1;
                o0.x = fma(-wp, u0, fma(-vp, q0, m0)); o0.y = fma(-wp, u1, fma(-vp, q1, m1));
                o1.x = fma(-wp, u2, fma(-vp, q2, m2)); o1.y = fma(-wp, u3, fma(-vp, q3, m3));
                if (upd) { met2_d2 *dst = (met2_d2 *)__builtin_assume_aligned(Mrow + b, 16); dst[0] = o0; dst[1] = o1; }
                p = fma(o0.x, t0, p); p2 = fma(o0.y, t1, p2); p = fma(o1.x, t2, p); p2 = fma(o1.y, t3, p2);
            }
        } else {
#pragma unroll 2
            for (int b = b0; b < n; b += 4) {
                double m0, m1, m2, m3, t0, t1, t2, t3;
                lds_quad128(Mrow + b, m0, m1, m2, m3);
                lds_quad128(vnext + b, t0, t1, t2, t3);
                p = fma(m0, t0, p); p2 = fma(m1, t1, p2); p = fma(m2, t2, p); p2 = fma(m3, t3, p2);
            }
        }
        p = below ? tau * (p + p2) : 0.0;
        const double K = 0.5 * tau * wave_sum(p * v);
        const double w = fma(-K, v, p);
        __builtin_amdgcn_wave_barrier();
        if (lane < n) wb[lane] = w;                                   // (the pass above was the last reader of the old w)
        __builtin_amdgcn_wave_barrier();
        if (lane == j) ej = alpha;
        if (j < 0) e_pre = alpha;
        vp = v; wp = w; pend = true;
        double *tswap = vprev; vprev = vnext; vnext = tswap;
    }
    if (pend) flush(n - 2);                                            // the trailing 2 x 2 block still waits for the last update
    if (lane == n - 2) { dj = Mrow[n - 2]; ej = M[(n - 1) * np + n - 2]; }
    if (lane == n - 1) { dj = Mrow[n - 1]; ej = 0.0; }
    {   // lane a holds the entries of C's row a = row a + 1 of T: one lane up; lane 0 takes (d_0 set below, e_pre)
        const double dsh = gather(dj, (lane + 63) & 63), esh = gather(ej, (lane + 63) & 63);
        dj = (lane >= 1 && lane <= m) ? dsh : 0.0;
        ej = (lane == 0) ? e_pre : ((lane >= 1 && lane <= m) ? esh : 0.0);
    }
#pragma unroll
    for (int q = 0; q < MET2_GCV_CACHE; ++q) if (slot == q) { tc.d[q] = dj; tc.e[q] = ej; }
    } else {
#pragma unroll
        for (int q = 0; q < MET2_GCV_CACHE; ++q) if (slot == q) { dj = tc.d[q]; ej = tc.e[q]; }
        MET2_GCV_LAP(8);
    }
    if (lane == 0) { dj = c * (double)k; ej = sqrt(c) * ej; }          // the only entries of T that depend on c
    // (d_j, e_{j-1}^2) pairs; Gershgorin bound
    const double eg = gather(ej, (lane + 63) & 63);                    // cross-lane reads need the full wave: select afterwards
    const double eprev = (lane > 0 && lane < n) ? eg : 0.0;
    const double e2prev = eprev * eprev;                               // lane j: (d_j, e_{j-1}^2)
    const double bound = fmax(wave_max((lane < n) ? fabs(dj) + fabs(ej) + fabs(eprev) : 0.0), 1e-300);
    const double dmax = fmax(wave_max((lane < n) ? dj : 0.0), bound * 1e-18);      // the largest eigenvalue is at least the largest diagonal entry (Rayleigh)
    __builtin_amdgcn_wave_barrier();
    MET2_GCV_LAP(9);
    // ---- 3. the eigenvalues that survive the cut, by multisection on the bit patterns.
    // Only eigenvalues above eps k mu_max enter the trace (np.linalg.lstsq drops the rest) -- 7 to 16 of the 33 / 49 -- so the lanes are
    // spent on those instead of one lane per eigenvalue:
    //   A. the largest eigenvalue, coarsely: it lies between the largest diagonal entry and the Gershgorin bound (at most 3 x apart
    //      for a positive semi-definite T: e_i^2 <= d_i d_{i+1}); all 64 lanes x 2 shifts = a 129-section per Sturm pass, 2 passes
    //      -> 1e-4 relative, enough for the cut eps k mu_max (the reference's own sigma_max carries more noise than that at the cut);
    //   B. r = number of eigenvalues above the cut: one pass;
    //   C. the r largest, mu_max among them: a group of G = 64 / 16 = 4 lanes x 2 shifts per eigenvalue = a 9-section per pass
    //      (3.17 bits), 9 passes; more than 16 survivors (not seen on the reference's recipe): the one-lane-per-eigenvalue
    //      quaternary search of round 2.
    // 12 passes of 2 chains instead of 14 of 3, and the eigenvalues under the cut are never refined.
    // the Gershgorin bound is at most sqrt(n) mu_max, so bound * 1e-18 lies below any cut eps k mu_max
    double mu;
    int r_keep;
    {
        unsigned long long lo = (unsigned long long)__double_as_longlong(dmax * 0.9999999);
        unsigned long long hi = (unsigned long long)__double_as_longlong(bound * 1.0000001);
        for (int step = 0; step < 2; ++step) {                       // A: 129^2 sections of at most log2(3) octaves -> 7e-5 relative
            const unsigned long long q = (hi - lo) / 129ull;
            const unsigned long long m1 = lo + (unsigned long long)(2 * lane + 1) * q, m2 = m1 + q;
            const double xs[2] = {__longlong_as_double((long long)m1), __longlong_as_double((long long)m2)};
            int c[2];
            sturm_count<2>(dj, e2prev, n, xs, c);
            const u64 a1 = ballot(c[0] >= n), a2 = ballot(c[1] >= n);                // all n eigenvalues below the shift
            // shifts in ascending order: lane 0 (m1, m2), lane 1, ...: the first shift above the largest eigenvalue
            const int f1 = a1 ? first_lane(a1) : 64, f2 = a2 ? first_lane(a2) : 64;
            const int first = min(2 * f1, 2 * f2 + 1);                              // index 0 .. 127 into the shifts, 128: none
            const unsigned long long nhi = first < 128 ? lo + (unsigned long long)(first + 1) * q : hi;
            const unsigned long long nlo = first > 0 ? lo + (unsigned long long)first * q : lo;
            hi = nhi; lo = nlo;
        }
        const double mumax_ = __longlong_as_double((long long)hi);
        const double cut_ = 2.220446049250313e-16 * (double)k * mumax_;
        {                                                                            // B
            const double xs[1] = {cut_};
            int c[1];
            sturm_count<1>(dj, e2prev, n, xs, c);
            r_keep = n - c[0];
        }
        if (r_keep <= 16) {                                                          // C
            // (groups of 8 or 16 lanes when at most 8 or 4 eigenvalues survive -- 7 passes of a 17-section, 6 of a 33-section --
            //  measured the same as the fixed groups of four: 234.2 against 232.4 ms on configs[4]'s 131 072 voxels)
            const int e = lane >> 2, sidx = lane & 3;
            const int want = n - e;                                                  // the (e+1)-th largest: at least `want` eigenvalues below the shift
            unsigned long long glo = (e == 0) ? lo : (unsigned long long)__double_as_longlong(cut_ * 0.9999999);   // group 0: mu_max, inside A's bracket
            unsigned long long ghi = hi;
            for (int step = 0; step < 9; ++step) {                                   // 9^9 = 3.9e8 > 2^28
                const unsigned long long q = (ghi - glo) / 9ull;
                const unsigned long long m1 = glo + (unsigned long long)(2 * sidx + 1) * q, m2 = m1 + q;
                const double xs[2] = {__longlong_as_double((long long)m1), __longlong_as_double((long long)m2)};
                int c[2];
                sturm_count<2>(dj, e2prev, n, xs, c);
                const u64 a1 = ballot(c[0] >= want), a2 = ballot(c[1] >= want);
                const unsigned q1 = (unsigned)(a1 >> (lane & ~3)) & 15u, q2 = (unsigned)(a2 >> (lane & ~3)) & 15u;   // the group's four lanes
                const int f1 = q1 ? __builtin_ctz(q1) : 4, f2 = q2 ? __builtin_ctz(q2) : 4;
                const int first = min(2 * f1, 2 * f2 + 1);                          // 0 .. 7, 8: none
                const unsigned long long nhi = first < 8 ? glo + (unsigned long long)(first + 1) * q : ghi;
                const unsigned long long nlo = first > 0 ? glo + (unsigned long long)first * q : glo;
                ghi = nhi; glo = nlo;
            }
            // lane i < r takes eigenvalue i (its group's result sits in lanes 4 i .. 4 i + 3)
            const double mug = __longlong_as_double((long long)ghi);
            mu = gather(mug, (4 * lane) & 63);
            mu = (lane < r_keep) ? mu : 0.0;                                         // under the cut: dropped below (mu > cut fails)
        } else {
            unsigned long long lo2 = (unsigned long long)__double_as_longlong(bound * 1e-18);
            unsigned long long hi2 = (unsigned long long)__double_as_longlong(bound * 1.0000001);
            const int want = n - lane;                                               // eigenvalue number (1-based, ascending) this lane is after
            for (int step = 0; step < 14; ++step) {
                const unsigned long long q = (hi2 - lo2) >> 2;
                const unsigned long long m1 = lo2 + q, m2 = lo2 + 2 * q, m3 = lo2 + 3 * q;
                const double xs[3] = {__longlong_as_double((long long)m1), __longlong_as_double((long long)m2), __longlong_as_double((long long)m3)};
                int c[3];
                sturm_count<3>(dj, e2prev, n, xs, c);
                const bool u1 = c[0] >= want, u2 = c[1] >= want, u3 = c[2] >= want;
                hi2 = u1 ? m1 : (u2 ? m2 : (u3 ? m3 : hi2));
                lo2 = u1 ? lo2 : (u2 ? m1 : (u3 ? m2 : m3));
            }
            mu = __longlong_as_double((long long)hi2);
        }
    }
    MET2_GCV_LAP(10);
    // ---- 4. squared first eigenvector components: 1 / r_0'(mu)
    const double pivmin = 1e-290;
    double rp = 1.0;
    {
        double r = mu - bcast(dj, n - 1);
        r = (fabs(r) < pivmin) ? pivmin : r;
        for (int jj = n - 2; jj >= 0; --jj) {
            const double e2 = bcast(e2prev, jj + 1);         // e_jj^2
            const double inv = rcp_nr(r);
            rp = fma(e2 * rp * inv, inv, 1.0);
            r = fma(-e2, inv, mu - bcast(dj, jj));
            r = (jj > 0 && fabs(r) < pivmin) ? pivmin : r;
        }
    }
    const double w0 = rcp_nr(rp);
    const double mumax = bcast(mu, 0);
    const double cut = 2.220446049250313e-16 * (double)k * mumax;
    const bool keep = (lane < n) && (mu > cut);
    MET2_GCV_LAP(11);
    return wave_sum(keep ? (1.0 - w0) : 0.0);
}

// algorithms.py:285-296 given the NNLS solution st.x at lambda = x:
//   log( (r^2/m) / ((m - trace(Dr G^+ Dr^T))/m)^2 )
template <int NB, bool LR = false>
__device__ __forceinline__ double gcv_objective(const WaveShared &S, const Band<NB> &bd, const NnlsState<NB> &st, double x, double b,
                                                int lane, int &overflow, GcvCache<NB> &tc)
{
    const int n = S.n, m = S.m, mt = LR ? MET2_GCV_LR_RANK : S.m;      // mt: order of the matrix the trace is taken from
#ifdef MET2_CYCSTATS
    NnlsState<NB> &stw = const_cast<NnlsState<NB> &>(st);
#endif
    const double sse = sse_of<NB>(S, st, b, lane);
    const double rn2 = sse + x * seminorm2<NB>(bd, st.x, n, lane);   // squared residual norm of the augmented system
    bool inS[NB];
    u64 Sm[NB];
    int k = 0;
    double l2 = 0.0;
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        inS[bb] = (lane + 64 * bb < n) && (st.x[bb] > 0.0);
        Sm[bb] = ballot(inS[bb]);
        k += __popcll(Sm[bb]);
        const double ld = bd.lb[bb][2];
        l2 += inS[bb] ? ld * ld : 0.0;
    }
    if (k == 0) return NAN;
    if (gcv_lds_doubles(mt, k) > S.rcap) { overflow = 1; return INFINITY; }
    const double c = x * wave_sum(l2);
    int slot = -1;
#pragma unroll
    for (int q = 0; q < MET2_GCV_CACHE; ++q) {
        bool hit = ((tc.valid >> q) & 1) != 0;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) hit = hit && (tc.Sm[q][bb] == Sm[bb]);
        if (hit) slot = q;
    }
    const bool reuse = slot >= 0;
    if (!reuse) slot = tc.next;
    tc.next = (slot + 1 == MET2_GCV_CACHE) ? 0 : slot + 1;             // two entries: a miss overwrites the one not used last
    // support list (ascending bins) behind C in the wave's LDS region (where the Householder vectors go once the Gram contraction has read it)
    int *list = (int *)(S.R + mt * gcv_row_stride(mt));
    if (!reuse) {
        int base = 0;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int rank = base + __popcll(Sm[bb] & ((1ull << lane) - 1ull));
            if (inS[bb]) list[rank] = lane + 64 * bb;
            base += __popcll(Sm[bb]);
#pragma unroll
            for (int q = 0; q < MET2_GCV_CACHE; ++q) if (slot == q) tc.Sm[q][bb] = Sm[bb];
        }
        tc.valid |= 1 << slot;
        __builtin_amdgcn_wave_barrier();
    }
#ifdef MET2_CYCSTATS
    const unsigned long long cs0 = __builtin_readcyclecounter();
#endif
#ifdef MET2_CYCSTATS
    const double tr = gcv_trace_direct<NB, LR>(S, k, c, list, lane, tc, reuse, slot, stw.cyc);
#else
    const double tr = gcv_trace_direct<NB, LR>(S, k, c, list, lane, tc, reuse, slot);
#endif
#ifdef MET2_CYCSTATS
    stw.cyc[5] += __builtin_readcyclecounter() - cs0; stw.cyc[6] += 1; stw.cyc[4] += k;
#endif
    const double num = (1.0 / m) * rn2;
    const double den = (1.0 / m) * ((double)m - tr);
    return log(num / (den * den));
}

} // namespace met2
