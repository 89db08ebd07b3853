// objectives.hpp -- lambda-selection objectives that need more than the NNLS solve:
//   BayesReg  bayesian_interpolation.py:107-126  full n x n Cholesky of beta (B + lambda K), erf, log
//   GCV       algorithms.py:285-296              truncated pseudo-inverse of the support Gram matrix
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

// Upper Cholesky factor U (A = U^T U) of A = beta*B + (beta*lam)*K, packed by columns into S.R like the solver's
// factor (entry (r, c) at col_base(c) + r; needs kmax == n).  Row by row as in refactor(): a lane owns columns
// lane + 64 b, row j costs j independent LDS reads + FMAs in batches of four, the rows of B and of the dense K for
// the next pivot are in flight meanwhile.  Returns false when a pivot is not positive (scipy raises LinAlgError there).
template <int NB>
__device__ __forceinline__ bool chol_full(const WaveShared &S, const Band<NB> &bd, double beta, double lam, int lane, double &det_u)
{
    const int n = S.n;
    const double bl = beta * lam;
    int cbl[NB], cbc[NB];
    unsigned jc[NB];
    double diag[NB], gb0[NB], gk0[NB], gb1[NB], gk1[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        cbl[b] = col_base(pl);
        cbc[b] = col_base(min(pl, n - 1));
        jc[b] = (unsigned)min(pl, n - 1);
        diag[b] = 1.0;
    }
    auto fetch = [&](int j, double (&vb)[NB], double (&vk)[NB]) {
        const int jj = min(j, n - 1);
        const double *Brow = S.B + jj * S.bstride, *Krow = S.K + jj * n;
#pragma unroll
        for (int b = 0; b < NB; ++b) { vb[b] = Brow[jc[b]]; vk[b] = Krow[jc[b]]; }
    };
    // scale a finished row, store its column entries; false when the pivot is not positive
    auto finish = [&](int j, double (&a)[NB], double (&u)[NB]) -> bool {
        const double d = bcastN<NB>(a, j);
        const double rinv = rsqrt_nr(d);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c = lane + 64 * b;
            u[b] = a[b] * rinv;                                         // lane j: d * rinv = U[j][j]
            if (c >= j && c < n) S.R[cbl[b] + j] = u[b];
            if (c == j) diag[b] = u[b];
        }
        return d > 0.0;
    };
    fetch(0, gb0, gk0);
    fetch(1, gb1, gk1);
    int cbj = 0;                                                        // col_base(j)
    int j = 0;
    for (; j + 1 < n; j += 2) {                                         // rows j and j + 1 together, as in refactor()
        double a[NB], a2[NB], c[NB], c2[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            a[b] = fma(bl, gk0[b], beta * gb0[b]); c[b] = fma(bl, gk1[b], beta * gb1[b]);                     // A[j][.], A[j+1][.]
            a2[b] = 0.0; c2[b] = 0.0;
        }
        fetch(j + 2, gb0, gk0);
        fetch(j + 3, gb1, gk1);
        const double *cj = S.R + cbj, *cj1 = cj + j + 1;                // columns j and j + 1
        int k = 0;
#pragma clang loop unroll(disable)
        for (; k + 2 <= j; k += 2) {
            const double s0 = cj[k], s1 = cj[k + 1];
            const double u0 = cj1[k], u1 = cj1[k + 1];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const double *cc = S.R + cbc[b] + k;
                const double q0 = cc[0], q1 = cc[1];
                a[b] = fma(-s0, q0, a[b]); a2[b] = fma(-s1, q1, a2[b]);
                c[b] = fma(-u0, q0, c[b]); c2[b] = fma(-u1, q1, c2[b]);
            }
        }
        for (; k < j; ++k) {
            const double s0 = cj[k], u0 = cj1[k];
#pragma unroll
            for (int b = 0; b < NB; ++b) { const double q0 = S.R[cbc[b] + k]; a[b] = fma(-s0, q0, a[b]); c[b] = fma(-u0, q0, c[b]); }
        }
        double u[NB], w[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { a[b] += a2[b]; c[b] += c2[b]; }
        if (!finish(j, a, u)) return false;
        const double su = bcastN<NB>(u, j + 1);                         // U[j][j+1]
#pragma unroll
        for (int b = 0; b < NB; ++b) c[b] = fma(-su, u[b], c[b]);
        if (!finish(j + 1, c, w)) return false;
        __builtin_amdgcn_wave_barrier();
        cbj += 2 * j + 3;
    }
    if (j < n) {                                                        // odd n: the last row on its own
        double a[NB], u[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) a[b] = fma(bl, gk0[b], beta * gb0[b]);
        const double *cj = S.R + cbj;
        for (int k = 0; k < j; ++k) {
            const double s0 = cj[k];
#pragma unroll
            for (int b = 0; b < NB; ++b) a[b] = fma(-s0, S.R[cbc[b] + k], a[b]);
        }
        if (!finish(j, a, u)) return false;
        __builtin_amdgcn_wave_barrier();
    }
    double dp = 1.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) dp *= (lane + 64 * b < n) ? diag[b] : 1.0;
    det_u = wave_prod(dp);
    return true;
}

// (U f)_i for the factor in S.R; the owner of row i gets row i . f
template <int NB>
__device__ __forceinline__ void upper_times(const WaveShared &S, const double (&f)[NB], int lane, double (&out)[NB])
{
    const int n = S.n;
#pragma unroll
    for (int b = 0; b < NB; ++b) out[b] = 0.0;
    int cbj = 0;
    for (int j = 0; j < n; ++j) {
        const double fj = bcastN<NB>(f, j);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = lane + 64 * b;
            double u = (i <= j) ? S.R[cbj + i] : 0.0;                   // column j: rows 0..j
            out[b] = fma(u, fj, out[b]);
        }
        cbj += j + 1;
    }
}

struct BayesCtx {
    double beta, log_detL;
    int failed;     // Cholesky failure seen
};

// bayesian_interpolation.py:107-126, given the NNLS solution st.x at lambda = x
template <int NB>
__device__ __forceinline__ double bayes_objective(const WaveShared &S, const Band<NB> &bd, const NnlsState<NB> &st, BayesCtx &bc,
                                                  double x, double b, int lane)
{
    const int n = S.n, m = S.m;
    const double beta = bc.beta;
    const double ED = 0.5 * sse_of<NB>(S, st, b, lane);
    const double EW = 0.5 * seminorm2<NB>(bd, st.x, n, lane);
    double det_u;
#ifdef MET2_CYCSTATS
    NnlsState<NB> &stw = const_cast<NnlsState<NB> &>(st);
    const unsigned long long c0 = __builtin_readcyclecounter();
#endif
    if (!chol_full<NB>(S, bd, beta, x, lane, det_u)) { bc.failed = 1; return NAN; }
#ifdef MET2_CYCSTATS
    const unsigned long long c1 = __builtin_readcyclecounter();
    stw.cyc[5] += c1 - c0;
#endif
    double uf[NB];
    upper_times<NB>(S, st.x, lane, uf);
#ifdef MET2_CYCSTATS
    const unsigned long long c2 = __builtin_readcyclecounter();
    stw.cyc[6] += c2 - c1;
#endif
    double term = 0.0;
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) term += (lane + 64 * bb < n) ? log(1.0 + erf((1.0 / sqrt(2.0)) * uf[bb])) : 0.0;
    const double series = wave_sum(term);
#ifdef MET2_CYCSTATS
    stw.cyc[7] += __builtin_readcyclecounter() - c2;
#endif
    const double PI = M_PI;
    double cost1 = beta * ED + beta * x * EW + log(det_u) - (n / 2.0) * log(PI / 2.0) - series;
    double cost2 = (m / 2.0) * log(2.0 * PI) - (m / 2.0) * log(beta) + (n / 2.0) * log(PI) - (n / 2.0) * log(2 * beta * x) - bc.log_detL;
    return cost1 + cost2;
}

// One-sided Jacobi on the k columns of E ((m+1) x k), for supports of at most 32 bins: E V = U Sigma, so after
// convergence column r holds sigma_r u_r -- its squared norm is an eigenvalue of G = E^T E and its last entry over
// sigma_r is U[m][r]; no rotation has to be accumulated.  Round-robin ordering over the k columns (k - 1 rounds of
// up to 16 disjoint pairs instead of the m rounds of the E^T variant); a pair owns four lanes, each keeps its slice
// of both columns (<= MET2_GCV_ROWS rows) in registers between the inner product and the rotation.
// Returns trace(Dr G^+ Dr^T) = sum_{kept r} (1 - U[m][r]^2).
#define MET2_GCV_ROWS 13                                      // ceil((n_te + 1) / 4) for n_te <= 51
// column stride = 4 mod 32 doubles: the four lanes of a pair read consecutive rows, the 16 pairs of a round land
// on banks 4 apart
__device__ __forceinline__ int gcv_small_stride(int mm) { return ((mm + 27) / 32) * 32 + 4; }
template <int NB>
__device__ __forceinline__ double gcv_trace_small(const WaveShared &S, const int (&sp)[NB], int k, double sc, int lane, int &nsweep_done)
{
    nsweep_done = 0;
    const int m = S.m, mm = m + 1, mmp = gcv_small_stride(mm);
    double *Bm = S.R;                                       // column r at Bm + r * mmp, rows 0..m
    double *cn2 = Bm + k * mmp;                             // [k] squared column norms
    {
        const unsigned ec = (unsigned)min(lane, m - 1);
        for (int r = 0; r < k; r += 4) {
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = min(r + q, k - 1);
                const double *Drow = S.Dt + bcastN_i<NB>(sp, rr) * S.dtstride;
                v[q] = Drow[ec];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) if (r + q < k && lane < mm) Bm[(r + q) * mmp + lane] = (lane < m) ? v[q] : sc;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int odd = k & 1;
    const int N = k + odd, nreal = N / 2 - odd;             // with k odd the fixed player is a dummy, its pair is skipped
    const int pi = lane >> 2, sub = lane & 3;
    const bool mine = pi < nreal;
    const int ti = pi + odd;
    for (int sweep = 0; sweep < 30; ++sweep) {
        {
            double t2 = 0.0;
            if (lane < k) for (int e = 0; e < mm; ++e) { const double v = Bm[lane * mmp + e]; t2 = fma(v, v, t2); }
            if (lane < k) cn2[lane] = t2;
            __builtin_amdgcn_wave_barrier();
        }
        const double big = wave_max(lane < k ? cn2[lane] : 0.0);
        const double floor2 = 1e-6 * (2.220446049250313e-16 * (double)k * big);
        const double noise2 = 16.0 * 4.930380657631324e-32 * big * (double)mm;
        u64 rotated = 0ull;
        for (int rd = 0; rd < N - 1; ++rd) {
            int ca, cb;
            if (ti == 0) { ca = rd; cb = N - 1; }
            else {
                ca = rd + ti; ca = (ca >= N - 1) ? ca - (N - 1) : ca;
                cb = rd - ti; cb = (cb < 0) ? cb + (N - 1) : cb;
            }
            if (ca > cb) { int t = ca; ca = cb; cb = t; }
            if (!mine) { ca = 0; cb = 0; }
            double *pa = Bm + ca * mmp + sub, *pb = Bm + cb * mmp + sub;
            double xa[MET2_GCV_ROWS], xb[MET2_GCV_ROWS];
            double gamma = 0.0;
#pragma unroll
            for (int q = 0; q < MET2_GCV_ROWS; ++q) {
                const bool in = sub + 4 * q < mm;
                xa[q] = in ? pa[4 * q] : 0.0; xb[q] = in ? pb[4 * q] : 0.0;
                gamma = fma(xa[q], xb[q], gamma);
            }
            const double alpha = mine ? cn2[ca] : 0.0, beta = mine ? cn2[cb] : 0.0;
            gamma += __shfl_xor(gamma, 1);
            gamma += __shfl_xor(gamma, 2);
            const double g2 = gamma * gamma;
            const bool rot = mine && !(alpha < floor2 && beta < floor2) && (g2 > 1e-30 * alpha * beta) && (g2 > noise2 * fmax(alpha, beta));
            rotated |= ballot(rot);
            const double a = beta - alpha, g = 2.0 * gamma;
            const double h2 = fma(a, a, g * g);
            const double hyp = (h2 > 0.0) ? h2 * rsqrt_nr(h2) : 0.0;
            const double den = (a >= 0.0) ? a + hyp : a - hyp;
            const double t = (den != 0.0) ? g * rcp_nr(den) : 0.0;
            const double cs = rsqrt_nr(fma(t, t, 1.0)), sn = cs * t;
            if (rot) {
#pragma unroll
                for (int q = 0; q < MET2_GCV_ROWS; ++q)
                    if (sub + 4 * q < mm) { pa[4 * q] = cs * xa[q] - sn * xb[q]; pb[4 * q] = sn * xa[q] + cs * xb[q]; }
                if (sub == 0) { cn2[ca] = alpha - t * gamma; cn2[cb] = beta + t * gamma; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        MET2_STAT(6, sweep + 1);
        nsweep_done = sweep + 1;
        if (!rotated) break;
    }
    double t2 = 0.0, last = 0.0;
    if (lane < k) {
        for (int e = 0; e < mm; ++e) { const double v = Bm[lane * mmp + e]; t2 = fma(v, v, t2); }
        last = Bm[lane * mmp + m];
    }
    const double smax = wave_max(lane < k ? t2 : 0.0);                   // eigenvalues of G = sigma(E)^2
    const double cut = 2.220446049250313e-16 * (double)k * smax;
    const bool keep = (lane < k) && (t2 > cut);
    return wave_sum(keep ? (1.0 - last * last / t2) : 0.0);
}

// algorithms.py:285-296 given the NNLS solution st.x at lambda = x:
//   log( (r^2/m) / ((m - trace(Dr G^+ Dr^T))/m)^2 ),  G = Dr^T Dr + c 11^T,  c = x * sum_{j in S} L_jj^2
// (the scalar-broadcast quirk of algorithms.py:289-293), G^+ = SVD-truncated pseudo-inverse with
// np.linalg.lstsq's cutoff eps*k*s_max.  With E = [Dr; sqrt(c) 1^T] ((m+1) x k) one has G = E^T E, so
// the singular values of G are the squared singular values of E and, for E = U S V^T,
//   trace(Dr G^+ Dr^T) = sum_{kept i} (1 - U[m][i]^2).
// E^T (k x (m+1)) is small in its column count whatever the support size: one-sided Jacobi on its
// m+1 columns in a round-robin ordering that rotates up to 16 (nTE=32) or 24 (nTE=48) disjoint pairs at once,
// carrying only the last row of the accumulated rotations.  Column norms are cached in LDS and refreshed
// every sweep; pairs whose columns both sit far below the cutoff are skipped.
template <int NB>
__device__ __forceinline__ double gcv_objective(const WaveShared &S, const Band<NB> &bd, const NnlsState<NB> &st, double x, double b,
                                                int lane, int &overflow)
{
    const int n = S.n, m = S.m, mm = m + 1;
#ifdef MET2_CYCSTATS
    NnlsState<NB> &stw = const_cast<NnlsState<NB> &>(st);
    const unsigned long long c0 = __builtin_readcyclecounter();
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
    if (mm * k + 2 * mm > S.rcap) { overflow = 1; return INFINITY; }
    const double c = x * wave_sum(l2);
    // support list through LDS: rank-th support bin -> sp of the owner of row `rank`
    int *list = (int *)S.R;
    int base = 0;
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        const int rank = base + __popcll(Sm[bb] & ((1ull << lane) - 1ull));
        if (inS[bb]) list[rank] = lane + 64 * bb;
        base += __popcll(Sm[bb]);
    }
    __builtin_amdgcn_wave_barrier();
    int sp[NB];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) sp[bb] = (lane + 64 * bb < k) ? list[lane + 64 * bb] : 0;
    __builtin_amdgcn_wave_barrier();
    double *A = S.R;                                   // column-major k x (m+1): A[e*k + r] = E[e][s_r]
    double *cn2 = A + mm * k;                          // [mm] squared column norms
    double *wl = cn2 + mm;                             // [mm] last row of the accumulated rotations
    const double sc = sqrt(c);
    if (k <= 32 && S.Dt && mm <= 4 * MET2_GCV_ROWS && k * gcv_small_stride(mm) + k <= S.rcap) {
#ifdef MET2_CYCSTATS
        const unsigned long long cs0 = __builtin_readcyclecounter();
#endif
        int nsw;
        const double tr = gcv_trace_small<NB>(S, sp, k, sc, lane, nsw);
#ifdef MET2_CYCSTATS
        stw.cyc[5] += __builtin_readcyclecounter() - cs0; stw.cyc[6] += 1; stw.cyc[7] += nsw; stw.cyc[4] += k;
#endif
        const double num = (1.0 / m) * rn2;
        const double den = (1.0 / m) * ((double)m - tr);
        return log(num / (den * den));
    }
    for (int e = 0; e < mm; ++e) {
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int r = lane + 64 * bb;
            if (r < k) A[e * k + r] = (e < m) ? S.D[e * S.dstride + sp[bb]] : sc;
        }
    }
    if (lane < mm) wl[lane] = (lane == m) ? 1.0 : 0.0;
    __builtin_amdgcn_wave_barrier();
    // Round-robin ("tournament") ordering: N = mm rounded up to even players, N-1 rounds of N/2 disjoint column
    // pairs; with mm odd the fixed player is a dummy and its pair is skipped.  Every pair gets LP lanes: the
    // lanes of a pair split the k rows, partial inner products meet through LP-wide xor shuffles, and the
    // rotation parameters (the expensive sqrt/div chain) are computed once per round for all pairs at once.
    const int odd = mm & 1;
    const int N = mm + odd, nreal = N / 2 - odd;
    const int LP = (nreal <= 16) ? 4 : (nreal <= 32 ? 2 : 1);
    const int pi = lane / LP, sub = lane - pi * LP;
    const bool mine = pi < nreal;
    const int ti = pi + odd;                           // pair index inside the round (0 = the fixed player's pair)
    for (int sweep = 0; sweep < 30; ++sweep) {
        // refresh the cached column norms (lane e walks its own column)
        {
            double t2 = 0.0;
            for (int r = 0; r < k; ++r) { double v = (lane < mm) ? A[lane * k + r] : 0.0; t2 = fma(v, v, t2); }
            if (lane < mm) cn2[lane] = t2;
            __builtin_amdgcn_wave_barrier();
        }
        const double big = wave_max(lane < mm ? cn2[lane] : 0.0);
        const double floor2 = 1e-6 * (2.220446049250313e-16 * (double)k * big);   // far below lstsq's cutoff on sigma^2
        const double noise2 = 16.0 * 4.930380657631324e-32 * big * (double)k;      // (4 eps)^2 * big * k
        u64 rotated = 0ull;
        for (int rd = 0; rd < N - 1; ++rd) {
            int ca, cb;
            if (ti == 0) { ca = rd; cb = N - 1; }
            else {                                      // (rd +- ti) mod (N-1): both operands are below N-1
                ca = rd + ti; ca = (ca >= N - 1) ? ca - (N - 1) : ca;
                cb = rd - ti; cb = (cb < 0) ? cb + (N - 1) : cb;
            }
            if (ca > cb) { int t = ca; ca = cb; cb = t; }
            const double alpha = mine ? cn2[ca] : 0.0, beta = mine ? cn2[cb] : 0.0;
            double gamma = 0.0;
            if (mine) for (int r = sub; r < k; r += LP) gamma = fma(A[ca * k + r], A[cb * k + r], gamma);
            if (LP >= 2) gamma += __shfl_xor(gamma, 1);
            if (LP >= 4) gamma += __shfl_xor(gamma, 2);
            const double g2 = gamma * gamma;
            // converged pair: orthogonal to working precision, or the inner product is at the level of the absolute
            // rounding noise (eps * sqrt(big) per entry) that cancellation left in small columns
            const bool rot = mine && !(alpha < floor2 && beta < floor2) && (g2 > 1e-30 * alpha * beta) && (g2 > noise2 * fmax(alpha, beta));
            rotated |= ballot(rot);
            // rotation parameters through v_rsq_f64 / v_rcp_f64 + Newton steps instead of IEEE sqrt and divisions
            // (a Jacobi rotation only has to be orthogonal, which cs and sn = cs t are to rounding)
            const double a = beta - alpha, g = 2.0 * gamma;
            const double h2 = fma(a, a, g * g);
            const double hyp = (h2 > 0.0) ? h2 * rsqrt_nr(h2) : 0.0;
            const double den = (a >= 0.0) ? a + hyp : a - hyp;
            const double t = (den != 0.0) ? g * rcp_nr(den) : 0.0;           // tan of the rotation angle, |t| <= 1
            const double cs = rsqrt_nr(fma(t, t, 1.0)), sn = cs * t;
            if (rot) {
                for (int r = sub; r < k; r += LP) {
                    const double ap = A[ca * k + r], aq = A[cb * k + r];
                    A[ca * k + r] = cs * ap - sn * aq;
                    A[cb * k + r] = sn * ap + cs * aq;
                }
                if (sub == 0) {
                    const double wp = wl[ca], wq = wl[cb];
                    wl[ca] = cs * wp - sn * wq; wl[cb] = sn * wp + cs * wq;
                    cn2[ca] = alpha - t * gamma; cn2[cb] = beta + t * gamma;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        MET2_STAT(6, sweep + 1);
        if (!rotated) break;
    }
    double t2 = 0.0;
    for (int r = 0; r < k; ++r) { double v = (lane < mm) ? A[lane * k + r] : 0.0; t2 = fma(v, v, t2); }
    const double smax = wave_max(lane < mm ? t2 : 0.0);                  // singular values of G = sigma(E)^2
    const double cut = 2.220446049250313e-16 * (double)k * smax;
    const bool keep = (lane < mm) && (t2 > cut);
    const double wv = (lane < mm) ? wl[lane] : 0.0;
    const double tr = wave_sum(keep ? (1.0 - wv * wv) : 0.0);
    const double num = (1.0 / m) * rn2;
    const double den = (1.0 / m) * ((double)m - tr);
    return log(num / (den * den));
}

} // namespace met2
